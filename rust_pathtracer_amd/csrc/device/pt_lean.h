// pt_lean.h -- the ordered walk, per lane (pt_device.h, "the ordered walk": the four numbered points there are what makes its
// answer the reference's, bvh.rs:107-121), with its per-step code written for the instruction count (rounds 1-2 spelt a box
// step as nested branches: 109 VALU + 102 SALU instructions; this form: 63 + ~20, same traversal, same tests, same proof step,
// same fallback, bit-identical results on the whole GPU suite):
//   * The guide tree and a copy of the record stream sit in ONE allocation (guide nodes first), so a lane's next fetch is a
//     32-bit offset from one scalar base whatever it points at -- no per-lane 64-bit pointer select.
//   * A step has no branches.  The far child is stored above the stack top whether or not it is wanted (the array has one
//     level to spare), the entry below the top is read at the start of the step whether or not it will be popped (its latency
//     hides behind the box tests), and "which child next / pop / nothing left" is a handful of selects.
//   * What happens when the traversal is over (miss / near tie -> literal walk / proof step) is decided in a phase of its own
//     (PT_LW_FINISH), once per ray, instead of being woven into every step.
//   * The f32 culling bound is kept in a register and refreshed when the best hit changes (a primitive test), not recomputed
//     from the f64 hit distance at every box step.
//   * A step does not load: it ends by aiming the lane at its next record (`cur`), and the wave loop (pt_ordered.h) issues the loads
//     right behind it, one site per kind of step, a primitive record's 80 bytes at once (1M-triangle mesh, 64 spp: 36.7 -> 34.5 ms
//     against a load inside every step; profiles/r03_wave_loop_aim_load_ab.txt).
// Modes of a lane: IDLE (no ray), GUIDE (a guide node's two child boxes are in r0..r3), LEAF (a primitive record is), FINISH (the
// traversal is over), VERIFY (the box of the best primitive's parent node is), LITERAL(_BEGIN) (the per-lane literal walk).
#pragma once
#include "pt_device.h"

static_assert(PT_GUIDE_STACK >= 6, "the literal walk parks the ray's three reciprocals in the first six slots of its lane's stack");
enum { PT_LW_IDLE = 0, PT_LW_GUIDE = 1, PT_LW_LEAF = 2, PT_LW_FINISH = 3, PT_LW_VERIFY = 4, PT_LW_LITERAL = 5, PT_LW_LITERAL_BEGIN = 6 };

struct LeanWalk {
    StreamWalk w;  // the ray and the best hit so far; its record fields (kind, aux, g[6]), inv, t_min and off: literal walk only
    // The fetched 64 bytes as the four 16-byte loads deliver them -- header and first double | two doubles | two doubles | sixth
    // double and the eight bytes behind it (a triangle's seventh double) -- so that every fetch site defines the same four
    // register quads and nothing is copied after a load.
    uint4 r0, r1, r2, r3;
    uint4 rt;             // a primitive record's bytes 64..79 (a triangle's last two doubles): fetched with the record, not after it
    uint32_t best_parent; // stream offset of the node that is the best primitive's parent in the reference's tree
    uint32_t cur;         // offset (from OrderedScene::base) of the fetched record
    uint32_t mode, sp;
    float bound;          // f32 culling bound of the box tests: near(best_t) rounded up (refreshed by step_leaf)
    pt_f32x2 q[5];        // (of.x, of.y) (of.z, invf.x) (invf.y, invf.z) (nrinv.x, nrinv.y) (nrinv.z, second_t): OrderedWalk::guide_slab

    // A step ends by aiming the lane at its next record (aim); the wave loop then loads it (load).
    // 64 bytes at base + cur: header, six doubles, and the eight bytes behind them; a primitive record: 80
    __device__ __forceinline__ void aim(const OrderedScene& sc, uint32_t off) {
        cur = min(off, sc.fetch_limit); // (whatever the tables hold, no fetch leaves the allocation: it is padded by 128 bytes)
    }
    __device__ __forceinline__ void load(const OrderedScene& sc) {
        const uint4* p = reinterpret_cast<const uint4*>(sc.base + cur);
        r0 = p[0]; r1 = p[1]; r2 = p[2]; r3 = p[3];
        if (mode == PT_LW_LEAF) rt = p[4]; // (cur <= fetch_limit: inside the padding)
    }
    static __device__ __forceinline__ double f64_of(uint32_t lo, uint32_t hi) { return __hiloint2double((int)hi, (int)lo); }
    static __device__ __forceinline__ pt_f32x2 pair(uint32_t lo, uint32_t hi) {
        pt_f32x2 r;
        r.x = __uint_as_float(lo);
        r.y = __uint_as_float(hi);
        return r;
    }
    // The entry below the stack top (anything when the stack is empty): read at the start of a step, used at its end.
    __device__ __forceinline__ uint32_t peek(const uint32_t* stack) const { return stack[(max(sp, 1u) - 1u) * PT_BLOCK]; }
    // Where the lane goes next: `ref` if `any`, else the popped entry, else the traversal is over.
    __device__ __forceinline__ void go(const OrderedScene& sc, bool any, uint32_t ref, uint32_t popped, bool overflow) {
        const bool pop = !any && sp > 0u;
        ref = any ? ref : popped;
        sp -= pop ? 1u : 0u;
        const bool leaf = (int32_t)ref < 0; // PT_GUIDE_LEAF
        mode = overflow ? (uint32_t)PT_LW_LITERAL_BEGIN : ((any || pop) ? (leaf ? (uint32_t)PT_LW_LEAF : (uint32_t)PT_LW_GUIDE) : (uint32_t)PT_LW_FINISH);
        aim(sc, leaf ? (ref & ~PT_GUIDE_LEAF) + sc.guide_bytes : ref << 6);
    }
    __device__ __forceinline__ void refresh_bound() { bound = OrderedWalk::f32_not_below(OrderedWalk::near(w.best_t)) + PT_GUIDE_ABS_SLACK; }
    __device__ __forceinline__ void begin(const OrderedScene& sc, const OrderedBounds& B, V3 o, V3 d) {
        w.o = o; w.d = d;
        w.best_t = B.t_max;
        w.best_prim = -1;
        w.off = 0;
        best_parent = 0; sp = 0;
        // Ordinary rays only: direction components in [2^-60, 2^100], origin within the scene's limit (at most 1e18).  Then (with
        // the scene's coordinates finite and below 1e18, checked at upload) no primitive test and no box test can overflow or produce
        // a NaN, which is what the guide's arithmetic, the nesting argument AND the premise "a primitive that reports a
        // hit is crossed inside its box" rest on: the reference accepts a sphere whose discriminant is inf - inf = NaN
        // (geometry.rs:132-148: every comparison with NaN is false) wherever the ray is, and a NaN bound then lets any
        // later hit replace a nearer one.  The same premise fails without any NaN when the origin is far from a
        // primitive relative to the primitive's size: the tests cancel and report hits for rays that pass whole units
        // beside it (a sphere of radius 0.5 seen from 1e9 away), outside the 1e-5 padding of its guide box, where the
        // reference -- which tests it whenever its parent's box is crossed -- still takes the hit.  The upload turns the
        // sizes of the scene's primitives into the two limits below.  All other rays take the literal walk.
        const double ax = fabs(d.x), ay = fabs(d.y), az = fabs(d.z);
        const double omax = fmax(fmax(fabs(o.x), fabs(o.y)), fabs(o.z)), dmax = fmax(fmax(ax, ay), az), dmin = fmin(fmin(ax, ay), az);
        const bool ordinary = dmin >= 0x1p-60 && dmax <= 0x1p100 && omax <= sc.origin_limit && (omax + sc.tri_extent) * dmax <= sc.od_limit &&
                              B.t_min == B.t_min && B.t_max == B.t_max;
        const double oc[3] = {o.x, o.y, o.z}, dc[3] = {d.x, d.y, d.z};
        float of[3], invf[3], nrinv[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            of[k] = (float)oc[k];
            const float rf = (float)(oc[k] - (double)of[k]);
            invf[k] = __builtin_amdgcn_rcpf((float)dc[k]);
            nrinv[k] = -(rf * invf[k]);
        }
        q[0].x = of[0]; q[0].y = of[1]; q[1].x = of[2]; q[1].y = invf[0]; q[2].x = invf[1]; q[2].y = invf[2];
        q[3].x = nrinv[0]; q[3].y = nrinv[1]; q[4].x = nrinv[2]; q[4].y = __builtin_huge_valf(); // no second candidate yet
        refresh_bound();
        mode = ordinary ? (uint32_t)PT_LW_GUIDE : (uint32_t)PT_LW_LITERAL_BEGIN;
        aim(sc, 0u); // the guide tree's root
        load(sc);
    }
    // OrderedWalk::guide_slab with the bound from the register and both acceptance conditions in one comparison:
    //   max(entry, t_min) <= min(exit, bound)   <=>   entry <= exit, entry <= bound, t_min <= exit   (and t_min <= bound, which
    //   holds whenever anything can be hit at all: the bound is a hit distance >= t_min, or the caller's t_max).
    __device__ __forceinline__ bool slab(pt_f32x2 x, pt_f32x2 y, pt_f32x2 z, float tmin_dn, float& entry) const {
        pt_f32x2 tx, ty, tz;
        PT_PK_SLAB(tx, x, q[0], 0, q[1], 1, q[3], 0);
        PT_PK_SLAB(ty, y, q[0], 1, q[2], 0, q[3], 1);
        PT_PK_SLAB(tz, z, q[1], 0, q[2], 1, q[4], 0);
        const float nx = OrderedWalk::raw_min(tx.x, tx.y), ny = OrderedWalk::raw_min(ty.x, ty.y), nz = OrderedWalk::raw_min(tz.x, tz.y);
        const float fx = OrderedWalk::raw_max(tx.x, tx.y), fy = OrderedWalk::raw_max(ty.x, ty.y), fz = OrderedWalk::raw_max(tz.x, tz.y);
        float tn, tf;
        asm("v_max3_f32 %0, %1, %2, %3" : "=v"(tn) : "v"(nx), "v"(ny), "v"(nz));
        asm("v_min3_f32 %0, %1, %2, %3" : "=v"(tf) : "v"(fx), "v"(fy), "v"(fz));
        tn = __builtin_fmaf(__builtin_fabsf(tn), -5e-7f, tn) - 2.0f * PT_GUIDE_ABS_SLACK;
        tf = __builtin_fmaf(__builtin_fabsf(tf), 5e-7f, tf);
        entry = tn;
        return OrderedWalk::raw_max(tn, tmin_dn) <= OrderedWalk::raw_min(tf, bound);
    }
    __device__ __forceinline__ void step_guide(const OrderedScene& sc, uint32_t* stack, const OrderedBounds& B) {
        const uint32_t popped = peek(stack);
        float e0, e1; // (the fetched 48 bytes: child 0's x, y, z pairs, then child 1's)
        const bool h0 = slab(pair(r0.z, r0.w), pair(r1.x, r1.y), pair(r1.z, r1.w), B.tmin_dn, e0);
        const bool h1 = slab(pair(r2.x, r2.y), pair(r2.z, r2.w), pair(r3.x, r3.y), B.tmin_dn, e1);
        const uint32_t c0 = r0.x, c1 = r0.y;
        const bool both = h0 && h1, swap = both && e1 < e0;
        const bool push = both && sp < PT_GUIDE_STACK;
        stack[sp * PT_BLOCK] = swap ? c0 : c1; // above the top; it only counts when sp moves (sp <= PT_GUIDE_STACK: the spare level)
        sp += push ? 1u : 0u;
        go(sc, h0 || h1, (swap || !h0) ? c1 : c0, popped, both && !push);
    }
    // a candidate from a primitive test run with the bound near(best): OrderedWalk::consider
    __device__ __forceinline__ void consider(bool hit, double t, uint32_t prim, uint32_t link) {
        if (!hit) return;
        if (t != t) {
            q[4].y = -__builtin_huge_valf();
        } else if ((int32_t)prim == w.best_prim) {
        } else if (t < w.best_t || w.best_prim < 0) {
            if (w.best_prim >= 0) q[4].y = __builtin_fminf(q[4].y, OrderedWalk::f32_not_above(w.best_t) - PT_GUIDE_ABS_SLACK);
            w.best_t = t; w.best_prim = (int32_t)prim; best_parent = link;
        } else {
            q[4].y = __builtin_fminf(q[4].y, OrderedWalk::f32_not_above(t) - PT_GUIDE_ABS_SLACK);
        }
    }
    __device__ __forceinline__ void step_leaf(const OrderedScene& sc, uint32_t* stack, const OrderedBounds& B) {
        const uint32_t popped = peek(stack);
        const double bound_t = w.best_prim < 0 ? B.t_max : OrderedWalk::near(w.best_t);
        const V3 o = w.o, d = w.d;
        const double g[6] = {f64_of(r0.z, r0.w), f64_of(r1.x, r1.y), f64_of(r1.z, r1.w), f64_of(r2.x, r2.y), f64_of(r2.z, r2.w), f64_of(r3.x, r3.y)};
        const double t_min = B.t_min;
        const uint32_t kind = r0.x & PT_WK_KIND_MASK, aux = r0.y, link = (r0.x >> PT_WK_LINK_SHIFT) * 8u;
        double t;
        if (kind == PT_WK_SPHERE) {
            consider(hit_sphere(g, o, d, t_min, bound_t, t), t, aux, link);
        } else if (kind == PT_WK_TRIANGLE) {
            const double tri[9] = {g[0], g[1], g[2], g[3], g[4], g[5], f64_of(r3.z, r3.w), f64_of(rt.x, rt.y), f64_of(rt.z, rt.w)};
            consider(hit_triangle(tri, o, d, t_min, bound_t, t), t, aux, link);
        } else if (kind == PT_WK_BOX6) {
            const double r0[5] = {g[2], g[0], g[3], g[1], g[4]}, r1[5] = {g[5], g[0], g[3], g[1], g[4]};
            const double r2[5] = {g[1], g[0], g[3], g[2], g[5]}, r3[5] = {g[4], g[0], g[3], g[2], g[5]};
            const double r4[5] = {g[0], g[1], g[4], g[2], g[5]}, r5[5] = {g[3], g[1], g[4], g[2], g[5]};
            // the bound is not tightened between the six sides: all candidates within near(best) must be seen
            consider(hit_rect(r0, o.z, d.z, o.x, d.x, o.y, d.y, t_min, bound_t, t), t, aux, link);
            consider(hit_rect(r1, o.z, d.z, o.x, d.x, o.y, d.y, t_min, bound_t, t), t, aux + 1u, link);
            consider(hit_rect(r2, o.y, d.y, o.x, d.x, o.z, d.z, t_min, bound_t, t), t, aux + 2u, link);
            consider(hit_rect(r3, o.y, d.y, o.x, d.x, o.z, d.z, t_min, bound_t, t), t, aux + 3u, link);
            consider(hit_rect(r4, o.x, d.x, o.y, d.y, o.z, d.z, t_min, bound_t, t), t, aux + 4u, link);
            consider(hit_rect(r5, o.x, d.x, o.y, d.y, o.z, d.z, t_min, bound_t, t), t, aux + 5u, link);
        } else if (kind == PT_WK_RECT_XY) {
            consider(hit_rect(g, o.z, d.z, o.x, d.x, o.y, d.y, t_min, bound_t, t), t, aux, link);
        } else if (kind == PT_WK_RECT_XZ) {
            consider(hit_rect(g, o.y, d.y, o.x, d.x, o.z, d.z, t_min, bound_t, t), t, aux, link);
        } else {
            consider(hit_rect(g, o.x, d.x, o.y, d.y, o.z, d.z, t_min, bound_t, t), t, aux, link);
        }
        refresh_bound();
        go(sc, false, 0u, popped, false);
    }
    // The guide traversal is over (pt_device.h, points 1-3): nothing reported at all -> the reference misses too; a tie or near
    // tie -> the literal walk; otherwise the box of P's parent node is fetched for the proof step.
    __device__ __forceinline__ void step_finish(const OrderedScene& sc) {
        const bool miss = w.best_prim < 0 && q[4].y == __builtin_huge_valf();
        const bool literal = !miss && !((double)q[4].y > OrderedWalk::near(w.best_t));
        mode = miss ? (uint32_t)PT_LW_IDLE : (literal ? (uint32_t)PT_LW_LITERAL_BEGIN : (uint32_t)PT_LW_VERIFY);
        aim(sc, sc.guide_bytes + best_parent);
    }
    // bvh.rs:18-35 on the box of P's parent node, operation for operation: OrderedWalk::step_verify
    __device__ __forceinline__ void step_verify(const OrderedBounds& B) {
        const V3 o = w.o, inv = v3(1.0 / w.d.x, 1.0 / w.d.y, 1.0 / w.d.z);
        const double g[6] = {f64_of(r0.z, r0.w), f64_of(r1.x, r1.y), f64_of(r1.z, r1.w), f64_of(r2.x, r2.y), f64_of(r2.z, r2.w), f64_of(r3.x, r3.y)};
        double lo = B.t_min, hi = fmin(OrderedWalk::near(w.best_t), B.t_max);
        double t0 = (g[0] - o.x) * inv.x, t1 = (g[3] - o.x) * inv.x;
        lo = fmax((inv.x < 0.0) ? t1 : t0, lo); hi = fmin((inv.x < 0.0) ? t0 : t1, hi);
        bool miss = hi <= lo;
        t0 = (g[1] - o.y) * inv.y; t1 = (g[4] - o.y) * inv.y;
        lo = fmax((inv.y < 0.0) ? t1 : t0, lo); hi = fmin((inv.y < 0.0) ? t0 : t1, hi);
        miss |= hi <= lo;
        t0 = (g[2] - o.z) * inv.z; t1 = (g[5] - o.z) * inv.z;
        lo = fmax((inv.z < 0.0) ? t1 : t0, lo); hi = fmin((inv.z < 0.0) ? t0 : t1, hi);
        miss |= hi <= lo;
        mode = (miss || r0.x != PT_WK_NODE) ? (uint32_t)PT_LW_LITERAL_BEGIN : (uint32_t)PT_LW_IDLE; // IDLE: the reference reaches P and returns (P, best_t)
    }
    // The literal walk (StreamWalk) from scratch.  A lane that turns to it has no more use for its guide stack: the ray's
    // reciprocals (bvh.rs:22) are parked in its first six slots, once, and read back by every literal step.
    __device__ __forceinline__ void step_literal(const OrderedScene& sc, uint32_t* stack, const OrderedBounds& B) {
        const unsigned char* stream = sc.base + sc.guide_bytes;
        if (mode == PT_LW_LITERAL_BEGIN) {
            const double ix = 1.0 / w.d.x, iy = 1.0 / w.d.y, iz = 1.0 / w.d.z;
            stack[0 * PT_BLOCK] = (uint32_t)__double2loint(ix); stack[1 * PT_BLOCK] = (uint32_t)__double2hiint(ix);
            stack[2 * PT_BLOCK] = (uint32_t)__double2loint(iy); stack[3 * PT_BLOCK] = (uint32_t)__double2hiint(iy);
            stack[4 * PT_BLOCK] = (uint32_t)__double2loint(iz); stack[5 * PT_BLOCK] = (uint32_t)__double2hiint(iz);
            w.best_t = B.t_max;
            w.best_prim = -1;
            sp = 0u;                        // (from here on: the literal walk's offset into the stream)
            aim(sc, sc.guide_bytes);        // StreamWalk::begin: the root record
            mode = PT_LW_LITERAL;
            return;
        }
        // the StreamWalk's record lives in r0..r3 between steps (no registers of its own outside this function)
        w.off = sp;
        w.kind = r0.x & PT_WK_KIND_MASK; w.aux = r0.y;
        w.g[0] = f64_of(r0.z, r0.w); w.g[1] = f64_of(r1.x, r1.y); w.g[2] = f64_of(r1.z, r1.w);
        w.g[3] = f64_of(r2.x, r2.y); w.g[4] = f64_of(r2.z, r2.w); w.g[5] = f64_of(r3.x, r3.y);
        w.t_min = B.t_min;
        w.inv = v3(__hiloint2double((int)stack[1 * PT_BLOCK], (int)stack[0 * PT_BLOCK]), __hiloint2double((int)stack[3 * PT_BLOCK], (int)stack[2 * PT_BLOCK]),
                   __hiloint2double((int)stack[5 * PT_BLOCK], (int)stack[4 * PT_BLOCK]));
        if (w.at_node()) w.advance_node();
        else w.advance_leaf(stream);
        sp = w.off;
        if (w.off >= sc.stream_bytes) mode = PT_LW_IDLE;
        aim(sc, sc.guide_bytes + w.off); // (clamped to the padding when the walk is over)
    }
};
