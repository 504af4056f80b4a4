// pt_device.h -- device-side building blocks of the wavefront path tracer (gfx950).
//
// All geometry runs in f64 with -ffp-contract=off, in the reference's operand order, so that a hit
// (primitive index, t, normal) is the same IEEE-754 result the reference computes (rustc never fuses
// multiply-add; HIP's f64 divide and sqrt expansions are correctly rounded).  Radiometry (throughput,
// emitted radiance, the accumulation buffer) is f32.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../../include/ptgpu.h"

#define PT_WAVE 64
#define PT_BLOCK 256
#define PT_REJECTION_CAP 256

// ------------------------------------------------------------------ f64 3-vector (vec3.rs)
struct V3 {
    double x, y, z;
};
__device__ __forceinline__ V3 v3(double x, double y, double z) { return V3{x, y, z}; }
__device__ __forceinline__ V3 operator+(V3 a, V3 b) { return V3{a.x + b.x, a.y + b.y, a.z + b.z}; }   // vec3.rs:130-136
__device__ __forceinline__ V3 operator-(V3 a, V3 b) { return V3{a.x - b.x, a.y - b.y, a.z - b.z}; }   // :146-152
__device__ __forceinline__ V3 operator*(V3 a, double t) { return V3{a.x * t, a.y * t, a.z * t}; }     // :178-184
__device__ __forceinline__ V3 operator*(double t, V3 a) { return V3{a.x * t, a.y * t, a.z * t}; }     // :194-200
__device__ __forceinline__ V3 operator/(V3 a, double t) { return V3{a.x / t, a.y / t, a.z / t}; }     // :218-224
__device__ __forceinline__ V3 neg(V3 a) { return V3{a.x * -1.0, a.y * -1.0, a.z * -1.0}; }            // :248-254
__device__ __forceinline__ double dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }       // :87-89
__device__ __forceinline__ double len_squared(V3 a) { return a.x * a.x + a.y * a.y + a.z * a.z; }     // :83-85
__device__ __forceinline__ V3 cross(V3 a, V3 b) {                                                      // :91-97
    return V3{a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}
__device__ __forceinline__ V3 unit_vector_ieee(V3 a) { return a / sqrt(len_squared(a)); }              // :99-101
__device__ __forceinline__ V3 reflect(V3 v, V3 n) { return v - (2.0 * dot(v, n)) * n; }                // :114-116
__device__ __forceinline__ V3 refract(V3 v, V3 n, double etai_over_etat) {                              // :118-127
    double cos_theta = fmin(dot(neg(v), n), 1.0);
    V3 perp = etai_over_etat * (v + cos_theta * n);
    V3 parallel = (-sqrt(fabs(1.0 - len_squared(perp)))) * n;
    return parallel + perp;
}
__device__ __forceinline__ bool near_zero(V3 a) {                                                      // :110-112
    return fabs(a.x) < 1e-8 && fabs(a.y) < 1e-8 && fabs(a.z) < 1e-8;
}

// ------------------------------------------------------------------ counter-based RNG
// Philox4x32-10 (Salmon et al., SC'11).  Draw k of stream (pixel, sample, stream) is the 64-bit half k%2 of
// block k/2 = philox(counter = {pixel, sample, stream, k/2}, key = seed).  Replaces the reference's
// thread_rng (renderer.rs:23-24, vec3.rs:27-29,35-37,59-60, material.rs:132).
__device__ __forceinline__ void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1,
                                              uint32_t out[4]) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        // one 32x32->64 product per multiplier (v_mad_u64_u32) instead of separate low and high multiplies
        uint64_t p0 = (uint64_t)c0 * 0xD2511F53ull, p1 = (uint64_t)c2 * 0xCD9E8D57ull;
        uint32_t hi0 = (uint32_t)(p0 >> 32), lo0 = (uint32_t)p0;
        uint32_t hi1 = (uint32_t)(p1 >> 32), lo1 = (uint32_t)p1;
        uint32_t n0 = hi1 ^ c1 ^ k0;
        uint32_t n2 = hi0 ^ c3 ^ k1;
        c0 = n0; c1 = lo1; c2 = n2; c3 = lo0;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

struct SampleRng {
    uint32_t pixel, sample, stream, n, k0, k1;
    uint32_t cached_block;
    uint32_t cache[4];
    __device__ __forceinline__ void init(uint64_t seed, uint32_t pixel_, uint32_t sample_, uint32_t stream_) {
        pixel = pixel_; sample = sample_; stream = stream_; n = 0;
        k0 = (uint32_t)seed; k1 = (uint32_t)(seed >> 32);
        cached_block = 0xFFFFFFFFu;
        cache[0] = cache[1] = cache[2] = cache[3] = 0;
    }
    __device__ __forceinline__ double f64() { // 53 bits in [0,1)
        uint32_t k = n++;
        uint32_t block = k >> 1;
        if (block != cached_block) {
            philox4x32_10(pixel, sample, stream, block, k0, k1, cache);
            cached_block = block;
        }
        uint32_t lo = (k & 1u) ? cache[2] : cache[0];
        uint32_t hi = (k & 1u) ? cache[3] : cache[1];
        uint64_t bits = ((uint64_t)hi << 32) | lo;
        return (double)(bits >> 11) * (1.0 / 9007199254740992.0);
    }
    __device__ __forceinline__ double range(double lo, double hi) { return lo + (hi - lo) * f64(); }
};

__device__ __forceinline__ V3 random_in_unitsphere(SampleRng& rng) { // vec3.rs:41-49
    V3 p = v3(0.0, 0.0, 0.0);
    for (int i = 0; i < PT_REJECTION_CAP; ++i) {
        double x = rng.range(-1.0, 1.0);
        double y = rng.range(-1.0, 1.0);
        double z = rng.range(-1.0, 1.0);
        p = v3(x, y, z);
        if (len_squared(p) < 1.0) break;
    }
    return p;
}
__device__ __forceinline__ V3 random_in_unitdisk_xy(SampleRng& rng) { // vec3.rs:55-67
    V3 p = v3(0.0, 0.0, 0.0);
    for (int i = 0; i < PT_REJECTION_CAP; ++i) {
        double x = rng.range(-1.0, 1.0);
        double y = rng.range(-1.0, 1.0);
        p = v3(x, y, 0.0);
        if (len_squared(p) < 1.0) break;
    }
    return p;
}

// One draw as SampleRng::range(-1.0, 1.0) returns it, from the two words of its 64-bit half: with m = bits >> 11,
// -1 + 2 * (m * 2^-53) is a single rounding of m * 2^-52 - 1, which the FMA below performs (m itself is exact).
__device__ __forceinline__ double draw_pm1(uint32_t lo, uint32_t hi) {
    const uint32_t m_lo = __builtin_amdgcn_alignbit(hi, lo, 11);
    const uint32_t m_hi = hi >> 11;
    const double m = __builtin_fma((double)m_hi, 0x1p32, (double)m_lo);
    return __builtin_fma(m, 0x1p-52, -1.0);
}
// vec3.rs:41-49 on stream (pixel, sample, stream) of the key, from its first draw: the same candidates in the same order
// as random_in_unitsphere(SampleRng&) -- the first accepted one is returned -- but produced by the WAVE, not the lane.
//   * Tries come in pairs: tries 2j and 2j+1 are draws 6j..6j+5, exactly Philox blocks 3j..3j+2.
//   * Round one: every lane that needs a point evaluates its own pair 0; 77 % are done.
//   * Later rounds: the m lanes still waiting publish (pixel, sample) in the wave's 64-slot exchange area; with S the
//     power of two >= m, lane h evaluates pair base + h/S for the waiting lane of rank h%S, so each waiting lane has
//     64/S consecutive pairs tried at once; it then takes the candidate of its first successful helper (lane mask ->
//     strided bit scan -> six ds_bpermute).  A lone-lane tail of ~4 more rounds becomes one.
//   * If more than 32 lanes wait there is nothing to share and every waiting lane evaluates its own next pair.
// The cut-off of the reference's (endless) loop after PT_REJECTION_CAP tries keeps its meaning: the candidate of the last
// try stands.  Must be called by all 64 lanes of the wave together (`need` says which lanes want a point);
// `exchange` = this wave's 64 uint2 of LDS.
__device__ __forceinline__ void unitsphere_pair(uint32_t pixel, uint32_t sample, uint32_t stream, uint32_t pair, uint32_t k0, uint32_t k1,
                                                V3& cand, bool& accepted) {
    uint32_t b0[4], b1[4], b2[4];
    philox4x32_10(pixel, sample, stream, 3u * pair, k0, k1, b0);
    philox4x32_10(pixel, sample, stream, 3u * pair + 1u, k0, k1, b1);
    philox4x32_10(pixel, sample, stream, 3u * pair + 2u, k0, k1, b2);
    const V3 c0 = v3(draw_pm1(b0[0], b0[1]), draw_pm1(b0[2], b0[3]), draw_pm1(b1[0], b1[1]));
    const V3 c1 = v3(draw_pm1(b1[2], b1[3]), draw_pm1(b2[0], b2[1]), draw_pm1(b2[2], b2[3]));
    const bool a0 = len_squared(c0) < 1.0, a1 = len_squared(c1) < 1.0;
    cand = a0 ? c0 : c1; // both rejected: the later candidate
    accepted = a0 | a1;
}
__device__ __forceinline__ double lane_fetch(double v, uint32_t lane_byte_address) {
    const int lo = __builtin_amdgcn_ds_bpermute((int)lane_byte_address, __double2loint(v));
    const int hi = __builtin_amdgcn_ds_bpermute((int)lane_byte_address, __double2hiint(v));
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ V3 random_in_unitsphere_wave(bool need, uint32_t pixel, uint32_t sample, uint32_t stream, uint32_t k0,
                                                        uint32_t k1, uint2* exchange) {
    const uint32_t lane = __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
    V3 p = v3(0.0, 0.0, 0.0);
    unsigned long long waiting = __builtin_amdgcn_ballot_w64(need);
    uint32_t base = 0; // pairs [0, base) have been tried by every waiting lane
    const uint32_t cap_pairs = PT_REJECTION_CAP / 2;
    while (waiting != 0ull && base < cap_pairs) {
        const uint32_t m = (uint32_t)__builtin_popcountll(waiting);
        const bool mine_waits = __builtin_amdgcn_inverse_ballot_w64(waiting);
        if (m > 32u || base + 64u > cap_pairs) { // every waiting lane for itself
            V3 cand;
            bool acc;
            unitsphere_pair(pixel, sample, stream, base, k0, k1, cand, acc);
            if (mine_waits) p = cand;
            waiting &= ~__builtin_amdgcn_ballot_w64(acc);
            base += 1u;
        } else {
            const uint32_t shift = m <= 1u ? 0u : 32u - (uint32_t)__builtin_clz(m - 1u); // S = 1 << shift >= m, S <= 32
            const uint32_t S = 1u << shift;
            const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(waiting >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)waiting, 0u));
            if (mine_waits) exchange[rank] = make_uint2(pixel, sample);
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
            const uint32_t served = lane & (S - 1u);
            const uint2 who = exchange[served];
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
            V3 cand;
            bool acc;
            unitsphere_pair(who.x, who.y, stream, base + (lane >> shift), k0, k1, cand, acc);
            const unsigned long long ok = __builtin_amdgcn_ballot_w64(acc & (served < m));
            // helpers of the lane of rank r sit at lanes r, r + S, r + 2S, ...: the lowest one that succeeded
            unsigned long long every_S = 1ull; // bits 0, S, 2S, ...
            for (uint32_t w = S; w < 64u; w <<= 1) every_S |= every_S << w;
            const unsigned long long mine = mine_waits ? ((ok >> rank) & every_S) : 0ull;
            const uint32_t helper = rank + (uint32_t)__builtin_ctzll(mine | (1ull << 63));
            const uint32_t addr = (helper & 63u) << 2;
            const V3 got = v3(lane_fetch(cand.x, addr), lane_fetch(cand.y, addr), lane_fetch(cand.z, addr));
            if (mine != 0ull) p = got;
            waiting &= ~__builtin_amdgcn_ballot_w64(mine != 0ull);
            base += 64u >> shift;
        }
    }
    return p;
}

// ------------------------------------------------------------------ device scene tables
struct DevNode { // bvh.rs:57-62, padded to 64 B (four 16-B pieces); stored breadth-first so the top levels form a prefix
    double bmin[3];
    double bmax[3];
    uint32_t left, right;
    uint32_t pad[2];
};
static_assert(sizeof(DevNode) == 64, "DevNode must be 64 bytes");
struct DevPrim { // 80 B: what the hit test reads
    // sphere: c[3], r | rect: k, a_lo, a_hi, b_lo, b_hi | triangle: point1[3], point2-point1 [3], point3-point1 [3]
    double d[9];
    uint32_t type;
    uint32_t material;
};
static_assert(sizeof(DevPrim) == 80, "DevPrim must be 80 bytes");
// The wave-uniform walk used for small scenes reads the tree as a byte stream of records in depth-first pre-order
// (what the reference's recursion visits, bvh.rs:107-121), each {u32 kind, u32 aux, f64 payload[]}:
//   PT_WK_NODE      payload box min xyz, max xyz (6)       aux = byte offset of the first record after the subtree
//   PT_WK_RECT_XY/XZ/YZ  payload k, a_lo, a_hi, b_lo, b_hi (5)   aux = primitive index
//   PT_WK_BOX6      payload box min xyz, max xyz (6)       aux = index of the first of the six rectangles of an AABox
//                   (geometry.rs:391-446 builds the sides from exactly these six numbers; the upload verifies it)
//   PT_WK_SPHERE    payload center, radius (4)             aux = primitive index
//   PT_WK_TRIANGLE  payload point1, edge1, edge2 (9)       aux = primitive index
enum { PT_WK_NODE = 0, PT_WK_RECT_XY = 1, PT_WK_RECT_XZ = 2, PT_WK_RECT_YZ = 3, PT_WK_BOX6 = 4, PT_WK_SPHERE = 5, PT_WK_TRIANGLE = 6 };
// Streams built for the ordered walk also carry, in a primitive record's kind word above PT_WK_LINK_SHIFT, the offset
// (in 8-byte words) of the node record that is its parent in the reference's tree.
#define PT_WK_KIND_MASK 7u
#define PT_WK_LINK_SHIFT 3
#define PT_WK_NODE_BYTES 56u
#define PT_WK_RECT_BYTES 48u
#define PT_WK_BOX6_BYTES 56u
#define PT_WK_SPHERE_BYTES 40u
#define PT_WK_TRIANGLE_BYTES 80u
#define PT_WALK_MAX_ITEMS 192 /* nodes + primitives: the stream stays within the scalar cache */
#define PT_WALK_PLANES_IN_RANGE 1u /* walk_flags: see walk_division_is_proven */
#define PT_WALK_BOXES_NESTED 2u    /* every node's box is finite and contains its child nodes' boxes (OrderedWalk) */
enum { PT_PROC_NONE = 0, PT_PROC_CHECKER = 1, PT_PROC_MARBLE = 2 }; // (marble: with or without checkers)
enum { PT_TRACE_LOCKSTEP = 0, PT_TRACE_STACK = 1, PT_TRACE_STREAM = 2, PT_TRACE_ORDERED = 3 };
// Everything shade needs about the primitive it hit, in ONE 80-byte record (one dependent fetch instead of the
// chain primitive -> material -> texture): the geometry the normal comes from, the material, and the colour
// when the material's texture is a solid one.
struct ShadePrim {
    double g[4];        // sphere: center, radius | rectangle, triangle: outward / stored normal, 0
    double param;       // metal fuzz / dielectric index of refraction
    uint32_t prim_type; // PT_PRIM_*
    uint32_t mat_type;  // PT_MAT_*
    uint32_t texture;   // texture index (procedural lookups)
    uint32_t tex_solid; // 1: `color` is the texture's value everywhere; 2: a checker of two solid colours, `color` where the
                        // sine product is not negative and `color_odd` where it is (texture.rs:36-45); 0: look the texture up
    float color[3];     // (f32 of the f64 colour: what the f32 radiometry multiplies by)
    float color_odd[3];
};
static_assert(sizeof(ShadePrim) == 80, "ShadePrim must be 80 bytes");
#define PT_SHADE_LDS_MAX_PRIMS 192

struct DevScene {
    const DevNode* nodes;
    const DevPrim* prims;
    const double* aux; // 3 per primitive: outward normal of a rectangle / stored normal of a triangle
    const PtMaterial* materials;
    const PtTexture* textures;
    const uint8_t* perlin;
    uint32_t n_perlin; // 256-byte permutation tables
    uint32_t n_nodes, n_prims, n_materials, n_textures;
    uint32_t root;
    uint32_t lds_nodes; // nodes [0, lds_nodes) are staged in LDS by the trace kernel
    uint32_t lds_prims; // primitives [0, lds_prims) likewise (all of them for small scenes)
    uint32_t stack_depth;
    const unsigned char* walk; // the record stream (PT_WK_*), padded by 64 bytes
    uint32_t walk_bytes;
    uint32_t trace_mode; // PT_TRACE_*: which kernel walks it
    uint32_t walk_flags; // PT_WALK_*
    uint32_t has_procedural; // PT_PROC_*: which textures the scene has beyond solid colours
    const ShadePrim* shade_prims;
    const unsigned char* guide; // the guide tree of the ordered walk (guide_tree.h), node 0 = root, followed by a copy of the record
                                // stream (one allocation: OrderedScene); null when there is none
    uint32_t guide_nodes;
    uint32_t deep_tree; // PT_TAIL_MIN_PRIMS primitives or more: the tail launch by default, and the wave loop's thresholds for thin waves (pt_ordered.h)
    // which rays the ordered walk may take (computed at upload from the primitives' sizes, see pt_scene_upload):
    double ow_origin_limit; // |o|inf at most this ...
    double ow_tri_extent;   // ... and (|o|inf + ow_tri_extent) * |d|inf at most ow_od_limit
    double ow_od_limit;
};

// ------------------------------------------------------------------ hit tests (t only; the record is rebuilt in shade)
// Sphere::hit, geometry.rs:132-148
__device__ __forceinline__ bool hit_sphere(const double* g, V3 o, V3 d, double t_min, double t_max, double& t_out) {
    V3 oc = o - v3(g[0], g[1], g[2]);
    double a = len_squared(d);
    double half_b = dot(d, oc);
    double c = len_squared(oc) - g[3] * g[3];
    double dis = half_b * half_b - a * c;
    if (dis < 0.0) return false;
    double sq = sqrt(dis);
    double root = (-half_b - sq) / a;
    if (root < t_min || root > t_max) {
        root = (-half_b + sq) / a;
        if (root < t_min || root > t_max) return false;
    }
    t_out = root;
    return true;
}
// RectangleXY/XZ/YZ::hit, geometry.rs:210-221, :273-284, :336-347.  (on, dn) = ray origin/direction along the
// plane normal, (oa, da), (ob, db) along the two in-plane axes in the reference's order.
__device__ __forceinline__ bool hit_rect(const double* g, double on, double dn, double oa, double da, double ob, double db,
                                         double t_min, double t_max, double& t_out) {
    double t = (g[0] - on) / dn;
    if (t < t_min || t > t_max) return false;
    double a = oa + t * da;
    double b = ob + t * db;
    if (a < g[1] || a > g[2] || b < g[3] || b > g[4]) return false;
    t_out = t;
    return true;
}
// Triangle::hit, geometry.rs:519-546 (Moller-Trumbore, back-face culled at det < 0.0001)
__device__ __forceinline__ bool hit_triangle(const double* g, V3 o, V3 d, double t_min, double t_max, double& t_out) {
    V3 v0v1 = v3(g[3], g[4], g[5]);
    V3 v0v2 = v3(g[6], g[7], g[8]);
    V3 pvec = cross(d, v0v2);
    double det = dot(v0v1, pvec);
    if (det < 0.0001) return false;
    double inv_det = 1.0 / det;
    V3 tvec = o - v3(g[0], g[1], g[2]);
    double u = dot(tvec, pvec) * inv_det;
    if (!(u >= 0.0 && u <= 1.0)) return false;
    V3 qvec = cross(tvec, v0v1);
    double v = dot(d, qvec) * inv_det;
    if (v < 0.0 || u + v > 1.0) return false;
    double t = dot(v0v2, qvec) * inv_det;
    if (!(t >= t_min && t <= t_max)) return false;
    t_out = t;
    return true;
}
__device__ __forceinline__ bool hit_prim(const double* g, uint32_t type, V3 o, V3 d, double t_min, double t_max, double& t_out) {
    switch (type) {
        case PT_PRIM_SPHERE: return hit_sphere(g, o, d, t_min, t_max, t_out);
        case PT_PRIM_RECT_XY: return hit_rect(g, o.z, d.z, o.x, d.x, o.y, d.y, t_min, t_max, t_out);
        case PT_PRIM_RECT_XZ: return hit_rect(g, o.y, d.y, o.x, d.x, o.z, d.z, t_min, t_max, t_out);
        case PT_PRIM_RECT_YZ: return hit_rect(g, o.x, d.x, o.y, d.y, o.z, d.z, t_min, t_max, t_out);
        default: return hit_triangle(g, o, d, t_min, t_max, t_out);
    }
}

// Aabb::hit, bvh.rs:18-35.  inv = 1/direction is the same value the reference recomputes at every node.
__device__ __forceinline__ bool hit_aabb(const double* bmin, const double* bmax, V3 o, V3 inv, double t_min, double t_max) {
    double t0 = (bmin[0] - o.x) * inv.x, t1 = (bmax[0] - o.x) * inv.x;
    if (inv.x < 0.0) { double s = t0; t0 = t1; t1 = s; }
    t_min = fmax(t0, t_min);
    t_max = fmin(t1, t_max);
    if (t_max <= t_min) return false;
    t0 = (bmin[1] - o.y) * inv.y; t1 = (bmax[1] - o.y) * inv.y;
    if (inv.y < 0.0) { double s = t0; t0 = t1; t1 = s; }
    t_min = fmax(t0, t_min);
    t_max = fmin(t1, t_max);
    if (t_max <= t_min) return false;
    t0 = (bmin[2] - o.z) * inv.z; t1 = (bmax[2] - o.z) * inv.z;
    if (inv.z < 0.0) { double s = t0; t0 = t1; t1 = s; }
    t_min = fmax(t0, t_min);
    t_max = fmin(t1, t_max);
    if (t_max <= t_min) return false;
    return true;
}

// ------------------------------------------------------------------ BvhNode::hit (bvh.rs:107-121), iteratively
// The reference recursion (left with t_max, right with t_max shrunk to the left hit, right wins) is a
// depth-first, left-before-right walk in which every primitive is tested against the closest t found so
// far and accepted when t <= that bound: the last accepted primitive is the answer.  The walk below keeps
// that order exactly; pending right children wait on a per-lane stack in LDS (stack[level * blockDim + tid]).
// lds_nodes / lds_prims point at the staged prefixes of the node and primitive tables.
// ALL_LDS: the whole scene is staged (small scenes); otherwise only nodes [0, sc.lds_nodes) are.
template <bool ALL_LDS>
__device__ __forceinline__ void traverse(const DevScene& sc, const DevNode* lds_nodes, const DevPrim* lds_prims,
                                         uint32_t* stack, V3 o, V3 d, double t_min, double t_max, double& best_t,
                                         int32_t& best_prim) {
    V3 inv = v3(1.0 / d.x, 1.0 / d.y, 1.0 / d.z);
    best_t = t_max;
    best_prim = -1;
    uint32_t sp = 0;
    uint32_t cur = sc.root;
    const uint32_t stride = blockDim.x;
    for (;;) {
        if (cur & PT_REF_LEAF) {
            uint32_t first = cur & PT_REF_INDEX_MASK;
            uint32_t count = (cur & PT_REF_GROUP6) ? 6u : 1u;
            for (uint32_t k = 0; k < count; ++k) { // geometry.rs:75-80 for a group
                uint32_t p = first + k;
                double t;
                bool h;
                if (ALL_LDS) {
                    const DevPrim* pr = lds_prims + p;
                    h = hit_prim(pr->d, pr->type, o, d, t_min, best_t, t);
                } else {
                    const DevPrim* pr = sc.prims + p;
                    h = hit_prim(pr->d, pr->type, o, d, t_min, best_t, t);
                }
                if (h) {
                    best_t = t;
                    best_prim = (int32_t)p;
                }
            }
            if (sp == 0) break;
            cur = stack[(--sp) * stride];
        } else {
            double bmin[3], bmax[3];
            uint32_t left, right;
            if (ALL_LDS || cur < sc.lds_nodes) {
                const DevNode* nd = lds_nodes + cur;
                bmin[0] = nd->bmin[0]; bmin[1] = nd->bmin[1]; bmin[2] = nd->bmin[2];
                bmax[0] = nd->bmax[0]; bmax[1] = nd->bmax[1]; bmax[2] = nd->bmax[2];
                left = nd->left; right = nd->right;
            } else {
                const DevNode* nd = sc.nodes + cur;
                bmin[0] = nd->bmin[0]; bmin[1] = nd->bmin[1]; bmin[2] = nd->bmin[2];
                bmax[0] = nd->bmax[0]; bmax[1] = nd->bmax[1]; bmax[2] = nd->bmax[2];
                left = nd->left; right = nd->right;
            }
            if (hit_aabb(bmin, bmax, o, inv, t_min, best_t)) {
                stack[sp * stride] = right;
                ++sp;
                cur = left;
            } else {
                if (sp == 0) break;
                cur = stack[(--sp) * stride];
            }
        }
    }
}

// ------------------------------------------------------------------ the same walk, wave-uniform (small scenes)
// All 64 lanes step through the pre-order item list together: the item index is a scalar, the item's box or
// primitive parameters arrive by scalar loads and feed the VALU as SGPR operands, and the primitive-type
// dispatch is a scalar branch -- no divergent control flow at all.  A lane that failed a node's box test
// (bvh.rs:108-110) sits out until the walk leaves that subtree (`skip_until`); when no lane is left inside a
// subtree the whole wave jumps over it.  Per lane this performs exactly the sequence of box and primitive
// tests, with the same running t_max, as the stack walk above and as the reference recursion.
// Straight-line (predicated) forms of the tests for the lockstep walk: every comparison of the reference is kept
// as written (so NaNs fall the same way), only the early returns are folded together.  Each comparison goes
// straight into a 64-bit lane mask in SGPRs (one v_cmp each), the masks are combined by the scalar unit, and a
// mask comes back as a per-lane predicate only where a lane-wise select needs it -- no VGPR round trips.
typedef unsigned long long LaneMask;
__device__ __forceinline__ LaneMask lanes_where(bool p) { return __builtin_amdgcn_ballot_w64(p); }
__device__ __forceinline__ bool lane_in(LaneMask m) { return __builtin_amdgcn_inverse_ballot_w64(m); }
__device__ __forceinline__ bool wave_any(bool p) { return lanes_where(p) != 0ull; }

// lanes whose ray misses the box (bvh.rs:18-35)
__device__ __forceinline__ LaneMask walk_aabb_miss(const double* b, V3 o, V3 inv, double t_min, double t_max) {
    double t0 = (b[0] - o.x) * inv.x, t1 = (b[3] - o.x) * inv.x;
    double lo = (inv.x < 0.0) ? t1 : t0, hi = (inv.x < 0.0) ? t0 : t1;
    t_min = fmax(lo, t_min);
    t_max = fmin(hi, t_max);
    LaneMask miss = lanes_where(t_max <= t_min);
    t0 = (b[1] - o.y) * inv.y; t1 = (b[4] - o.y) * inv.y;
    lo = (inv.y < 0.0) ? t1 : t0; hi = (inv.y < 0.0) ? t0 : t1;
    t_min = fmax(lo, t_min);
    t_max = fmin(hi, t_max);
    miss |= lanes_where(t_max <= t_min);
    t0 = (b[2] - o.z) * inv.z; t1 = (b[5] - o.z) * inv.z;
    lo = (inv.z < 0.0) ? t1 : t0; hi = (inv.z < 0.0) ? t0 : t1;
    t_min = fmax(lo, t_min);
    t_max = fmin(hi, t_max);
    miss |= lanes_where(t_max <= t_min);
    return miss;
}

// ---- n / d from the ray's reciprocal.
// Given y = 1.0 / d correctly rounded, q0 = n*y, r = fma(-q0, d, n) (the exact residual) and q = fma(r, y, q0) is
// n / d correctly rounded (Markstein's theorem): mul, fma, fma at full rate instead of the 11-instruction, ~58-cycle
// IEEE sequence (profiles/r01_instruction_rates.txt).  The proof needs y and q0 to be normal numbers and the residual
// not to underflow; n = +0 also works (every step gives the correctly signed zero), and it is the common special
// case: a ray leaving an axis-aligned rectangle starts exactly on its plane.
//   div_given_reciprocal  checks those conditions per quotient and lets the wave vote: any lane outside them sends the
//                         whole wave through the IEEE division.  Result = n / d bit for bit for ALL inputs.
//   div_proven            the three instructions alone, for callers that have established the conditions up front
//                         (traverse_walk does so once per ray batch, see walk_division_is_proven).
// Tests: pt_selftest_division against pt_selftest_arith; every bit-exact hit comparison, including rays and planes of
// extreme magnitude (tests/test_gpu_parity.py).
__device__ __forceinline__ bool reciprocal_in_range(double y) { return fabs(y) >= 0x1p-150 && fabs(y) <= 0x1p150; }
__device__ __forceinline__ double reciprocal_for_division(double y) {
    return reciprocal_in_range(y) ? y : __builtin_nan("");
}
// lanes of the wave whose x is in the IEEE classes of `classes` (v_cmp_class bit set), as a scalar mask
__device__ __forceinline__ LaneMask wave_class_mask(double x, uint32_t classes) {
    LaneMask m;
    asm("v_cmp_class_f64 %0, %1, %2" : "=s"(m) : "v"(x), "s"(classes));
    return m;
}
__device__ __forceinline__ double div_proven(double n, double d, double y) {
    double q = n * y;
    double r = __builtin_fma(-q, d, n);
    return __builtin_fma(r, y, q);
}
__device__ __forceinline__ double div_ieee_unspeculated(double n, double d) {
    asm volatile("; IEEE division" : "+v"(n)); // pins the division inside its (wave-uniform) branch: never hoisted
    return n / d;
}
__device__ __forceinline__ double div_given_reciprocal(double n, double d, double y) {
    y = reciprocal_for_division(y);
    const double q = n * y;
    const LaneMask q_ok = wave_class_mask(q, 0x168u);   // +-normal, +-0
    const LaneMask n_zero = wave_class_mask(n, 0x40u);  // +0
    const LaneMask n_big = lanes_where(fabs(n) > 0x1p-900);
    if (__builtin_expect((q_ok & (n_big | n_zero)) != lanes_where(true), 0)) return div_ieee_unspeculated(n, d);
    return div_proven(n, d, y);
}
// The conditions of div_proven for every rectangle quotient (k - o_c) / d_c of a ray, established once per ray:
//   scene (checked at upload, PT_WALK_PLANES_IN_RANGE): every rectangle plane k is 0 or 2^-800 <= |k| <= 2^400, and every
//          node box is finite with min <= max (for walk_aabb_miss_finite below);
//   ray: each origin component is 0 or in that same range, each reciprocal 1/d_c is in [2^-150, 2^150].
// Then n = k - o_c is +-0 or 2^-853 <= |n| <= 2^401 (a nonzero difference of two such numbers is at least half an ulp of
// the larger), q0 = n*y is 0 or in [2^-1003, 2^551]: normal, and the residual, a multiple of ulp(q0)*ulp(d) >= 2^-957,
// is exact.  (n = -0 needs k = -0, o_c = +0; the quotient may then come out as +0 instead of -0, which no comparison
// and no consumer of t distinguishes.)
__device__ __forceinline__ bool coordinate_in_range(double c) { return c == 0.0 || (fabs(c) >= 0x1p-800 && fabs(c) <= 0x1p400); }
__device__ __forceinline__ bool walk_division_is_proven(V3 o, V3 inv) {
    const bool ok = reciprocal_in_range(inv.x) && reciprocal_in_range(inv.y) && reciprocal_in_range(inv.z) &&
                    coordinate_in_range(o.x) && coordinate_in_range(o.y) && coordinate_in_range(o.z);
    return lanes_where(ok) == lanes_where(true);
}

// vec3.rs:99-101, a / |a| component by component, with ONE division: y = 1 / |a|, then each quotient by div_proven.
// Its conditions, voted by the wave: 2^-1000 <= |a| <= 2^200 (y is a normal number, no quotient underflows) and every
// component at least 2^-800 in magnitude (exact residuals; |component / length| <= 1 rules out overflow).  Any lane
// outside -- including an exactly zero component -- sends the wave through the three IEEE divisions.
__device__ __forceinline__ V3 unit_vector(V3 a) {
    const double len = sqrt(len_squared(a));
    const double smallest = fmin(fmin(fabs(a.x), fabs(a.y)), fabs(a.z));
    const LaneMask ok = lanes_where(smallest >= 0x1p-800) & lanes_where(len <= 0x1p200) & lanes_where(len >= 0x1p-1000);
    if (__builtin_expect(ok != lanes_where(true), 0))
        return v3(div_ieee_unspeculated(a.x, len), div_ieee_unspeculated(a.y, len), div_ieee_unspeculated(a.z, len));
    const double y = 1.0 / len;
    return v3(div_proven(a.x, len, y), div_proven(a.y, len, y), div_proven(a.z, len, y));
}

// The box test when no product can be NaN (finite nonzero reciprocals, finite box with min <= max -- both established
// before this variant is chosen): (min - o)*inv and (max - o)*inv are then ordered by the sign of inv, so the swap of
// bvh.rs:25-27 is min/max of the pair; and since the interval only ever shrinks, "empty after some axis" (the early
// returns of bvh.rs:30-32) is "empty after the last axis": one comparison instead of three.
__device__ __forceinline__ LaneMask walk_aabb_miss_finite(const double* b, V3 o, V3 inv, double t_min, double t_max) {
    double t0 = (b[0] - o.x) * inv.x, t1 = (b[3] - o.x) * inv.x;
    t_min = fmax(fmin(t0, t1), t_min);
    t_max = fmin(fmax(t0, t1), t_max);
    t0 = (b[1] - o.y) * inv.y; t1 = (b[4] - o.y) * inv.y;
    t_min = fmax(fmin(t0, t1), t_min);
    t_max = fmin(fmax(t0, t1), t_max);
    t0 = (b[2] - o.z) * inv.z; t1 = (b[5] - o.z) * inv.z;
    t_min = fmax(fmin(t0, t1), t_min);
    t_max = fmin(fmax(t0, t1), t_max);
    return lanes_where(t_max <= t_min);
}

// One rectangle (geometry.rs:210-221 and the XZ / YZ twins) for the lanes of `act`: plane k on axis n, extent
// [a_lo, a_hi] x [b_lo, b_hi] on the other two.  The six rejection tests of the reference narrow EXEC one v_cmpx at a
// time (a lane survives "not (t < t_min)" etc., so NaNs fall as in the reference), the survivors take the hit with
// three moves, and EXEC is put back: no mask arithmetic on the scalar unit, no selects.
// The wave runs the walk with all 64 lanes enabled, so "back" is all ones.
template <bool PROVEN>
__device__ __forceinline__ void walk_rect(LaneMask act, double k, double a_lo, double a_hi, double b_lo, double b_hi, double on,
                                          double dn, double inv_dn, double oa, double da, double ob, double db, double t_min,
                                          uint32_t prim, double& best_t, int32_t& best_prim) {
    const double n = k - on;
    const double t = PROVEN ? div_proven(n, dn, inv_dn) : div_ieee_unspeculated(n, dn);
    const double a = oa + t * da;
    const double b = ob + t * db;
    asm volatile(
        "s_mov_b64 exec, %[act]\n\t"
        "v_cmpx_nlt_f64 vcc, %[t], %[tmin]\n\t"
        "v_cmpx_ngt_f64 vcc, %[t], %[best]\n\t"
        "v_cmpx_nlt_f64 vcc, %[a], %[alo]\n\t"
        "v_cmpx_ngt_f64 vcc, %[a], %[ahi]\n\t"
        "v_cmpx_nlt_f64 vcc, %[b], %[blo]\n\t"
        "v_cmpx_ngt_f64 vcc, %[b], %[bhi]\n\t"
        "v_mov_b64 %[best], %[t]\n\t"
        "v_mov_b32 %[bp], %[prim]\n\t"
        "s_mov_b64 exec, -1"
        : [best] "+v"(best_t), [bp] "+v"(best_prim)
        : [act] "s"(act), [t] "v"(t), [a] "v"(a), [b] "v"(b), [tmin] "s"(t_min), [alo] "s"(a_lo), [ahi] "s"(a_hi), [blo] "s"(b_lo),
          [bhi] "s"(b_hi), [prim] "s"(prim)
        : "vcc");
}

__device__ __forceinline__ LaneMask walk_sphere(const double* g, LaneMask act, V3 o, V3 d, double t_min, double t_max, double& t_out) {
    V3 oc = o - v3(g[0], g[1], g[2]); // geometry.rs:132-148
    double a = len_squared(d);
    double half_b = dot(d, oc);
    double c = len_squared(oc) - g[3] * g[3];
    double dis = half_b * half_b - a * c;
    const LaneMask cand = act & ~lanes_where(dis < 0.0);
    LaneMask hit = 0;
    if (cand) { // wave-uniform: skip the sqrt and divides when the whole wave misses
        double sq = sqrt(dis);
        double root = (-half_b - sq) / a;
        const LaneMask out1 = lanes_where(root < t_min) | lanes_where(root > t_max);
        hit = cand & ~out1;
        if (cand & out1) {
            double root2 = (-half_b + sq) / a;
            const LaneMask out2 = lanes_where(root2 < t_min) | lanes_where(root2 > t_max);
            if (lane_in(out1)) root = root2;
            hit = cand & ~(out1 & out2);
        }
        t_out = root;
    }
    return hit;
}
__device__ __forceinline__ LaneMask walk_triangle(const double* g, LaneMask act, V3 o, V3 d, double t_min, double t_max, double& t_out) {
    V3 v0v1 = v3(g[3], g[4], g[5]); // geometry.rs:519-546
    V3 v0v2 = v3(g[6], g[7], g[8]);
    V3 pvec = cross(d, v0v2);
    double det = dot(v0v1, pvec);
    const LaneMask cand = act & ~lanes_where(det < 0.0001);
    LaneMask hit = 0;
    if (cand) {
        double inv_det = 1.0 / det;
        V3 tvec = o - v3(g[0], g[1], g[2]);
        double u = dot(tvec, pvec) * inv_det;
        LaneMask ok = cand & lanes_where(u >= 0.0) & lanes_where(u <= 1.0);
        V3 qvec = cross(tvec, v0v1);
        double v = dot(d, qvec) * inv_det;
        ok &= ~(lanes_where(v < 0.0) | lanes_where(u + v > 1.0));
        double t = dot(v0v2, qvec) * inv_det;
        ok &= lanes_where(t >= t_min) & lanes_where(t <= t_max);
        t_out = t;
        hit = ok;
    }
    return hit;
}

// PROVEN: the conditions of walk_division_is_proven hold for this wave's rays and this scene.
template <bool PROVEN>
__device__ __forceinline__ void traverse_walk_loop(const unsigned char* __restrict__ walk, uint32_t walk_bytes, bool lane_live, V3 o,
                                                   V3 d, V3 inv, double t_min, double& best_t, int32_t& best_prim) {
    uint32_t skip_until = lane_live ? 0u : 0xFFFFFFFFu; // byte offset from which this lane takes part again
    uint32_t off = 0;
    while (off < walk_bytes) {
        off = __builtin_amdgcn_readfirstlane(off);
        // one definition point per iteration for the per-lane state: keeps each in a single register across the
        // dispatch below instead of a web of copies at every join
        asm volatile("" : "+v"(best_t), "+v"(best_prim), "+v"(skip_until));
        const double* gw = reinterpret_cast<const double*>(walk + off + 8u);
        const uint32_t* hdr = reinterpret_cast<const uint32_t*>(walk + off); // scalar fetches: the record feeds the VALU from SGPRs
        const uint32_t kind = hdr[0];
        const uint32_t aux = hdr[1];
        // header and the first 48 bytes of payload in one go (the stream is padded): ONE scalar-cache latency per record instead of
        // header -> branch on the kind -> payload.  Worth little where it was measured -- three alternating rounds on two boxes said
        // -1.1 % on the Cornell frame, six rounds on a third 135.5 against 135.8 ms: the box's drift is larger -- but never slower.  (A
        // triangle's last three doubles follow inside its branch.  Fetching the NEXT record one iteration ahead as well was slower,
        // 137.7 against 134.1 ms: the second register set's moves cost more than the latency hidden; profiles/r03_walk_record_fetch_ab.txt.)
        const double g[6] = {gw[0], gw[1], gw[2], gw[3], gw[4], gw[5]};
        asm volatile("" ::"s"(kind), "s"(g[0]), "s"(g[1]), "s"(g[2]), "s"(g[3]), "s"(g[4]), "s"(g[5]));
        const LaneMask act = lanes_where(skip_until <= off);
        if (kind == PT_WK_NODE) {
            const LaneMask miss = PROVEN ? walk_aabb_miss_finite(g, o, inv, t_min, best_t) : walk_aabb_miss(g, o, inv, t_min, best_t);
            if (lane_in(act & miss)) skip_until = aux; // bvh.rs:108-110: this lane sits the subtree out
            off = (act & ~miss) ? off + PT_WK_NODE_BYTES : aux;
        } else if (kind == PT_WK_BOX6) { // geometry.rs:70-83 over the six sides, in list order
            if (act) {
                const double x0 = g[0], y0 = g[1], z0 = g[2], x1 = g[3], y1 = g[4], z1 = g[5];
                walk_rect<PROVEN>(act, z0, x0, x1, y0, y1, o.z, d.z, inv.z, o.x, d.x, o.y, d.y, t_min, aux, best_t, best_prim);
                walk_rect<PROVEN>(act, z1, x0, x1, y0, y1, o.z, d.z, inv.z, o.x, d.x, o.y, d.y, t_min, aux + 1u, best_t, best_prim);
                walk_rect<PROVEN>(act, y0, x0, x1, z0, z1, o.y, d.y, inv.y, o.x, d.x, o.z, d.z, t_min, aux + 2u, best_t, best_prim);
                walk_rect<PROVEN>(act, y1, x0, x1, z0, z1, o.y, d.y, inv.y, o.x, d.x, o.z, d.z, t_min, aux + 3u, best_t, best_prim);
                walk_rect<PROVEN>(act, x0, y0, y1, z0, z1, o.x, d.x, inv.x, o.y, d.y, o.z, d.z, t_min, aux + 4u, best_t, best_prim);
                walk_rect<PROVEN>(act, x1, y0, y1, z0, z1, o.x, d.x, inv.x, o.y, d.y, o.z, d.z, t_min, aux + 5u, best_t, best_prim);
            }
            off += PT_WK_BOX6_BYTES;
        } else if (kind == PT_WK_RECT_XY) {
            if (act) walk_rect<PROVEN>(act, g[0], g[1], g[2], g[3], g[4], o.z, d.z, inv.z, o.x, d.x, o.y, d.y, t_min, aux, best_t, best_prim);
            off += PT_WK_RECT_BYTES;
        } else if (kind == PT_WK_RECT_XZ) {
            if (act) walk_rect<PROVEN>(act, g[0], g[1], g[2], g[3], g[4], o.y, d.y, inv.y, o.x, d.x, o.z, d.z, t_min, aux, best_t, best_prim);
            off += PT_WK_RECT_BYTES;
        } else if (kind == PT_WK_RECT_YZ) {
            if (act) walk_rect<PROVEN>(act, g[0], g[1], g[2], g[3], g[4], o.x, d.x, inv.x, o.y, d.y, o.z, d.z, t_min, aux, best_t, best_prim);
            off += PT_WK_RECT_BYTES;
        } else {
            if (act) {
                double t = 0.0;
                const LaneMask h = (kind == PT_WK_SPHERE) ? walk_sphere(g, act, o, d, t_min, best_t, t) : walk_triangle(gw, act, o, d, t_min, best_t, t);
                if (lane_in(h)) {
                    best_t = t;
                    best_prim = (int32_t)aux;
                }
            }
            off += (kind == PT_WK_SPHERE) ? PT_WK_SPHERE_BYTES : PT_WK_TRIANGLE_BYTES;
        }
    }
}

// ------------------------------------------------------------------ the same record stream, one lane per ray
// Larger scenes: every lane walks the pre-order stream on its own.  A node whose box the ray misses is left through its
// skip offset, anything else by stepping to the next record -- the depth-first, left-before-right order of
// bvh.rs:107-121 without a stack: no per-lane stack in LDS, no child references, one 56-byte fetch per step (header +
// 48 bytes of payload, issued together; only a triangle needs a second one).  The walk is a small state machine
// (StreamWalk) so that the render kernel can hand a lane its next ray the moment it finishes one.
struct StreamWalk {
    V3 o, d, inv;
    double t_min, best_t;
    int32_t best_prim;
    uint32_t off;      // byte offset of the current record
    uint32_t kind, aux; // its header
    double g[6];       // and the first 48 bytes of its payload
#ifdef PT_DEBUG_BOUNDS
    uint32_t lim = 0xFFFFFFFFu;
#endif
#ifdef PT_DEBUG_STEPS
    uint32_t steps_nodes, steps_leaves;
#endif
    // (`limit`: callers that want a fetch outside the stream to be impossible whatever the stream holds pass its size)
    __device__ __forceinline__ void fetch(const unsigned char* __restrict__ stream, uint32_t limit = 0xFFFFFFFFu) {
#ifdef PT_DEBUG_BOUNDS
        if (off > lim) {
            printf("StreamWalk: off %u past the end %u (previous kind %u aux %u)\n", off, lim, kind, aux);
            off = lim;
        }
#endif
        off = min(off, limit);
        const uint2 hdr = *reinterpret_cast<const uint2*>(stream + off);
        const double2* q = reinterpret_cast<const double2*>(stream + off + 8u);
        const double2 a = q[0], b = q[1], c = q[2]; // within the stream's padding for the last record (and past the end: zeros)
        kind = hdr.x & PT_WK_KIND_MASK; aux = hdr.y; // (the rest of the word: PT_WK_LINK_SHIFT)
        g[0] = a.x; g[1] = a.y; g[2] = b.x; g[3] = b.y; g[4] = c.x; g[5] = c.y;
    }
    __device__ __forceinline__ void begin(const unsigned char* __restrict__ stream, V3 o_, V3 d_, double t_min_, double t_max_) {
        o = o_; d = d_;
        inv = v3(1.0 / d.x, 1.0 / d.y, 1.0 / d.z);
        t_min = t_min_; best_t = t_max_;
        best_prim = -1;
        off = 0;
#ifdef PT_DEBUG_STEPS
        steps_nodes = steps_leaves = 0;
#endif
        fetch(stream);
    }
    __device__ __forceinline__ bool at_node() const { return kind == PT_WK_NODE; }
    __device__ __forceinline__ void accept(bool hit, double t, uint32_t prim) {
        if (hit) { best_t = t; best_prim = (int32_t)prim; }
    }
    // the current record is a node: bvh.rs:108-110 (straight-line box test: no divergent early returns)
    __device__ __forceinline__ void step_node(const unsigned char* __restrict__ stream, uint32_t limit = 0xFFFFFFFFu) {
        advance_node();
        fetch(stream, limit);
    }
    // (the step without its fetch: for callers that keep the record elsewhere, pt_lean.h)
    __device__ __forceinline__ void advance_node() {
#ifdef PT_DEBUG_STEPS
        steps_nodes += 1;
#endif
        double lo = t_min, hi = best_t;
        double t0 = (g[0] - o.x) * inv.x, t1 = (g[3] - o.x) * inv.x;
        lo = fmax((inv.x < 0.0) ? t1 : t0, lo); hi = fmin((inv.x < 0.0) ? t0 : t1, hi);
        bool miss = hi <= lo;
        t0 = (g[1] - o.y) * inv.y; t1 = (g[4] - o.y) * inv.y;
        lo = fmax((inv.y < 0.0) ? t1 : t0, lo); hi = fmin((inv.y < 0.0) ? t0 : t1, hi);
        miss |= hi <= lo;
        t0 = (g[2] - o.z) * inv.z; t1 = (g[5] - o.z) * inv.z;
        lo = fmax((inv.z < 0.0) ? t1 : t0, lo); hi = fmin((inv.z < 0.0) ? t0 : t1, hi);
        miss |= hi <= lo;
        off = miss ? aux : off + PT_WK_NODE_BYTES;
    }
    // the current record is a primitive (or the six sides of a box)
    __device__ __forceinline__ void step_leaf(const unsigned char* __restrict__ stream, uint32_t limit = 0xFFFFFFFFu) {
        advance_leaf(stream);
        fetch(stream, limit);
    }
    __device__ __forceinline__ void advance_leaf(const unsigned char* __restrict__ stream) {
#ifdef PT_DEBUG_STEPS
        steps_leaves += 1;
#endif
        double t;
        if (kind == PT_WK_SPHERE) {
            accept(hit_sphere(g, o, d, t_min, best_t, t), t, aux);
            off += PT_WK_SPHERE_BYTES;
        } else if (kind == PT_WK_TRIANGLE) {
            const double* rest = reinterpret_cast<const double*>(stream + off + 56u);
            const double tri[9] = {g[0], g[1], g[2], g[3], g[4], g[5], rest[0], rest[1], rest[2]};
            accept(hit_triangle(tri, o, d, t_min, best_t, t), t, aux);
            off += PT_WK_TRIANGLE_BYTES;
        } else if (kind == PT_WK_BOX6) { // geometry.rs:70-83 over the six sides, in list order
            const double r0[5] = {g[2], g[0], g[3], g[1], g[4]}, r1[5] = {g[5], g[0], g[3], g[1], g[4]};
            const double r2[5] = {g[1], g[0], g[3], g[2], g[5]}, r3[5] = {g[4], g[0], g[3], g[2], g[5]};
            const double r4[5] = {g[0], g[1], g[4], g[2], g[5]}, r5[5] = {g[3], g[1], g[4], g[2], g[5]};
            accept(hit_rect(r0, o.z, d.z, o.x, d.x, o.y, d.y, t_min, best_t, t), t, aux);
            accept(hit_rect(r1, o.z, d.z, o.x, d.x, o.y, d.y, t_min, best_t, t), t, aux + 1u);
            accept(hit_rect(r2, o.y, d.y, o.x, d.x, o.z, d.z, t_min, best_t, t), t, aux + 2u);
            accept(hit_rect(r3, o.y, d.y, o.x, d.x, o.z, d.z, t_min, best_t, t), t, aux + 3u);
            accept(hit_rect(r4, o.x, d.x, o.y, d.y, o.z, d.z, t_min, best_t, t), t, aux + 4u);
            accept(hit_rect(r5, o.x, d.x, o.y, d.y, o.z, d.z, t_min, best_t, t), t, aux + 5u);
            off += PT_WK_BOX6_BYTES;
        } else {
            bool h;
            if (kind == PT_WK_RECT_XY) h = hit_rect(g, o.z, d.z, o.x, d.x, o.y, d.y, t_min, best_t, t);
            else if (kind == PT_WK_RECT_XZ) h = hit_rect(g, o.y, d.y, o.x, d.x, o.z, d.z, t_min, best_t, t);
            else h = hit_rect(g, o.x, d.x, o.y, d.y, o.z, d.z, t_min, best_t, t);
            accept(h, t, aux);
            off += PT_WK_RECT_BYTES;
        }
    }
    __device__ __forceinline__ double result_t() const {
#ifdef PT_DEBUG_STEPS
        return (double)steps_nodes + 1e-4 * (double)steps_leaves; // debugging aid: step counts instead of t
#else
        return best_t;
#endif
    }
};

// ------------------------------------------------------------------ the ordered walk (larger scenes)
// The literal walk visits ~40 records per ray in SphereField and ~180 per ray that reaches a million-triangle mesh: the
// reference's tree (random axis, median split, always left first) culls poorly.  Its RESULT, though, is almost always
// "the nearest primitive"; the tree only matters when two primitives are equally near, or when one of its boxes rejects
// a ray that still hits something inside (rounding at a box face).  The ordered walk separates the two questions:
//   1. GUIDE/LEAF: a nearest-first traversal of the guide tree (guide_tree.h: surface-area splits, children's boxes in
//      the parent, padded far beyond rounding) with the SAME primitive tests finds the nearest hit (P, t*) and
//      `second_t`, the nearest among all the other candidates; subtrees are culled against near(best) = best * (1+1e-9),
//      so every primitive whose test reports t <= near(t*) has been tested.  No hit at all => the reference misses too
//      (it tests a subset of the primitives with the same arithmetic).
//   2. second_t <= near(t*) (a tie or a near tie: the order of the reference's walk may decide) => LITERAL.
//   3. VERIFY: otherwise repeat the reference's slab test (bvh.rs:18-35) on the box of P's parent node in the
//      reference's tree, with the bound b' = min(near(t*), t_max).  (a) The reference's own bound, whenever it tests an
//      ancestor of P, is the caller's t_max or a hit other than P, all of which are farther than near(t*); the slab
//      test is monotone in its bound, so passing with b' means passing there.  (b) The boxes are nested (a node's box
//      is the union of its children's, bvh.rs:37-46; checked at upload: PT_WALK_BOXES_NESTED) and every operation of
//      the slab test is monotone -- (min - o) * inv rounds monotonically, fmax/fmin are monotone, no NaN can arise
//      for a finite ray with nonzero direction and finite boxes -- so entry(parent) <= entry(child) and exit(parent)
//      >= exit(child) hold IN floating point: a ray that passes a node's box with bound b' passes every ancestor's.
//      So the reference reaches P, accepts it (t* <= its bound), and nothing it tests afterwards is accepted (no
//      other t <= t*): it returns (P, t*).  A failed box => LITERAL.
//   4. LITERAL: the per-lane literal walk (StreamWalk) from scratch -- also for rays with a zero or non-finite
//      component and when the guide stack overflows.
// Correctness never rests on the guide tree being good, only on its boxes containing their primitives.
#define PT_GUIDE_LEAF 0x80000000u
#ifndef PT_GUIDE_STACK /* (a build with a small stack, -DPT_GUIDE_STACK=6, sends many rays through the overflow -> literal-walk path: a test aid) */
#define PT_GUIDE_STACK 20 /* 20 KB of LDS per block: six blocks of the tail kernel (path.hip) fit a CU; deeper => literal walk */
#endif
#define PT_GUIDE_STACK_SLOTS (PT_GUIDE_STACK + 1) /* levels the kernels allocate: one to spare for pt_lean.h's unconditional store */
#define PT_GUIDE_ABS_SLACK 1e-37f /* see the box test's error budget at OrderedWalk below */
// Wave-uniform: what the ordered walk reads.  The guide tree's allocation holds the nodes (64 bytes each, node 0 = root) and,
// right behind them, a copy of the record stream with parent links (pt_scene_upload), so that everything the walk fetches is
// `base` + a 32-bit offset.
struct OrderedScene {
    const unsigned char* base;
    uint32_t guide_bytes;        // guide nodes * 64: where the stream copy starts
    uint32_t stream_bytes;       // of the stream proper (128 bytes of padding follow)
    uint32_t fetch_limit;        // guide_bytes + stream_bytes: the last offset a fetch may start at
    double origin_limit, tri_extent, od_limit; // DevScene::ow_*
};
__host__ __device__ inline OrderedScene ordered_scene(const DevScene& sc) {
    return OrderedScene{sc.guide, sc.guide_nodes * 64u, sc.walk_bytes, sc.guide_nodes * 64u + sc.walk_bytes, sc.ow_origin_limit,
                        sc.ow_tri_extent, sc.ow_od_limit};
}
#ifdef PT_DEBUG_PHASES /* debugging aid (pt_ordered.h): counters summed over all waves, read with pt_debug_read_phases */
__device__ unsigned long long pt_debug_phase[32];
#endif
typedef float pt_f32x2 __attribute__((ext_vector_type(2)));
// What is the same for every ray of a kernel (wave-uniform: it lives in scalar registers, not once per lane).
struct OrderedBounds {
    double t_min, t_max; // the caller's window (ray.rs:25: 0.001, +inf)
    float tmin_dn;       // t_min moved down for the guide tree's f32 box tests
};
// Arithmetic shared by the steps of the ordered walk (the walk itself: pt_lean.h LeanWalk).
struct OrderedWalk {
    static __device__ __forceinline__ double near(double b) { return fma(fabs(b), 1e-9, b); }
    static __device__ __forceinline__ float f32_not_below(double x) { // >= x (x finite or +inf), up to PT_GUIDE_ABS_SLACK where the conversion underflows
        const float f = (float)x;
        return __builtin_fmaf(__builtin_fabsf(f), 0x1p-23f, f);
    }
    static __device__ __forceinline__ float f32_not_above(double x) {
        const float f = (float)x;
        return __builtin_fmaf(__builtin_fabsf(f), -0x1p-23f, f);
    }
    static __device__ __forceinline__ OrderedBounds bounds(double t_min, double t_max) {
        return OrderedBounds{t_min, t_max, f32_not_above(t_min) - 2.0f * PT_GUIDE_ABS_SLACK};
    }
    // ---- The guide tree's box test, in packed f32.  It only has to be CONSERVATIVE: accept every child box (stored as f32,
    // rounded outwards, padded by 1e-5 of the leaf's largest coordinate m) that the ray crosses within [t_min, bound]; what
    // it accepts beyond that costs time, never correctness.  A node stores a child's box as (lo, hi) pairs per axis, so one
    // packed subtraction and one packed fused multiply-add give both slab parameters of an axis:
    //     a = fma(l - of, invf, -(o - of) * invf)      with u = 2^-24:
    //   l - of       rounds by u of itself; |l - of| <= |l - o| + u |o|;
    //   invf         is 1/d to 3u (begin); the residual o - of is exact in f64, its f32 rounding and the product with invf
    //                add 2u of the residual; the fma rounds once more by u;
    //   together     a = (l - o)/d * (1 + 5u) + |1/d| * 3 u^2 |o|: a relative error of 3e-7 and a shift of the plane by
    //                1.1e-14 |o|, far inside the padding (begin() admits |o| <= 5e8 m only), plus 2^-149 where a product
    //                underflows.  No product overflows (the upload refuses a guide tree for coordinates beyond 1e18, begin()
    //                admits |1/d| <= 2^60 only): no inf, no NaN.  min / max are exact.
    // Entry and exit are therefore moved outwards by 5e-7 of themselves, the bound (near(best) rounded to f32) up and t_min
    // down by 2^-23 of themselves, and PT_GUIDE_ABS_SLACK covers every underflow (of a product, or of a conversion to f32).
    // (Every VALU instruction, f32 or f64, takes a SIMD four cycles per wave; only the packed forms do two lanes' worth of
    // f32 work in one: 12 packed + ~30 plain instructions for the two boxes of a step, against ~80 in f64.)
    // (Written as instructions because the compiler materialises a broadcast operand of a packed operation as a register
    // pair: nine constants of the ray would take eighteen registers.  Here they sit in five pairs -- (of.x, of.y), (of.z,
    // invf.x), (invf.y, invf.z), (nrinv.x, nrinv.y), (nrinv.z, -) -- and op_sel picks the half both lanes of the operation read.)
#define PT_PK_SLAB(out, planes, c_of, sel_of, c_inv, sel_inv, c_nr, sel_nr)                                                          \
    asm("v_pk_add_f32 %0, %1, %2 op_sel:[0," #sel_of "] op_sel_hi:[1," #sel_of "] neg_lo:[0,1] neg_hi:[0,1]\n\t"                    \
        "v_pk_fma_f32 %0, %0, %3, %4 op_sel:[0," #sel_inv "," #sel_nr "] op_sel_hi:[1," #sel_inv "," #sel_nr "]"                       \
        : "=&v"(out)                                                                                                                 \
        : "v"(planes), "v"(c_of), "v"(c_inv), "v"(c_nr))
    static __device__ __forceinline__ float raw_min(float a, float b) { float r; asm("v_min_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
    static __device__ __forceinline__ float raw_max(float a, float b) { float r; asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
};

__device__ __forceinline__ void traverse_walk(const unsigned char* __restrict__ walk, uint32_t walk_bytes, uint32_t walk_flags,
                                              bool lane_live, V3 o, V3 d, double t_min, double t_max, double& best_t,
                                              int32_t& best_prim) {
    V3 inv = v3(1.0 / d.x, 1.0 / d.y, 1.0 / d.z);
    best_t = t_max;
    best_prim = -1;
    if ((walk_flags & PT_WALK_PLANES_IN_RANGE) && walk_division_is_proven(o, inv))
        traverse_walk_loop<true>(walk, walk_bytes, lane_live, o, d, inv, t_min, best_t, best_prim);
    else
        traverse_walk_loop<false>(walk, walk_bytes, lane_live, o, d, inv, t_min, best_t, best_prim);
}

// Outward normal of primitive `prim` at point p: geometry.rs:151 (sphere), :227/:290/:353 (rectangles),
// :554 (triangle: the stored normal).
__device__ __forceinline__ V3 outward_normal(const DevScene& sc, const DevPrim* pr, uint32_t prim, V3 p) {
    if (pr->type == PT_PRIM_SPHERE) return (p - v3(pr->d[0], pr->d[1], pr->d[2])) / pr->d[3];
    const double* a = sc.aux + (size_t)prim * 3;
    return v3(a[0], a[1], a[2]);
}

// ------------------------------------------------------------------ textures (texture.rs)
// noise 0.8.2 Perlin stand-in: improved 3-D Perlin over a 256-entry permutation.
// The 16 gradient cases of the noise crate's table -- 0|12: x+y, 1|13: -x+y, 2: x-y, 3: -x-y, 4: x+z, 5: -x+z,
// 6: x-z, 7: -x-z, 8: y+z, 9|14: -y+z, 10: y-z, 11|15: -y-z -- without a 16-way branch: fold 12..15 onto
// 0, 1, 9, 11, then case h is (+-a) + (+-b) with (a, b) = (x,y), (x,z), (y,z) for h>>2 = 0, 1, 2.
// (-a) + (-b) has the same bits as -a - b, and a + (-b) the same as a - b.
__device__ __forceinline__ double perlin_grad(int h, double x, double y, double z) {
    const unsigned long long fold = 0xB910BA9876543210ull;
    const unsigned hh = (unsigned)(fold >> ((h & 15) * 4)) & 15u;
    const unsigned group = hh >> 2;
    double a = (group == 2u) ? y : x;
    double b = (group == 0u) ? y : z;
    a = (hh & 1u) ? -a : a;
    b = (hh & 2u) ? -b : b;
    return a + b;
}
__device__ __forceinline__ int perlin_hash(const uint8_t* perm, long x, long y, long z) {
    int a = perm[x & 0xff];
    int b = perm[a ^ (int)(y & 0xff)];
    return perm[b ^ (int)(z & 0xff)];
}
__device__ __forceinline__ double perlin_quintic(double t) { return t * t * t * (t * (t * 6.0 - 15.0) + 10.0); }
__device__ __forceinline__ double perlin_get(const uint8_t* perm, double px, double py, double pz) {
    double fx = floor(px), fy = floor(py), fz = floor(pz);
    long cx = (long)fx, cy = (long)fy, cz = (long)fz;
    double dx = px - fx, dy = py - fy, dz = pz - fz;
    double g000 = perlin_grad(perlin_hash(perm, cx, cy, cz), dx, dy, dz);
    double g100 = perlin_grad(perlin_hash(perm, cx + 1, cy, cz), dx - 1.0, dy, dz);
    double g010 = perlin_grad(perlin_hash(perm, cx, cy + 1, cz), dx, dy - 1.0, dz);
    double g110 = perlin_grad(perlin_hash(perm, cx + 1, cy + 1, cz), dx - 1.0, dy - 1.0, dz);
    double g001 = perlin_grad(perlin_hash(perm, cx, cy, cz + 1), dx, dy, dz - 1.0);
    double g101 = perlin_grad(perlin_hash(perm, cx + 1, cy, cz + 1), dx - 1.0, dy, dz - 1.0);
    double g011 = perlin_grad(perlin_hash(perm, cx, cy + 1, cz + 1), dx, dy - 1.0, dz - 1.0);
    double g111 = perlin_grad(perlin_hash(perm, cx + 1, cy + 1, cz + 1), dx - 1.0, dy - 1.0, dz - 1.0);
    double u = perlin_quintic(dx), v = perlin_quintic(dy), w = perlin_quintic(dz);
    double k0 = g000;
    double k1 = g100 - g000;
    double k2 = g010 - g000;
    double k3 = g001 - g000;
    double k4 = g000 + g110 - g100 - g010;
    double k5 = g000 + g101 - g100 - g001;
    double k6 = g000 + g011 - g010 - g001;
    double k7 = g100 + g010 + g001 + g111 - g000 - g110 - g101 - g011;
    double r = k0 + k1 * u + k2 * v + k3 * w + k4 * u * v + k5 * u * w + k6 * v * w + k7 * u * v * w;
    r = r * 1.1547005383792515;
    if (r < -1.0) r = -1.0;
    if (r > 1.0) r = 1.0;
    return r;
}
// ---- The same value, with about half the instructions (marble shading is seven of these per hit).
//   * lattice coordinates by the native f64 -> i32 conversion (the caller has checked |floor| < 2^31; x & 255 is the same);
//   * the permutation look-ups shared between corners: 2 + 4 + 8 instead of 24;
//   * a gradient is s.x * x + s.y * y + s.z * z with (s.x, s.y, s.z) from a 16-entry table of -1 / 0 / +1 (one of them 0):
//     two fused multiply-adds and a product whose single rounding is that of (+-a) + (+-b) above -- the same bits;
//   * the trilinear blend as seven nested lerps a + w (b - a) instead of the expanded polynomial: 14 operations instead
//     of 41.  THIS changes the rounding (by ~1e-16 of a value of order one; the oracle keeps the polynomial): texture values
//     are compared with a tolerance (they scale the f32 throughput and never steer a path), see DESIGN.md section 1.
#define PT_PERLIN_GRAD_DOUBLES 48
__device__ const double kPerlinGradTable[PT_PERLIN_GRAD_DOUBLES] = {
    1, 1, 0,  -1, 1, 0,  1, -1, 0,  -1, -1, 0,  1, 0, 1,  -1, 0, 1,  1, 0, -1,  -1, 0, -1,
    0, 1, 1,  0, -1, 1,  0, 1, -1,  0, -1, -1,  1, 1, 0,  -1, 1, 0,  0, -1, 1,  0, -1, -1};
__device__ __forceinline__ double perlin_grad_dot(const double* __restrict__ grad, int h, double x, double y, double z) {
    const double* s = grad + (h & 15) * 3;
    return __builtin_fma(s[0], x, __builtin_fma(s[1], y, s[2] * z));
}
__device__ __forceinline__ bool perlin_fast_applies(double px, double py, double pz) {
    return fabs(px) < 2147483000.0 && fabs(py) < 2147483000.0 && fabs(pz) < 2147483000.0;
}
__device__ __forceinline__ double perlin_get_fast(const uint8_t* __restrict__ perm, const double* __restrict__ grad, double px, double py, double pz) {
    const double fx = floor(px), fy = floor(py), fz = floor(pz);
    const int cx = (int)fx, cy = (int)fy, cz = (int)fz;
    const double x0 = px - fx, y0 = py - fy, z0 = pz - fz;
    const double x1 = x0 - 1.0, y1 = y0 - 1.0, z1 = z0 - 1.0;
    const int ya = cy & 255, yb = (cy + 1) & 255, za = cz & 255, zb = (cz + 1) & 255;
    const int a0 = perm[cx & 255], a1 = perm[(cx + 1) & 255];
    const int b00 = perm[a0 ^ ya], b10 = perm[a1 ^ ya], b01 = perm[a0 ^ yb], b11 = perm[a1 ^ yb];
    const double g000 = perlin_grad_dot(grad, perm[b00 ^ za], x0, y0, z0), g100 = perlin_grad_dot(grad, perm[b10 ^ za], x1, y0, z0);
    const double g010 = perlin_grad_dot(grad, perm[b01 ^ za], x0, y1, z0), g110 = perlin_grad_dot(grad, perm[b11 ^ za], x1, y1, z0);
    const double g001 = perlin_grad_dot(grad, perm[b00 ^ zb], x0, y0, z1), g101 = perlin_grad_dot(grad, perm[b10 ^ zb], x1, y0, z1);
    const double g011 = perlin_grad_dot(grad, perm[b01 ^ zb], x0, y1, z1), g111 = perlin_grad_dot(grad, perm[b11 ^ zb], x1, y1, z1);
    const double u = perlin_quintic(x0), v = perlin_quintic(y0), w = perlin_quintic(z0);
    const double e00 = __builtin_fma(u, g100 - g000, g000), e10 = __builtin_fma(u, g110 - g010, g010);
    const double e01 = __builtin_fma(u, g101 - g001, g001), e11 = __builtin_fma(u, g111 - g011, g011);
    const double f0 = __builtin_fma(v, e10 - e00, e00), f1 = __builtin_fma(v, e11 - e01, e01);
    double r = __builtin_fma(w, f1 - f0, f0);
    r = r * 1.1547005383792515;
    if (r < -1.0) r = -1.0;
    if (r > 1.0) r = 1.0;
    return r;
}
// Texture::value (texture.rs:19-23, :36-45, :75-80); u,v are unused by every texture the reference ships.
#ifndef PT_TEXTURE_CALL
// The sign of sin(a) without evaluating the sine: with k = rint(a / pi), a - k pi lies in about [-pi/2, pi/2] and
// sin(a) = (-1)^k sin(a - k pi), so the sign is that of the reduced argument, flipped for odd k.  The reduction uses pi as
// the sum of two doubles (106 bits) and two fused multiply-adds; for |a| <= 2^22 its error is below 1e-25, while no double
// in that range comes closer than ~1e-19 to a multiple of pi: the sign is exact, i.e. the one any correct sin() returns
// (the oracle's libm, the device's).  Outside [1e-100, 2^22] (zero, underflowing products, huge arguments) the caller
// evaluates the sines.
__device__ __forceinline__ bool sine_sign_is_cheap(double a) { return fabs(a) >= 1e-100 && fabs(a) <= 0x1p22; }
__device__ __forceinline__ bool sine_is_negative(double a) {
    const double k = rint(a * 0x1.45f306dc9c883p-2); // 1/pi
    double r = __builtin_fma(-k, 0x1.921fb54442d18p+1, a);  // pi, rounded
    r = __builtin_fma(-k, 0x1.1a62633145c07p-53, r);        // pi - the line above
    const bool odd = ((int)k & 1) != 0;
    return (r < 0.0) != odd;
}
// CheckerTexture::value's choice (texture.rs:36-45): sin(10 x) sin(10 y) sin(10 z) < 0 selects `odd`.  Only the SIGN of the
// product is used: it comes from the reduced arguments unless a factor is zero, tiny or huge -- then the sines are evaluated
// (behind a barrier the compiler cannot hoist them over: they are 40 times the work of the common path).
__device__ __forceinline__ bool checker_is_odd(V3 p) {
    const double ax = 10.0 * p.x, ay = 10.0 * p.y, az = 10.0 * p.z;
    if (sine_sign_is_cheap(ax) && sine_sign_is_cheap(ay) && sine_sign_is_cheap(az))
        return sine_is_negative(ax) != sine_is_negative(ay) != sine_is_negative(az); // no factor is zero, none underflows
    double bx = ax, by = ay, bz = az;
    asm volatile("; checker: the sines themselves" : "+v"(bx), "+v"(by), "+v"(bz));
    return sin(bx) * sin(by) * sin(bz) < 0.0;
}
#define PT_TEXTURE_CALL __forceinline__
#endif
template <bool MARBLE>
__device__ PT_TEXTURE_CALL V3 texture_value_procedural(const PtTexture* textures, uint32_t n_textures, const uint8_t* perlin,
                                                    const double* grad, uint32_t tex, V3 p) {
    for (uint32_t guard = 0; guard <= n_textures; ++guard) {
        const PtTexture* t = textures + tex;
        if (t->type == PT_TEX_CHECKER) {
            tex = checker_is_odd(p) ? t->odd : t->even;
            continue;
        }
        if (t->type == PT_TEX_SOLID) return v3(t->color[0], t->color[1], t->color[2]);
        if (!MARBLE) return v3(0.0, 0.0, 0.0); // (the upload says there is none: the kernel carries neither the code nor its registers)
        // marble: texture.rs:60-80, 7 octaves
        const uint8_t* perm = perlin + (size_t)t->perlin_table * 256;
        double accumulator = 0.0, weight = 1.0;
        double qx = p.x, qy = p.y, qz = p.z;
        // (the points of the seven octaves are p, 2p, ..., 64p: if the last one is within the fast path's range, all are)
        if (wave_any(!perlin_fast_applies(64.0 * qx, 64.0 * qy, 64.0 * qz))) {
            for (int i = 0; i < 7; ++i) {
                accumulator += weight * perlin_get(perm, qx, qy, qz);
                weight *= 0.5;
                qx *= 2.0; qy *= 2.0; qz *= 2.0;
            }
        } else {
#pragma unroll 1
            for (int i = 0; i < 7; ++i) {
                accumulator += weight * perlin_get_fast(perm, grad, qx, qy, qz);
                weight *= 0.5;
                qx *= 2.0; qy *= 2.0; qz *= 2.0;
            }
        }
        double turb = fabs(accumulator);
        double c = 0.5 * (1.0 - sin(t->scale * p.z + 10.0 * turb));
        return v3(1.0 * c, 1.0 * c, 1.0 * c);
    }
    return v3(0.0, 0.0, 0.0);
}
// Solid colours (every material of the Cornell box) are answered inline; checker and marble have ONE call site per kernel
// (shade looks a texture up once per hit), inlined so that the permutation tables can be read from LDS.
// PROCEDURAL = false is for scenes whose textures are all solid: the kernel then carries neither the call nor the
// callee's register budget.
template <bool PROCEDURAL>
__device__ __forceinline__ V3 texture_value(const DevScene& sc, uint32_t tex, V3 p) {
    const PtTexture* t = sc.textures + tex;
    if (!PROCEDURAL || t->type == PT_TEX_SOLID) return v3(t->color[0], t->color[1], t->color[2]);
    return texture_value_procedural<true>(sc.textures, sc.n_textures, sc.perlin, kPerlinGradTable, tex, p);
}

// ------------------------------------------------------------------ materials (material.rs)
// f64::powi(5) as compiler-rt's __powidf2 evaluates it: x * ((x*x)*(x*x)).
__device__ __forceinline__ double reflectance(double cosine, double refraction_index) { // material.rs:109-113
    double q = (1.0 - refraction_index) / (1.0 + refraction_index);
    double r0 = q * q;
    double m = 1.0 - cosine;
    double m2 = m * m;
    double m4 = m2 * m2;
    return r0 + (1.0 - r0) * (m * m4);
}

// Material::scatter (material.rs), split where the unit-sphere sample is needed so that the sampler can be run by the
// whole wave (random_in_unitsphere_wave):  scatter_begin decides per lane what happens -- Dielectric scatters at once
// (one draw of its own), DiffuseLight and an absorbed Metal reflection end the path, Lambertian and a surviving Metal
// reflection ask for a point (`need`) and leave the vector it will be added to in `base` -- and scatter_finish turns
// the point into the scattered direction.  n is the hit normal already flipped to oppose the ray; the random draws come
// from stream (pixel, sample, stream) of the key (k0, k1).
enum { PT_SCATTER_NONE = 0, PT_SCATTER_DONE = 1, PT_SCATTER_NEED_POINT = 2 };
__device__ __forceinline__ int scatter_begin(uint32_t mat_type, double mat_param, V3 ray_d, V3 n, bool front_face, uint32_t pixel,
                                             uint32_t sample, uint32_t stream, uint32_t k0, uint32_t k1, V3& out_d, V3& base) {
    if (mat_type == PT_MAT_LAMBERTIAN) { // material.rs:41-55
        base = n;
        return PT_SCATTER_NEED_POINT;
    }
    if (mat_type == PT_MAT_METAL) { // material.rs:77-93
        base = reflect(unit_vector(ray_d), n);
        return dot(base, n) > 0.0 ? PT_SCATTER_NEED_POINT : PT_SCATTER_NONE;
    }
    if (mat_type == PT_MAT_DIELECTRIC) { // material.rs:117-143
        double refraction_ratio = front_face ? 1.0 / mat_param : mat_param;
        V3 unit_direction = unit_vector(ray_d);
        double cos_theta = fmin(dot(neg(unit_direction), n), 1.0);
        double sin_theta = sqrt(1.0 - cos_theta * cos_theta);
        bool reflects = (refraction_ratio * sin_theta > 1.0);
        if (!reflects) {
            uint32_t b[4];
            philox4x32_10(pixel, sample, stream, 0u, k0, k1, b); // the stream's first draw (rng.gen::<f64>(), material.rs:132)
            const uint64_t bits = ((uint64_t)b[1] << 32) | b[0];
            reflects = reflectance(cos_theta, refraction_ratio) > (double)(bits >> 11) * (1.0 / 9007199254740992.0);
        }
        out_d = reflects ? reflect(unit_direction, n) : refract(unit_direction, n, refraction_ratio);
        return PT_SCATTER_DONE;
    }
    return PT_SCATTER_NONE; // DiffuseLight never scatters (material.rs:16-18)
}
__device__ __forceinline__ V3 scatter_finish(uint32_t mat_type, double mat_param, V3 base, V3 point) {
    if (mat_type == PT_MAT_METAL) return base + mat_param * point; // material.rs:84: reflected + fuzz * random_in_unit_sphere
    V3 dir = base + unit_vector(point);                              // material.rs:47: normal + random_unit_vector
    if (near_zero(dir)) dir = base;
    return dir;
}

// ------------------------------------------------------------------ stream layouts
// Rays live in HBM as three double2 streams (16 B per lane per access): (ox,oy) (oz,dx) (dy,dz).
struct RayStream {
    double2* a;
    double2* b;
    double2* c;
};
// Path state: x = path slot within the pass (sample_slot * npix + pixel), yzw = throughput as f32 bits.
// Hits: t (f64) and primitive (i32, -1 = miss).
