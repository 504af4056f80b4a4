// pt_ordered.h -- the wave loop of the ordered walk (pt_device.h "the ordered walk"; the per-lane steps: pt_lean.h), shared by
// the trace kernels (trace.hip) and the per-wave tail of the bounce loop (path.hip).
#pragma once
#include "pt_launch.h"
#include "pt_lean.h"

// Thresholds of the loop.  Measured on SphereField (1080p, 32 spp) in round 1: handing out rays when 40 lanes are idle, primitive
// tests when 16 lanes wait for one, four box tests per round: 26.9 ms of trace per frame (16 / 24 / 4, the stream walk's
// values: 29.7 ms); swept again in round 3 on the lean steps (profiles/r03_lean_walk_thresholds.txt) and on the aim + load loop
// (r03_wave_loop_thresholds_deep.txt): SphereField within 1.5 % everywhere; the million-triangle mesh is another matter (below).
#ifndef PT_ORD_REFILL_LANES
#define PT_ORD_REFILL_LANES 40
#endif
#ifndef PT_ORD_NODE_STEPS
#define PT_ORD_NODE_STEPS 4
#endif
#ifndef PT_ORD_LEAF_LANES
#define PT_ORD_LEAF_LANES 16
#endif
#ifndef PT_LW_FINISH_LANES
#define PT_LW_FINISH_LANES 12 /* end-of-traversal decisions and proof steps run when this many lanes wait for one */
#endif
// Scenes with deep trees (a million triangles: most rays leave after a few steps, a few take hundreds, so a wave is thin most of the
// time): rays are handed out earlier, and the two waiting thresholds are FRACTIONS (of 64) of the lanes that hold a ray, not lane counts:
// 33.0 -> 30.1 ms per 64-spp frame of the mesh (SphereField on these: 83.9 -> 87.2 ms, so they are picked per scene: DevScene::deep_tree,
// and used by every tail launch, whose waves are thin by construction).  Compile-time on purpose: as run-time kernel arguments the
// same thresholds made the register allocator spill 113 values.
#ifndef PT_ORD_DEEP_REFILL_LANES
#define PT_ORD_DEEP_REFILL_LANES 32
#endif
#ifndef PT_ORD_DEEP_LEAF_LANES
#define PT_ORD_DEEP_LEAF_LANES 16
#endif
#ifndef PT_ORD_DEEP_FINISH_LANES
#define PT_ORD_DEEP_FINISH_LANES 12
#endif
#ifdef PT_DEBUG_PHASES /* debugging aid: rounds and active lanes per kind of step, summed over all waves (read with pt_debug_read_phases) */
#ifdef PT_DEBUG_PHASES_TIME_ONLY /* (the counters' atomics would be most of what the clocks see) */
#define PT_PHASE_COUNT(slot, active_mask) do {} while (0)
#else
#define PT_PHASE_COUNT(slot, active_mask)                                                                   \
    do {                                                                                                    \
        const unsigned long long mask_ = (active_mask); /* (the ballot: all lanes take part) */             \
        if ((threadIdx.x & 63u) == 0u) {                                                                    \
            atomicAdd(&pt_debug_phase[2 * (slot)], 1ull);                                                   \
            atomicAdd(&pt_debug_phase[2 * (slot) + 1], (unsigned long long)__builtin_popcountll(mask_));   \
        }                                                                                                   \
    } while (0)
#endif
#define PT_PHASE_CLOCK(var) const unsigned long long var = __builtin_readcyclecounter()
#define PT_PHASE_TIME(slot, from, to)                                                                      \
    do { /* summed per wave in registers, one atomic per part when the wave is done (PT_PHASE_FLUSH) */    \
        const unsigned long long d_ = (to) - (from);                                                       \
        const int s_ = (slot);                                                                             \
        pt0 += s_ == 0 ? d_ : 0ull; pt1 += s_ == 1 ? d_ : 0ull; pt2 += s_ == 2 ? d_ : 0ull; pt3 += s_ == 3 ? d_ : 0ull; \
        pt4 += s_ == 4 ? d_ : 0ull; pt5 += s_ == 5 ? d_ : 0ull; pt6 += s_ == 6 ? d_ : 0ull;               \
    } while (0)
#define PT_PHASE_DECL unsigned long long pt0 = 0, pt1 = 0, pt2 = 0, pt3 = 0, pt4 = 0, pt5 = 0, pt6 = 0
#define PT_PHASE_FLUSH                                                                                     \
    do {                                                                                                   \
        if ((threadIdx.x & 63u) == 0u) {                                                                   \
            atomicAdd(&pt_debug_phase[24], pt0); atomicAdd(&pt_debug_phase[25], pt1); atomicAdd(&pt_debug_phase[26], pt2); \
            atomicAdd(&pt_debug_phase[27], pt3); atomicAdd(&pt_debug_phase[28], pt4); atomicAdd(&pt_debug_phase[29], pt5); \
            atomicAdd(&pt_debug_phase[30], pt6);                                                           \
        }                                                                                                  \
    } while (0)
#else
#define PT_PHASE_COUNT(slot, active_mask) do {} while (0)
#define PT_PHASE_CLOCK(var) do {} while (0)
#define PT_PHASE_TIME(slot, from, to) do {} while (0)
#define PT_PHASE_DECL do {} while (0)
#define PT_PHASE_FLUSH do {} while (0)
#endif
// Larger scenes, ordered: nearest-first over the guide tree, then the proof that the reference's walk agrees (pt_device.h).  Same
// refill scheme as trace_stream_kernel; a lane's pending far children wait on a stack in LDS laid out [level][thread].  The wave
// runs ONE kind of step per round -- hand out rays, box tests (PT_ORD_NODE_STEPS in a row), primitive tests, end-of-traversal
// decisions + proof steps, or the literal walk -- chosen from how many lanes want which; every step is followed by the loads of the
// lane's next record, so lanes that sit out a round are waiting for memory anyway.
// `io` supplies ray k of this wave's n rays and takes its result: the render kernel's region of the ray stream, or a slice of the
// caller's arrays in the parity entry (the same loop, so the parity tests exercise the refill and the phases).
// UNROLL_STEPS: the box steps of a round as straight-line code (the trace kernels: SphereField 32 spp 25.6 against 26.3 ms) or as a
// loop (the tail kernel, which also carries the shading code: fewer spilled values, 1M mesh 16 spp tail 7.8 -> 6.8 ms).
template <bool UNROLL_STEPS, bool DEEP, class RayIo>
__device__ __forceinline__ void lean_wave_loop(const OrderedScene& sc, uint32_t* stack, uint32_t n, double t_min, double t_max, RayIo& io) {
    uint32_t next = 0; // first ray not yet handed to a lane (wave-uniform)
    uint32_t mine = 0;
    const OrderedBounds B = OrderedWalk::bounds(t_min, t_max);
    LeanWalk w;
    w.rt = make_uint4(0u, 0u, 0u, 0u);
    w.begin(sc, B, v3(0, 0, 0), v3(1, 1, 1));
    w.mode = PT_LW_IDLE;
    PT_PHASE_DECL;
    for (;;) {
        PT_PHASE_CLOCK(c_top);
        const unsigned long long idle = __builtin_amdgcn_ballot_w64(w.mode == PT_LW_IDLE);
        const uint32_t n_idle = (uint32_t)__builtin_popcountll(idle);
        if (next < n && n_idle >= (DEEP ? PT_ORD_DEEP_REFILL_LANES : PT_ORD_REFILL_LANES)) {
            const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(idle >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)idle, 0u));
            const uint32_t k = next + rank;
            PT_PHASE_COUNT(0, __builtin_amdgcn_ballot_w64(w.mode == PT_LW_IDLE && k < n));
            if (w.mode == PT_LW_IDLE && k < n) {
                mine = k;
                V3 o, d;
                io.load(k, o, d);
                w.begin(sc, B, o, d);
            }
            next = min(n, next + n_idle);
        } else if (n_idle == 64u) {
            break;
        }
        PT_PHASE_CLOCK(c_refilled);
        PT_PHASE_TIME(0, c_top, c_refilled);
        const uint32_t n_guide = (uint32_t)__builtin_popcountll(__builtin_amdgcn_ballot_w64(w.mode == PT_LW_GUIDE));
        const uint32_t n_leaf = (uint32_t)__builtin_popcountll(__builtin_amdgcn_ballot_w64(w.mode == PT_LW_LEAF));
        const uint32_t n_finish = (uint32_t)__builtin_popcountll(__builtin_amdgcn_ballot_w64(w.mode == PT_LW_FINISH || w.mode == PT_LW_VERIFY));
        const uint32_t n_literal = (uint32_t)__builtin_popcountll(__builtin_amdgcn_ballot_w64(w.mode >= PT_LW_LITERAL));
        PT_PHASE_CLOCK(c_chosen);
        PT_PHASE_TIME(5, c_refilled, c_chosen);
        uint32_t leaf_lanes = PT_ORD_LEAF_LANES, finish_lanes = PT_LW_FINISH_LANES;
        if (DEEP) {
            const uint32_t n_busy = n_guide + n_leaf + n_finish + n_literal;
            leaf_lanes = max(1u, (n_busy * PT_ORD_DEEP_LEAF_LANES) >> 6);
            finish_lanes = max(1u, (n_busy * PT_ORD_DEEP_FINISH_LANES) >> 6);
        }
        const bool run_finish = n_finish >= finish_lanes || (n_finish != 0u && n_guide == 0u && n_leaf == 0u && n_literal == 0u);
        const bool run_guide = !run_finish && n_guide != 0u && n_leaf < leaf_lanes && n_literal < leaf_lanes;
        const bool run_leaf = !run_finish && !run_guide && n_leaf != 0u && n_leaf >= n_literal;
        const bool run_literal = !run_finish && !run_guide && !run_leaf;
        const int phase_slot = run_finish ? 1 : run_guide ? 2 : run_leaf ? 3 : 4;
        (void)phase_slot;
        // (Tried: "thin" rounds for a wave with at most 24 rays in flight -- every lane takes one step of whatever kind it wants, one
        // load for all of them at the end of the round, so that a lane at a primitive does not sit out the others' box steps.  Slower
        // everywhere, 1M mesh 64 spp 34.6 -> 37.1 ms, 1 spp 3.9 -> 5.2 ms, SphereField 128 spp 85.6 -> 92.0 ms: a round then executes
        // the code of every kind of step back to back, and a lone wave's f64 primitive test takes as long as a fetch;
        // profiles/r03_thin_rounds_ab.txt.)
        const uint32_t before = w.mode;
        if (run_finish) {
            // lanes whose parent box has arrived take the proof step; lanes whose traversal just ended decide and aim at it
            PT_PHASE_COUNT(1, __builtin_amdgcn_ballot_w64(before == PT_LW_FINISH || before == PT_LW_VERIFY));
            if (before == PT_LW_VERIFY) w.step_verify(B); // (ends the ray or turns it to the literal walk: nothing to load)
            if (before == PT_LW_FINISH) {
                w.step_finish(sc);
                w.load(sc);
            }
        }
        if (run_guide) {
            if (UNROLL_STEPS) {
#pragma unroll
                for (int r = 0; r < PT_ORD_NODE_STEPS; ++r) {
                    PT_PHASE_COUNT(2 + r, __builtin_amdgcn_ballot_w64(w.mode == PT_LW_GUIDE));
                    if (w.mode == PT_LW_GUIDE) {
                        w.step_guide(sc, stack, B);
                        w.load(sc);
                    }
                }
            } else {
#pragma nounroll
                for (int r = 0; r < PT_ORD_NODE_STEPS; ++r)
                    if (w.mode == PT_LW_GUIDE) {
                        w.step_guide(sc, stack, B);
                        w.load(sc);
                    }
            }
        }
        if (run_leaf) {
            PT_PHASE_COUNT(6, __builtin_amdgcn_ballot_w64(before == PT_LW_LEAF));
            if (before == PT_LW_LEAF) {
                w.step_leaf(sc, stack, B);
                w.load(sc);
            }
        }
        if (run_literal) {
            PT_PHASE_COUNT(7, __builtin_amdgcn_ballot_w64(before >= PT_LW_LITERAL));
            if (before >= PT_LW_LITERAL) { // rare: one step per round keeps the kernel small
                w.step_literal(sc, stack, B);
                w.load(sc);
            }
        }
        const bool finished = w.mode == PT_LW_IDLE && before != PT_LW_IDLE; // this lane's ray got its answer in this round
        PT_PHASE_CLOCK(c_stepped);
        PT_PHASE_TIME(phase_slot, c_chosen, c_stepped);
        if (finished) io.store(mine, w.w.o, w.w.d, w.w.best_t, w.w.best_prim);
        PT_PHASE_CLOCK(c_stored);
        PT_PHASE_TIME(6, c_stepped, c_stored);
    }
    PT_PHASE_FLUSH;
}

template <bool UNROLL_STEPS = true, bool DEEP = false, class RayIo>
__device__ __forceinline__ void ordered_wave_loop(const OrderedScene& sc, uint32_t* stack, uint32_t n, double t_min, double t_max, RayIo& io) {
    lean_wave_loop<UNROLL_STEPS, DEEP>(sc, stack, n, t_min, t_max, io);
}

struct RegionRayIo { // the wave's region of the wavefront ray stream
    RayStream rays;
    double* hit_t;
    int32_t* hit_prim;
    size_t base;
    __device__ __forceinline__ void load(uint32_t k, V3& o, V3& d) const {
        const double2 a = rays.a[base + k], b = rays.b[base + k], c = rays.c[base + k];
        o = v3(a.x, a.y, b.x);
        d = v3(b.y, c.x, c.y);
    }
    __device__ __forceinline__ void store(uint32_t k, V3, V3, double t, int32_t prim) const {
        hit_t[base + k] = t;
        hit_prim[base + k] = prim;
    }
};

