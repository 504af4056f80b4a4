// path.hip -- the bounce loop of a pass from some bounce on in ONE launch, each wave on its own region.
// First the version for scenes that take the wave-uniform walk (options "fused": from bounce 0; "tail_from": from a later one).
//
// Wave-private regions (pt_launch.h) mean that a wave never needs anything another wave produced: its rays stay in its
// own region through every bounce.  So the wave runs Ray::color's loop (ray.rs:20-41) by itself,
//     for each bounce:  for each 64-ray chunk of its region:  load ray + state -> trace (lockstep walk) -> shade ->
//                       append the survivors to the region of the other ping-pong buffer,
// with the hit record never leaving registers.  Per ray segment this moves 64 B in and 64 B per survivor out (the
// separate trace and shade kernels: 200 B), costs no kernel launches or grid-wide joins per bounce (a launch ends with
// its slowest wave; with 100 launches per pass the wave slots stood empty a third of the time), and interleaves the
// compute-heavy walk with shade's memory traffic at chunk granularity on every SIMD.
// The per-bounce count rows are still written (ray statistics; the stand-alone kernels' format).
//
// MEASURED (CornellBox 1080p 256 spp, one MI355X): 169 ms/frame on one stream against 167 ms for the separate kernels on
// one stream and 145 ms for the separate kernels on two overlapped streams.  The pass is bound by instruction issue
// (about 2300 wave instructions per 64 rays whichever way they are packaged), and this kernel's 87 VGPRs allow only
// 5 waves per SIMD where the walk alone runs 8.  It is therefore OFF by default (option "fused"); it stays as the
// variant that allocates the least traffic (115 B instead of 200 B per segment) for machines where bandwidth binds.
#include "pt_shade.h"
#include "pt_ordered.h"

struct PathArgs {
    RayStream rays[2];
    uint4* state[2];
    const uint32_t* counts_in; // the row of first_bounce (raygen's, or the preceding shade launch's)
    uint32_t* counts;          // all rows (n_waves + 1 words each); rows first_bounce + 1 .. max_bounces are written here
    uint32_t n_waves;
    uint32_t first_bounce, max_bounces;
    int first_buf;             // which ping-pong buffer holds first_bounce's rays
    ShadeEnv env;     // stream and last_bounce are set per bounce by the kernel
};

template <bool TABLE_LDS>
__global__ __launch_bounds__(PT_BLOCK) void path_kernel(DevScene sc, PathArgs a) {
    __shared__ __align__(16) unsigned char s_table[TABLE_LDS ? PT_SHADE_LDS_MAX_PRIMS * sizeof(ShadePrim) : 16];
    __shared__ __align__(16) uint8_t s_perlin[16];
    __shared__ uint2 s_exchange[PT_BLOCK]; // 64 slots per wave for the cooperative sampler
    shade_stage_tables<PT_PROC_NONE, TABLE_LDS>(sc, s_table, s_perlin);
    uint2* exchange = s_exchange + (threadIdx.x & ~63u);
    const uint32_t wave = blockIdx.x * (PT_BLOCK / PT_WAVE) + (threadIdx.x >> 6);
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t row = a.n_waves + 1u;
    uint32_t n = __builtin_amdgcn_readfirstlane(a.counts_in[wave]);
    const uint32_t region = a.counts_in[a.n_waves];
    const size_t region_base = (size_t)wave * region;
    ShadeEnv env = a.env;
    uint32_t bounce = a.first_bounce;
    for (; bounce < a.max_bounces && n != 0u; ++bounce) {
        env.stream = bounce + 1u;
        env.last_bounce = bounce + 1u >= a.max_bounces;
        const bool odd = ((bounce - a.first_bounce) & 1u) != (uint32_t)a.first_buf;
        const RayStream in = odd ? a.rays[1] : a.rays[0], out = odd ? a.rays[0] : a.rays[1];
        const uint4* state_in = odd ? a.state[1] : a.state[0];
        uint4* state_out = odd ? a.state[0] : a.state[1];
        uint32_t out_n = 0; // survivors this wave has written so far (wave-uniform)
        for (uint32_t first = 0; first < n; first += 64u) {
            const uint32_t k = first + lane;
            const size_t i = region_base + k;
            const bool live = k < n;
            // idle lanes carry an ordinary ray (see trace_walk_kernel)
            double2 ra = make_double2(0x1p40, 0x1p40), rb = make_double2(0x1p40, 1), rc = make_double2(1, 1);
            uint4 st = make_uint4(0, 0, 0, 0);
            if (live) {
                ra = in.a[i];
                rb = in.b[i];
                rc = in.c[i];
                st = state_in[i];
            }
            const V3 o = v3(ra.x, ra.y, rb.x), d = v3(rb.y, rc.x, rc.y);
            double t;
            int32_t prim;
            traverse_walk(sc.walk, sc.walk_bytes, sc.walk_flags, live, o, d, 0.001, __builtin_huge_val(), t, prim); // ray.rs:25
            bool alive;
            V3 next_o, next_d;
            shade_chunk<PT_PROC_NONE, TABLE_LDS>(sc, env, s_table, s_perlin, false, exchange, live, st, o, d, t, prim, alive, next_o, next_d);
            if (!env.last_bounce) out_n += append_survivors(alive, out, state_out, region_base, out_n, next_o, next_d, st);
        }
        if (lane == 0) a.counts[(size_t)(bounce + 1u) * row + wave] = out_n;
        if (wave == 0 && lane == 0) a.counts[(size_t)(bounce + 1u) * row + a.n_waves] = region;
        // the wave's own stores (survivors, radiance sums) before its loads of the next bounce: same CU, same L1
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
        n = out_n;
    }
    // rows this wave never reached (it ran out of rays): zero, as the stand-alone shade kernel would have written
    for (uint32_t b = bounce + 1u; b <= a.max_bounces; ++b)
        if (lane == 0) a.counts[(size_t)b * row + wave] = 0u;
}

void launch_path(hipStream_t s, const LaunchGeom& g, const DevScene& sc, const PassParams& p, const Wavefront& wf, const BounceIO& io,
                 uint32_t first_bounce) {
    PathArgs a;
    a.rays[0] = wf.rays[0];
    a.rays[1] = wf.rays[1];
    a.state[0] = wf.state[0];
    a.state[1] = wf.state[1];
    a.counts_in = io.counts_in;
    a.counts = wf.counts;
    a.n_waves = wf.n_waves;
    a.first_bounce = first_bounce;
    a.max_bounces = p.max_bounces;
    a.first_buf = io.in;
    a.env = make_shade_env(p, wf, first_bounce);
    if (sc.n_prims <= PT_SHADE_LDS_MAX_PRIMS) hipLaunchKernelGGL(path_kernel<true>, dim3(g.grid), dim3(PT_BLOCK), 0, s, sc, a);
    else hipLaunchKernelGGL(path_kernel<false>, dim3(g.grid), dim3(PT_BLOCK), 0, s, sc, a);
}

// ------------------------------------------------------------------ the tail of the bounce loop, per wave (ordered walk)
// Larger scenes: after a few bounces only a small fraction of the paths is alive, and a trace launch lasts as long as
// its slowest ray -- hundreds of dependent fetches for a ray grazing a fine mesh -- so fifty launches cost the SUM over
// the bounces of the slowest ray of each.  From bounce `first_bounce` on, each wave therefore runs the rest of the
// loop on its own region in this one launch (trace its rays with the ordered wave loop, shade them, append the survivors,
// next bounce): the pass then ends with the slowest WAVE, whose cost is a sum over its own few rays.
// Same results bit for bit: the per-ray work is that of trace_ordered_kernel + shade_kernel.
// (Tried: every LANE carrying its own path through the remaining bounces, shading whenever 16 lanes of the wave are ready --
// no faster, 19.9 against 20.2 ms on the 8-spp mesh frame and equal at 64 spp: the tail is bound by the longest single PATH,
// ~46 bounces x ~100 dependent fetches, not by lanes waiting for each other.)
struct TailArgs {
    RayStream rays[2];
    uint4* state[2];
    double* hit_t;
    int32_t* hit_prim;
    const uint32_t* counts_in; // the row of first_bounce
    uint32_t* counts;          // all rows (n_waves + 1 words each); rows first_bounce + 1 .. max_bounces are written here
    uint32_t n_waves, first_bounce, max_bounces;
    int first_buf;             // which ping-pong buffer holds first_bounce's rays
    ShadeEnv env;              // stream and last_bounce are set per bounce by the kernel
    OrderedScene os;
};

#ifndef PT_TAIL_WAVES
#define PT_TAIL_WAVES 5
#endif
template <int PROCEDURAL, bool TABLE_LDS>
// (Occupancy: where LDS allows -- tables in global memory, no marble -- the kernel is held to 96 VGPRs = five waves per SIMD at
// the price of ~28 spilled values; 1M-triangle mesh, 1080p 64 spp, ms/frame: 47.8 at five waves, 48.0 at four (112 VGPRs, no
// spills), 60.1 at six (80 VGPRs, 60-100 spills: scratch traffic on the critical path of every step).)
__global__ __launch_bounds__(PT_BLOCK) __attribute__((amdgpu_waves_per_eu((TABLE_LDS || PROCEDURAL == PT_PROC_MARBLE) ? 1 : PT_TAIL_WAVES)))
void tail_kernel(DevScene sc, TailArgs a) {
    __shared__ __align__(16) unsigned char s_table[TABLE_LDS ? PT_SHADE_LDS_MAX_PRIMS * sizeof(ShadePrim) : 16];
    __shared__ __align__(16) uint8_t s_perlin[shade_lds_aux_bytes<PROCEDURAL>()];
    __shared__ uint2 s_exchange[PT_BLOCK];
    __shared__ uint32_t s_stack[PT_GUIDE_STACK_SLOTS * PT_BLOCK];
    const bool perlin_in_lds = shade_stage_tables<PROCEDURAL, TABLE_LDS>(sc, s_table, s_perlin);
    uint2* exchange = s_exchange + (threadIdx.x & ~63u);
    const uint32_t wave = blockIdx.x * (PT_BLOCK / PT_WAVE) + (threadIdx.x >> 6);
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t row = a.n_waves + 1u;
    uint32_t n = __builtin_amdgcn_readfirstlane(a.counts_in[wave]);
    const uint32_t region = a.counts_in[a.n_waves];
    const size_t region_base = (size_t)wave * region;
    ShadeEnv env = a.env;
    uint32_t bounce = a.first_bounce;
    for (; bounce < a.max_bounces && n != 0u; ++bounce) {
        env.stream = bounce + 1u;
        env.last_bounce = bounce + 1u >= a.max_bounces;
        const bool odd = ((bounce - a.first_bounce) & 1u) != (uint32_t)a.first_buf;
        const RayStream in = odd ? a.rays[1] : a.rays[0], out = odd ? a.rays[0] : a.rays[1];
        const uint4* state_in = odd ? a.state[1] : a.state[0];
        uint4* state_out = odd ? a.state[0] : a.state[1];
        RegionRayIo io{in, a.hit_t, a.hit_prim, region_base};
        ordered_wave_loop<false, true>(a.os, s_stack + threadIdx.x, n, 0.001, __builtin_huge_val(), io); // ray.rs:25
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup"); // the wave's hits before it reads them back
        uint32_t out_n = 0;
        for (uint32_t first = 0; first < n; first += 64u) {
            const uint32_t k = first + lane;
            const size_t i = region_base + k;
            const bool live = k < n;
            uint4 st = make_uint4(0, 0, 0, 0);
            V3 o = v3(0, 0, 0), d = v3(0, 0, 0);
            double t = 0.0;
            int32_t prim = -1;
            if (live) {
                st = state_in[i];
                prim = a.hit_prim[i];
                const double2 ra = in.a[i], rb = in.b[i], rc = in.c[i];
                t = a.hit_t[i];
                o = v3(ra.x, ra.y, rb.x);
                d = v3(rb.y, rc.x, rc.y);
            }
            bool alive;
            V3 next_o, next_d;
            shade_chunk<PROCEDURAL, TABLE_LDS>(sc, env, s_table, s_perlin, perlin_in_lds, exchange, live, st, o, d, t, prim, alive, next_o, next_d);
            if (!env.last_bounce) out_n += append_survivors(alive, out, state_out, region_base, out_n, next_o, next_d, st);
        }
        if (lane == 0) a.counts[(size_t)(bounce + 1u) * row + wave] = out_n;
        if (wave == 0 && lane == 0) a.counts[(size_t)(bounce + 1u) * row + a.n_waves] = region;
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup"); // survivors and radiance sums before the next bounce's loads
        n = out_n;
    }
    // rows this wave never reached (it ran out of rays): zero, as the stand-alone shade kernel would have written
    for (uint32_t b = bounce + 1u; b <= a.max_bounces; ++b)
        if (lane == 0) a.counts[(size_t)b * row + wave] = 0u;
}

void launch_tail(hipStream_t s, const LaunchGeom& g, const DevScene& sc, const PassParams& p, const Wavefront& wf, const BounceIO& io,
                 uint32_t first_bounce) {
    TailArgs a;
    a.rays[0] = wf.rays[0];
    a.rays[1] = wf.rays[1];
    a.state[0] = wf.state[0];
    a.state[1] = wf.state[1];
    a.hit_t = wf.hit_t;
    a.hit_prim = wf.hit_prim;
    a.counts_in = io.counts_in;
    a.counts = wf.counts;
    a.n_waves = wf.n_waves;
    a.first_bounce = first_bounce;
    a.max_bounces = p.max_bounces;
    a.first_buf = io.in;
    a.env = make_shade_env(p, wf, first_bounce);
    a.os = ordered_scene(sc);
    const bool lds = sc.n_prims <= PT_SHADE_LDS_MAX_PRIMS;
    // one instance per (texture kinds present, ShadePrim table in LDS or not)
    auto launch = [&](auto kernel) { hipLaunchKernelGGL(kernel, dim3(g.grid), dim3(PT_BLOCK), 0, s, sc, a); };
    if (sc.has_procedural == PT_PROC_MARBLE) {
        if (lds) launch(tail_kernel<PT_PROC_MARBLE, true>);
        else launch(tail_kernel<PT_PROC_MARBLE, false>);
    } else if (sc.has_procedural == PT_PROC_CHECKER) {
        if (lds) launch(tail_kernel<PT_PROC_CHECKER, true>);
        else launch(tail_kernel<PT_PROC_CHECKER, false>);
    } else {
        if (lds) launch(tail_kernel<PT_PROC_NONE, true>);
        else launch(tail_kernel<PT_PROC_NONE, false>);
    }
}
