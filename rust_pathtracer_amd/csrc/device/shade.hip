// shade.hip -- one step of Ray::color (ray.rs:20-41) per live ray, iteratively:
//   miss            -> L += T * background, path ends                              (ray.rs:40)
//   hit             -> L += T * emitted                                            (ray.rs:26)
//   scatter == None -> path ends                                                   (ray.rs:36)
//   scatter         -> T *= attenuation, next ray = scattered                      (ray.rs:28-34)
//   depth exhausted -> path ends with nothing more added                           (ray.rs:21-23)
// The hit record (point, normal, front_face; geometry.rs:20-43) is rebuilt from (ray, t, primitive).
// Survivors are written compacted inside the wave's private region (pt_launch.h): wave ballot + mbcnt prefix,
// no atomics and no barriers.
// Reads 76 B/ray (ray 48, hit 12, state 16), writes 64 B per surviving ray (ray 48, state 16).
#include "pt_shade.h"

#ifndef PT_SHADE_WAVES_CHECKER
#define PT_SHADE_WAVES_CHECKER 5
#endif
#ifndef PT_SHADE_WAVES_SOLID
#define PT_SHADE_WAVES_SOLID 1
#endif
#ifndef PT_SHADE_WAVES_MARBLE
#define PT_SHADE_WAVES_MARBLE 4
#endif

struct ShadeArgs {
    RayStream in, out;
    const uint4* state_in;
    uint4* state_out;
    const double* hit_t;
    const int32_t* hit_prim;
    const uint32_t* counts_in; // this bounce's per-wave counts
    uint32_t* counts_out;      // next bounce's
    uint32_t n_waves;
    ShadeEnv env;
};

template <int PROCEDURAL, bool TABLE_LDS>
__global__ __launch_bounds__(PT_BLOCK) __attribute__((amdgpu_waves_per_eu(PROCEDURAL == PT_PROC_CHECKER ? PT_SHADE_WAVES_CHECKER : (PROCEDURAL == PT_PROC_MARBLE ? PT_SHADE_WAVES_MARBLE : PT_SHADE_WAVES_SOLID))))
void shade_kernel(DevScene sc, ShadeArgs a) {
    __shared__ __align__(16) unsigned char s_table[TABLE_LDS ? PT_SHADE_LDS_MAX_PRIMS * sizeof(ShadePrim) : 16];
    __shared__ __align__(16) uint8_t s_perlin[shade_lds_aux_bytes<PROCEDURAL>()];
    __shared__ uint2 s_exchange[PT_BLOCK]; // 64 slots per wave for the cooperative sampler
    const bool perlin_in_lds = shade_stage_tables<PROCEDURAL, TABLE_LDS>(sc, s_table, s_perlin);
    uint2* exchange = s_exchange + (threadIdx.x & ~63u);
    const uint32_t wave = blockIdx.x * (PT_BLOCK / PT_WAVE) + (threadIdx.x >> 6);
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t n = __builtin_amdgcn_readfirstlane(a.counts_in[wave]);
    const uint32_t region = a.counts_in[a.n_waves];
    const size_t region_base = (size_t)wave * region;
    uint32_t out_n = 0; // survivors this wave has written so far (wave-uniform)
    for (uint32_t first = 0; first < n; first += 64u) {
        const uint32_t k = first + lane;
        const size_t i = region_base + k;
        const bool live = k < n;
        uint4 st = make_uint4(0, 0, 0, 0);
        V3 o = v3(0, 0, 0), d = v3(0, 0, 0);
        double t = 0.0;
        int32_t prim = -1;
        if (live) { // all six stream loads are issued together: one memory latency, not a chain of them
            st = a.state_in[i];
            prim = a.hit_prim[i];
            const double2 ra = a.in.a[i], rb = a.in.b[i], rc = a.in.c[i];
            t = a.hit_t[i];
            o = v3(ra.x, ra.y, rb.x);
            d = v3(rb.y, rc.x, rc.y);
        }
        bool alive;
        V3 next_o, next_d;
        shade_chunk<PROCEDURAL, TABLE_LDS>(sc, a.env, s_table, s_perlin, perlin_in_lds, exchange, live, st, o, d, t, prim, alive, next_o, next_d);
        if (a.env.last_bounce) continue;
        out_n += append_survivors(alive, a.out, a.state_out, region_base, out_n, next_o, next_d, st);
    }
    if (lane == 0) a.counts_out[wave] = out_n;
    if (wave == 0 && lane == 0) a.counts_out[a.n_waves] = region; // survivors stay in the same layout
}

ShadeEnv make_shade_env(const PassParams& p, const Wavefront& wf, uint32_t bounce) {
    ShadeEnv e;
    e.plane = wf.plane;
    e.npix = p.npix;
    e.sample_base = p.sample_base;
    e.stream = bounce + 1u;
    e.k0 = (uint32_t)p.seed;
    e.k1 = (uint32_t)(p.seed >> 32);
    e.last_bounce = bounce + 1u >= p.max_bounces;
    e.has_bg = (p.bg[0] != 0.f) || (p.bg[1] != 0.f) || (p.bg[2] != 0.f);
    e.bg[0] = p.bg[0]; e.bg[1] = p.bg[1]; e.bg[2] = p.bg[2];
    return e;
}

void launch_shade(hipStream_t s, const LaunchGeom& g, const DevScene& sc, const PassParams& p, const Wavefront& wf,
                  const BounceIO& io, uint32_t bounce) {
    ShadeArgs a;
    a.in = wf.rays[io.in];
    a.out = wf.rays[io.out];
    a.state_in = wf.state[io.in];
    a.state_out = wf.state[io.out];
    a.hit_t = wf.hit_t;
    a.hit_prim = wf.hit_prim;
    a.counts_in = io.counts_in;
    a.counts_out = io.counts_out;
    a.n_waves = wf.n_waves;
    a.env = make_shade_env(p, wf, bounce);
    const bool lds = sc.n_prims <= PT_SHADE_LDS_MAX_PRIMS;
    // one instance per (texture kinds present, ShadePrim table in LDS or not)
    auto launch = [&](auto kernel) { hipLaunchKernelGGL(kernel, dim3(g.grid), dim3(PT_BLOCK), 0, s, sc, a); };
    if (sc.has_procedural == PT_PROC_MARBLE) {
        if (lds) launch(shade_kernel<PT_PROC_MARBLE, true>);
        else launch(shade_kernel<PT_PROC_MARBLE, false>);
    } else if (sc.has_procedural == PT_PROC_CHECKER) {
        if (lds) launch(shade_kernel<PT_PROC_CHECKER, true>);
        else launch(shade_kernel<PT_PROC_CHECKER, false>);
    } else {
        if (lds) launch(shade_kernel<PT_PROC_NONE, true>);
        else launch(shade_kernel<PT_PROC_NONE, false>);
    }
}

// ------------------------------------------------------------------ repack
// As paths die the per-wave regions thin out unevenly and a wave's last 64-ray chunk runs half empty.  Every few
// bounces the survivors are dealt out again: an exclusive prefix over the per-wave counts (every block scans the
// n_waves counts itself -- 32 KB from L2, no inter-block hand-off), each wave copies its rays to the contiguous
// position prefix + k, and the new layout is read off that contiguous stream: region' = max(64, ceil(total /
// n_waves) rounded up to 64), wave w' owning [w' * region', ...).  Copies 64 B per live ray.
// DEAL: the survivors are dealt out one at a time instead -- ray j of the contiguous order goes to wave j mod n_waves -- before
// the per-wave tail of the bounce loop (path.hip): the long-lived paths of a frame sit in a few places of the image (the gap under
// the mesh), so in a few waves, and a tail wave's cost per bounce is that of its slowest ray.
template <bool DEAL>
__global__ __launch_bounds__(PT_BLOCK) void repack_kernel(RayStream in, const uint4* __restrict__ st_in, RayStream out,
                                                          uint4* __restrict__ st_out, const uint32_t* __restrict__ counts_in,
                                                          uint32_t* __restrict__ counts_out, uint32_t n_waves, uint32_t deal_waves) {
    __shared__ uint32_t s_part[PT_BLOCK];
    const uint32_t per = (n_waves + PT_BLOCK - 1) / PT_BLOCK;
    uint32_t acc = 0;
    for (uint32_t k = 0; k < per; ++k) {
        uint32_t w = threadIdx.x * per + k;
        if (w < n_waves) acc += counts_in[w];
    }
    s_part[threadIdx.x] = acc;
    __syncthreads();
    const uint32_t wave = blockIdx.x * (PT_BLOCK / PT_WAVE) + (threadIdx.x >> 6);
    const uint32_t lane = threadIdx.x & 63u;
    // prefix of this wave and the grand total (wave-uniform; PT_BLOCK partial sums, read by every lane)
    uint32_t prefix = 0, total = 0;
    const uint32_t my_part = wave / per;
    for (uint32_t t = 0; t < PT_BLOCK; ++t) {
        uint32_t v = s_part[t];
        if (t < my_part) prefix += v;
        total += v;
    }
    for (uint32_t w = my_part * per; w < wave; ++w) prefix += counts_in[w];
    const uint32_t n = counts_in[wave];
    const size_t src_base = (size_t)wave * counts_in[n_waves];
    // (DEAL: over the first deal_waves waves only -- late in the tail a wave holds a handful of rays and pays a whole wave's
    // instructions for them; fewer, fuller waves pay them less often)
    uint32_t targets = DEAL ? deal_waves : n_waves;
    uint32_t region = ((total + targets - 1) / targets + 63u) & ~63u;
    if (region < 64u) region = 64u;
    // (fewer target waves mean larger regions, rounded up to 64 each: when nearly every path is still alive that can exceed
    // what the streams hold -- n_waves regions of the current size -- and the deal then goes over all the waves, which always fits:
    // round64(ceil(total / n_waves)) never exceeds the region size the pass started with)
    if (DEAL && (unsigned long long)targets * region > (unsigned long long)n_waves * counts_in[n_waves]) {
        targets = n_waves;
        region = ((total + targets - 1) / targets + 63u) & ~63u;
        if (region < 64u) region = 64u;
    }
    for (uint32_t k = lane; k < n; k += 64u) {
        const size_t src = src_base + k;
        const uint32_t j = prefix + k;
        const size_t dst = DEAL ? (size_t)(j % targets) * region + j / targets : (size_t)j;
        out.a[dst] = in.a[src];
        out.b[dst] = in.b[src];
        out.c[dst] = in.c[src];
        st_out[dst] = st_in[src];
    }
    if (lane == 0) {
        const unsigned long long start = (unsigned long long)wave * region;
        if (DEAL) counts_out[wave] = wave < targets ? total / targets + (wave < total % targets ? 1u : 0u) : 0u;
        else counts_out[wave] = start >= total ? 0u : (total - start < region ? (uint32_t)(total - start) : region);
    }
    if (wave == 0 && lane == 0) counts_out[n_waves] = region;
}

void launch_repack(hipStream_t s, const LaunchGeom& g, const Wavefront& wf, const BounceIO& io, uint32_t deal_waves) {
    if (deal_waves)
        hipLaunchKernelGGL(repack_kernel<true>, dim3(g.grid), dim3(PT_BLOCK), 0, s, wf.rays[io.in], wf.state[io.in], wf.rays[io.out],
                           wf.state[io.out], io.counts_in, io.counts_out, wf.n_waves, std::min(deal_waves, wf.n_waves));
    else
        hipLaunchKernelGGL(repack_kernel<false>, dim3(g.grid), dim3(PT_BLOCK), 0, s, wf.rays[io.in], wf.state[io.in], wf.rays[io.out],
                           wf.state[io.out], io.counts_in, io.counts_out, wf.n_waves, wf.n_waves);
}
