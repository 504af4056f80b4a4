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
#include "pt_launch.h"

#define PT_PERLIN_LDS_TABLES 16

struct ShadeArgs {
    RayStream in, out;
    const uint4* state_in;
    uint4* state_out;
    const double* hit_t;
    const int32_t* hit_prim;
    float4* plane;
    const uint32_t* counts_in; // this bounce's per-wave counts
    uint32_t* counts_out;      // next bounce's
    uint32_t n_waves;
    uint32_t bounce, max_bounces, npix, sample_base;
    uint64_t seed;
    float bg[3];
};

__device__ __forceinline__ void add_radiance(float4* plane, uint32_t path, float tr, float tg, float tb, float er, float eg,
                                             float eb) {
    float4 v = plane[path]; // this lane is the only owner of the slot
    v.x += tr * er;
    v.y += tg * eg;
    v.z += tb * eb;
    plane[path] = v;
}

// PROCEDURAL: the scene has checker / marble textures (else the kernel carries neither the call nor its registers).
// TABLE_LDS: the ShadePrim table is small enough to be staged in LDS once per block.
template <bool PROCEDURAL, bool TABLE_LDS>
__global__ __launch_bounds__(PT_BLOCK) void shade_kernel(DevScene sc, ShadeArgs a) {
    __shared__ __align__(16) unsigned char s_table[TABLE_LDS ? PT_SHADE_LDS_MAX_PRIMS * sizeof(ShadePrim) : 16];
    // marble's seven octaves make 168 dependent byte look-ups in a 256-byte permutation table: from LDS, not from L2
    __shared__ __align__(16) uint8_t s_perlin[PROCEDURAL ? PT_PERLIN_LDS_TABLES * 256 : 16];
    const bool perlin_in_lds = PROCEDURAL && sc.n_perlin <= PT_PERLIN_LDS_TABLES;
    if (TABLE_LDS) {
        const uint4* src = reinterpret_cast<const uint4*>(sc.shade_prims);
        uint4* dst = reinterpret_cast<uint4*>(s_table);
        for (uint32_t k = threadIdx.x; k < sc.n_prims * 5u; k += PT_BLOCK) dst[k] = src[k];
    }
    if (perlin_in_lds) {
        const uint32_t* src = reinterpret_cast<const uint32_t*>(sc.perlin);
        uint32_t* dst = reinterpret_cast<uint32_t*>(s_perlin);
        for (uint32_t k = threadIdx.x; k < sc.n_perlin * 64u; k += PT_BLOCK) dst[k] = src[k];
    }
    if (TABLE_LDS || perlin_in_lds) __syncthreads();
    const uint32_t wave = blockIdx.x * (PT_BLOCK / PT_WAVE) + (threadIdx.x >> 6);
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t n = __builtin_amdgcn_readfirstlane(a.counts_in[wave]);
    const uint32_t region = a.counts_in[a.n_waves];
    const size_t region_base = (size_t)wave * region;
    const bool last_bounce = (a.bounce + 1u >= a.max_bounces);
    const bool has_bg = (a.bg[0] != 0.f) || (a.bg[1] != 0.f) || (a.bg[2] != 0.f);
    uint32_t out_n = 0; // survivors this wave has written so far (wave-uniform)
    __shared__ uint2 s_exchange[PT_BLOCK]; // 64 slots per wave for the cooperative sampler
    uint2* exchange = s_exchange + (threadIdx.x & ~63u);
    const uint32_t k0 = (uint32_t)a.seed, k1 = (uint32_t)(a.seed >> 32), stream = a.bounce + 1u;
    for (uint32_t first = 0; first < n; first += 64u) {
        const uint32_t k = first + lane;
        const size_t i = region_base + k;
        bool alive = false, need_point = false;
        V3 next_o = v3(0, 0, 0), next_d = v3(0, 0, 0), base = v3(0, 0, 0);
        uint4 st = make_uint4(0, 0, 0, 0);
        uint32_t pixel = 0, sample = 0, mat_type = PT_MAT_DIFFUSE_LIGHT;
        double mat_param = 0.0;
        // ---- per lane: the hit record, emission, and what the material does next
        if (k < n) {
            // all six stream loads are issued together: one memory latency, not a chain of them
            st = a.state_in[i];
            const int32_t prim = a.hit_prim[i];
            const double2 ra = a.in.a[i], rb = a.in.b[i], rc = a.in.c[i];
            const double t = a.hit_t[i];
            float tr = __uint_as_float(st.y), tg = __uint_as_float(st.z), tb = __uint_as_float(st.w);
            if (prim < 0) {
                if (has_bg) add_radiance(a.plane, st.x, tr, tg, tb, a.bg[0], a.bg[1], a.bg[2]);
            } else {
                const ShadePrim sp = TABLE_LDS ? reinterpret_cast<const ShadePrim*>(s_table)[prim] : sc.shade_prims[prim];
                V3 o = v3(ra.x, ra.y, rb.x), d = v3(rb.y, rc.x, rc.y);
                V3 p = o + t * d;                                        // ray.rs:16-18
                V3 n_out = v3(sp.g[0], sp.g[1], sp.g[2]);                // geometry.rs:227 / :290 / :353 / :554
                if (sp.prim_type == PT_PRIM_SPHERE) n_out = (p - n_out) / sp.g[3]; // geometry.rs:151
                bool front = dot(d, n_out) < 0.0;                        // geometry.rs:29
                V3 nn = front ? n_out : neg(n_out);                      // geometry.rs:33-37
                bool emits = false;
                if (sp.mat_type == PT_MAT_DIFFUSE_LIGHT) {
                    emits = front;
                } else if (!last_bounce) {
                    uint32_t slot = st.x / a.npix;
                    pixel = st.x - slot * a.npix;
                    sample = a.sample_base + slot;
                    mat_type = sp.mat_type;
                    mat_param = sp.param;
                    const int what = scatter_begin(mat_type, mat_param, d, nn, front, pixel, sample, stream, k0, k1, next_d, base);
                    alive = what != PT_SCATTER_NONE;
                    need_point = what == PT_SCATTER_NEED_POINT;
                    next_o = p;
                }
                // ONE texture lookup per hit: the light's emission (material.rs:158-166) or the scatterer's albedo
                // (Lambertian / Metal attenuation, material.rs:48-53,86-91; a Dielectric's is (1,1,1))
                if (emits || need_point) {
                    float cr = sp.color[0], cg = sp.color[1], cb = sp.color[2];
                    if (PROCEDURAL && !sp.tex_solid) {
                        V3 c = perlin_in_lds ? texture_value_procedural(sc.textures, sc.n_textures, s_perlin, sp.texture, p)
                                             : texture_value_procedural(sc.textures, sc.n_textures, sc.perlin, sp.texture, p);
                        cr = (float)c.x; cg = (float)c.y; cb = (float)c.z;
                    }
                    if (emits) {
                        add_radiance(a.plane, st.x, tr, tg, tb, cr, cg, cb);
                    } else {
                        st.y = __float_as_uint(tr * cr);
                        st.z = __float_as_uint(tg * cg);
                        st.w = __float_as_uint(tb * cb);
                    }
                }
            }
        }
        if (last_bounce) continue; // nothing survives the last bounce
        // ---- the wave: unit-sphere points for the lanes that asked (vec3.rs:41-49), then their scattered directions
        if (wave_any(need_point)) {
#ifdef PT_ABLATE_NO_SAMPLER
            const V3 point = v3(0.3 + 1e-9 * pixel, 0.4, 0.5 - 1e-9 * sample);
#elif defined(PT_ABLATE_ONE_ROUND)
            V3 point; bool acc_; unitsphere_pair(pixel, sample, stream, 0u, k0, k1, point, acc_);
#else
            const V3 point = random_in_unitsphere_wave(need_point, pixel, sample, stream, k0, k1, exchange);
#endif
            if (need_point) next_d = scatter_finish(mat_type, mat_param, base, point);
        }
        // ---- stream compaction inside the wave's own region: rank by ballot + mbcnt, append at the running offset
        const unsigned long long mask = __ballot(alive);
        const uint32_t prefix = __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
        if (alive) {
            const size_t off = region_base + out_n + prefix;
            a.out.a[off] = make_double2(next_o.x, next_o.y);
            a.out.b[off] = make_double2(next_o.z, next_d.x);
            a.out.c[off] = make_double2(next_d.y, next_d.z);
            a.state_out[off] = st;
        }
        out_n += (uint32_t)__popcll(mask);
    }
    if (lane == 0) a.counts_out[wave] = out_n;
    if (wave == 0 && lane == 0) a.counts_out[a.n_waves] = region; // survivors stay in the same layout
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
    a.plane = wf.plane;
    a.counts_in = io.counts_in;
    a.counts_out = io.counts_out;
    a.n_waves = wf.n_waves;
    a.bounce = bounce;
    a.max_bounces = p.max_bounces;
    a.npix = p.npix;
    a.sample_base = p.sample_base;
    a.seed = p.seed;
    a.bg[0] = p.bg[0]; a.bg[1] = p.bg[1]; a.bg[2] = p.bg[2];
    const bool lds = sc.n_prims <= PT_SHADE_LDS_MAX_PRIMS;
    if (sc.has_procedural) {
        if (lds) hipLaunchKernelGGL((shade_kernel<true, true>), dim3(g.grid), dim3(PT_BLOCK), 0, s, sc, a);
        else hipLaunchKernelGGL((shade_kernel<true, false>), dim3(g.grid), dim3(PT_BLOCK), 0, s, sc, a);
    } else {
        if (lds) hipLaunchKernelGGL((shade_kernel<false, true>), dim3(g.grid), dim3(PT_BLOCK), 0, s, sc, a);
        else hipLaunchKernelGGL((shade_kernel<false, false>), dim3(g.grid), dim3(PT_BLOCK), 0, s, sc, a);
    }
}

// ------------------------------------------------------------------ repack
// As paths die the per-wave regions thin out unevenly and a wave's last 64-ray chunk runs half empty.  Every few
// bounces the survivors are dealt out again: an exclusive prefix over the per-wave counts (every block scans the
// n_waves counts itself -- 32 KB from L2, no inter-block hand-off), each wave copies its rays to the contiguous
// position prefix + k, and the new layout is read off that contiguous stream: region' = max(64, ceil(total /
// n_waves) rounded up to 64), wave w' owning [w' * region', ...).  Copies 64 B per live ray.
__global__ __launch_bounds__(PT_BLOCK) void repack_kernel(RayStream in, const uint4* __restrict__ st_in, RayStream out,
                                                          uint4* __restrict__ st_out, const uint32_t* __restrict__ counts_in,
                                                          uint32_t* __restrict__ counts_out, uint32_t n_waves) {
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
    for (uint32_t k = lane; k < n; k += 64u) {
        const size_t src = src_base + k, dst = (size_t)prefix + k;
        out.a[dst] = in.a[src];
        out.b[dst] = in.b[src];
        out.c[dst] = in.c[src];
        st_out[dst] = st_in[src];
    }
    uint32_t region = ((total + n_waves - 1) / n_waves + 63u) & ~63u;
    if (region < 64u) region = 64u;
    if (lane == 0) {
        const unsigned long long start = (unsigned long long)wave * region;
        counts_out[wave] = start >= total ? 0u : (total - start < region ? (uint32_t)(total - start) : region);
    }
    if (wave == 0 && lane == 0) counts_out[n_waves] = region;
}

void launch_repack(hipStream_t s, const LaunchGeom& g, const Wavefront& wf, const BounceIO& io) {
    hipLaunchKernelGGL(repack_kernel, dim3(g.grid), dim3(PT_BLOCK), 0, s, wf.rays[io.in], wf.state[io.in], wf.rays[io.out],
                       wf.state[io.out], io.counts_in, io.counts_out, wf.n_waves);
}
