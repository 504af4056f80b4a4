// pt_shade.h -- one step of Ray::color (ray.rs:20-41) for the 64 rays of a chunk, shared by the stand-alone shade
// kernel (shade.hip) and the fused per-wave path kernel (path.hip):
//   miss            -> L += T * background, path ends                              (ray.rs:40)
//   hit             -> L += T * emitted                                            (ray.rs:26)
//   scatter == None -> path ends                                                   (ray.rs:36)
//   scatter         -> T *= attenuation, next ray = scattered                      (ray.rs:28-34)
//   depth exhausted -> path ends with nothing more added                           (ray.rs:21-23)
// The hit record (point, normal, front_face; geometry.rs:20-43) is rebuilt from (ray, t, primitive).
#pragma once
#include "pt_launch.h"

#define PT_PERLIN_LDS_TABLES 16
#define PT_TEX_LDS_MAX 32
// The block's auxiliary LDS area (`s_perlin` in the kernels): [Perlin permutation tables (marble only)][texture table]
// [Perlin gradient table (marble only)]
template <int PROCEDURAL>
constexpr uint32_t shade_lds_perlin_bytes() { return PROCEDURAL == PT_PROC_MARBLE ? PT_PERLIN_LDS_TABLES * 256u : 0u; }
template <int PROCEDURAL>
constexpr uint32_t shade_lds_grad_offset() { return shade_lds_perlin_bytes<PROCEDURAL>() + PT_TEX_LDS_MAX * (uint32_t)sizeof(PtTexture); }
template <int PROCEDURAL>
constexpr uint32_t shade_lds_aux_bytes() {
    return PROCEDURAL == PT_PROC_NONE ? 16u
                                      : shade_lds_grad_offset<PROCEDURAL>() + (PROCEDURAL == PT_PROC_MARBLE ? PT_PERLIN_GRAD_DOUBLES * 8u : 0u);
}
static_assert(shade_lds_grad_offset<PT_PROC_MARBLE>() % 8u == 0u, "the gradient table is read as doubles");

// What is the same for every ray of a bounce.
struct ShadeEnv {
    float4* plane;       // per-path radiance sums of the pass
    uint32_t npix, sample_base;
    uint32_t stream;     // RNG stream of this bounce's scatter = bounce + 1
    uint32_t k0, k1;     // the key (seed)
    bool last_bounce, has_bg;
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

// The block's LDS copies of the small tables shade reads per hit: the ShadePrim records (TABLE_LDS), the Perlin
// permutation tables (marble's seven octaves make 168 dependent byte look-ups each: from LDS, not from L2) and the texture
// table (a checker is two dependent look-ups, parent then child: TwoSpheres' shade 19.5 -> see profiles/README.md).
template <int PROCEDURAL, bool TABLE_LDS>
__device__ __forceinline__ bool shade_stage_tables(const DevScene& sc, unsigned char* s_table, uint8_t* s_perlin) {
    const bool perlin_in_lds = PROCEDURAL == PT_PROC_MARBLE && sc.n_perlin <= PT_PERLIN_LDS_TABLES;
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
    const bool textures_in_lds = PROCEDURAL != PT_PROC_NONE && sc.n_textures <= PT_TEX_LDS_MAX;
    if (textures_in_lds) {
        const uint32_t* src = reinterpret_cast<const uint32_t*>(sc.textures);
        uint32_t* dst = reinterpret_cast<uint32_t*>(s_perlin + shade_lds_perlin_bytes<PROCEDURAL>());
        for (uint32_t k = threadIdx.x; k < sc.n_textures * (uint32_t)(sizeof(PtTexture) / 4u); k += PT_BLOCK) dst[k] = src[k];
    }
    if (PROCEDURAL == PT_PROC_MARBLE) { // the gradient table (always: 384 bytes)
        double* dst = reinterpret_cast<double*>(s_perlin + shade_lds_grad_offset<PROCEDURAL>());
        if (threadIdx.x < PT_PERLIN_GRAD_DOUBLES) dst[threadIdx.x] = kPerlinGradTable[threadIdx.x];
    }
    if (TABLE_LDS || perlin_in_lds || textures_in_lds || PROCEDURAL == PT_PROC_MARBLE) __syncthreads();
    return perlin_in_lds;
}

// Shades the chunk's rays: lane `live` holds ray (o, d), its path state st and its hit (t, prim; prim < 0 = miss).
// Returns, per lane, whether the path goes on (`alive`) with ray (next_o, next_d) and the updated state.
// Must be called by all 64 lanes of the wave together (the unit-sphere sampler is a wave-level routine).
// PROCEDURAL (PT_PROC_*): the scene has checker / marble textures (the kernel carries only the code and registers of what it has:
// 70 VGPRs for solid colours, 95 with checkers, 125 with marble).
// TABLE_LDS: the ShadePrim table is small enough to be staged in LDS once per block.
template <int PROCEDURAL, bool TABLE_LDS>
__device__ __forceinline__ void shade_chunk(const DevScene& sc, const ShadeEnv& env, const unsigned char* s_table, const uint8_t* s_perlin,
                                            bool perlin_in_lds, uint2* exchange, bool live, uint4& st, V3 o, V3 d, double t, int32_t prim,
                                            bool& alive, V3& next_o, V3& next_d) {
    alive = false;
    bool need_point = false;
    next_o = v3(0, 0, 0);
    next_d = v3(0, 0, 0);
    V3 base = v3(0, 0, 0);
    uint32_t pixel = 0, sample = 0, mat_type = PT_MAT_DIFFUSE_LIGHT;
    double mat_param = 0.0;
    // ---- per lane: the hit record, emission, and what the material does next
    if (live) {
        float tr = __uint_as_float(st.y), tg = __uint_as_float(st.z), tb = __uint_as_float(st.w);
        if (prim < 0) {
            if (env.has_bg) add_radiance(env.plane, st.x, tr, tg, tb, env.bg[0], env.bg[1], env.bg[2]);
        } else {
            const ShadePrim sp = TABLE_LDS ? reinterpret_cast<const ShadePrim*>(s_table)[prim] : sc.shade_prims[prim];
            V3 p = o + t * d;                                        // ray.rs:16-18
            V3 n_out = v3(sp.g[0], sp.g[1], sp.g[2]);                // geometry.rs:227 / :290 / :353 / :554
            if (sp.prim_type == PT_PRIM_SPHERE) n_out = (p - n_out) / sp.g[3]; // geometry.rs:151
            bool front = dot(d, n_out) < 0.0;                        // geometry.rs:29
            V3 nn = front ? n_out : neg(n_out);                      // geometry.rs:33-37
            bool emits = false;
            if (sp.mat_type == PT_MAT_DIFFUSE_LIGHT) {
                emits = front;
            } else if (!env.last_bounce) {
                uint32_t slot = st.x / env.npix;
                pixel = st.x - slot * env.npix;
                sample = env.sample_base + slot;
                mat_type = sp.mat_type;
                mat_param = sp.param;
                const int what = scatter_begin(mat_type, mat_param, d, nn, front, pixel, sample, env.stream, env.k0, env.k1, next_d, base);
                alive = what != PT_SCATTER_NONE;
                need_point = what == PT_SCATTER_NEED_POINT;
                next_o = p;
            }
            // ONE texture lookup per hit: the light's emission (material.rs:158-166) or the scatterer's albedo
            // (Lambertian / Metal attenuation, material.rs:48-53,86-91; a Dielectric's is (1,1,1))
            if (emits || need_point) {
                float cr = sp.color[0], cg = sp.color[1], cb = sp.color[2];
                if (PROCEDURAL && sp.tex_solid == 2u) { // a checker of two solid colours: no table look-ups
                    if (checker_is_odd(p)) { cr = sp.color_odd[0]; cg = sp.color_odd[1]; cb = sp.color_odd[2]; }
                } else if (PROCEDURAL && !sp.tex_solid) {
                    constexpr bool marble = PROCEDURAL == PT_PROC_MARBLE;
                    const PtTexture* textures = sc.n_textures <= PT_TEX_LDS_MAX
                                                    ? reinterpret_cast<const PtTexture*>(s_perlin + shade_lds_perlin_bytes<PROCEDURAL>())
                                                    : sc.textures;
                    const double* grad = marble ? reinterpret_cast<const double*>(s_perlin + shade_lds_grad_offset<PROCEDURAL>()) : nullptr;
                    V3 c = perlin_in_lds ? texture_value_procedural<marble>(textures, sc.n_textures, s_perlin, grad, sp.texture, p)
                                         : texture_value_procedural<marble>(textures, sc.n_textures, sc.perlin, grad, sp.texture, p);
                    cr = (float)c.x; cg = (float)c.y; cb = (float)c.z;
                }
                if (emits) {
                    add_radiance(env.plane, st.x, tr, tg, tb, cr, cg, cb);
                } else {
                    st.y = __float_as_uint(tr * cr);
                    st.z = __float_as_uint(tg * cg);
                    st.w = __float_as_uint(tb * cb);
                }
            }
        }
    }
    if (env.last_bounce) return; // nothing survives the last bounce
    // ---- the wave: unit-sphere points for the lanes that asked (vec3.rs:41-49), then their scattered directions
    if (wave_any(need_point)) {
        const V3 point = random_in_unitsphere_wave(need_point, pixel, sample, env.stream, env.k0, env.k1, exchange);
        if (need_point) next_d = scatter_finish(mat_type, mat_param, base, point);
    }
}

// Stream compaction inside the wave's own region: the survivors of a chunk, ranked by ballot + mbcnt, are appended at
// the wave's running offset out_n of the output buffers.  Returns the number appended.
__device__ __forceinline__ uint32_t append_survivors(bool alive, const RayStream& out, uint4* state_out, size_t region_base, uint32_t out_n,
                                                     V3 next_o, V3 next_d, uint4 st) {
    const unsigned long long mask = __ballot(alive);
    const uint32_t prefix = __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
    if (alive) {
        const size_t off = region_base + out_n + prefix;
        out.a[off] = make_double2(next_o.x, next_o.y);
        out.b[off] = make_double2(next_o.z, next_d.x);
        out.c[off] = make_double2(next_d.y, next_d.z);
        state_out[off] = st;
    }
    return (uint32_t)__popcll(mask);
}
