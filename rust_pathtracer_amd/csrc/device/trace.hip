// trace.hip -- closest-hit query per live ray: BvhNode::hit (bvh.rs:107-121) over the flattened tree.
//
// Persistent grid: wave w walks its own region of the ray stream (pt_launch.h).  Stack variant: each block stages
// the scene prefix (the top BVH levels, or the whole scene when it is small) in LDS once.  A lane's pending right children sit on a stack in
// LDS laid out [level][thread], so a wave's push/pop touches 64 consecutive banks.
// Reads 48 B/ray (three coalesced 16-B streams), writes 12 B/ray (t, primitive).
#include "pt_ordered.h"

__device__ __forceinline__ void stage_scene(const DevScene& sc, unsigned char* smem) {
    // nodes are 64 B, primitives 80 B: copy as 16-B pieces
    const uint4* src_n = reinterpret_cast<const uint4*>(sc.nodes);
    uint4* dst_n = reinterpret_cast<uint4*>(smem);
    const uint32_t pieces_n = sc.lds_nodes * 4u;
    for (uint32_t k = threadIdx.x; k < pieces_n; k += blockDim.x) dst_n[k] = src_n[k];
    const uint4* src_p = reinterpret_cast<const uint4*>(sc.prims);
    uint4* dst_p = reinterpret_cast<uint4*>(smem + (size_t)sc.lds_nodes * sizeof(DevNode));
    const uint32_t pieces_p = sc.lds_prims * 5u;
    for (uint32_t k = threadIdx.x; k < pieces_p; k += blockDim.x) dst_p[k] = src_p[k];
    __syncthreads();
}

template <bool ALL_LDS>
__global__ __launch_bounds__(PT_BLOCK) void trace_kernel(DevScene sc, RayStream rays, const uint32_t* __restrict__ counts,
                                                         uint32_t n_waves, double* hit_t, int32_t* hit_prim) {
    extern __shared__ __align__(16) unsigned char smem[];
    stage_scene(sc, smem);
    const DevNode* lds_nodes = reinterpret_cast<const DevNode*>(smem);
    const DevPrim* lds_prims = reinterpret_cast<const DevPrim*>(smem + (size_t)sc.lds_nodes * sizeof(DevNode));
    uint32_t* stack = reinterpret_cast<uint32_t*>(smem + (size_t)sc.lds_nodes * sizeof(DevNode) +
                                                  (size_t)sc.lds_prims * sizeof(DevPrim)) + threadIdx.x;
    const uint32_t wave = blockIdx.x * (PT_BLOCK / PT_WAVE) + (threadIdx.x >> 6);
    const uint32_t n = counts[wave];
    const size_t base = (size_t)wave * counts[n_waves];
    for (uint32_t k = threadIdx.x & 63u; k < n; k += 64u) {
        const size_t i = base + k;
        double2 a = rays.a[i], b = rays.b[i], c = rays.c[i];
        double t;
        int32_t prim;
        // ray.rs:25: t_min = 0.001, t_max = +inf
        traverse<ALL_LDS>(sc, lds_nodes, lds_prims, stack, v3(a.x, a.y, b.x), v3(b.y, c.x, c.y), 0.001,
                          __builtin_huge_val(), t, prim);
        hit_t[i] = t;
        hit_prim[i] = prim;
    }
}

// Small scenes: the wave-uniform walk (pt_device.h traverse_walk).  No LDS, no per-lane stack.
__global__ __launch_bounds__(PT_BLOCK) void trace_walk_kernel(const unsigned char* __restrict__ walk, uint32_t walk_bytes, uint32_t walk_flags, RayStream rays,
                                                              const uint32_t* __restrict__ counts, uint32_t n_waves,
                                                              double* __restrict__ hit_t, int32_t* __restrict__ hit_prim) {
    const uint32_t wave = blockIdx.x * (PT_BLOCK / PT_WAVE) + (threadIdx.x >> 6);
    const uint32_t n = __builtin_amdgcn_readfirstlane(counts[wave]);
    const size_t base = (size_t)wave * counts[n_waves];
    for (uint32_t first = 0; first < n; first += 64u) { // whole waves iterate together; idle lanes are masked inside
        const uint32_t k = first + (threadIdx.x & 63u);
        const bool live = k < n;
        const size_t i = base + k;
        // idle lanes carry an ordinary ray (finite reciprocals, nonzero plane offsets): they take part in the wave-wide
        // votes of the walk (div_given_reciprocal) and must not drag the wave onto its slow path
        double2 a = make_double2(0x1p40, 0x1p40), b = make_double2(0x1p40, 1), c = make_double2(1, 1);
        if (live) {
            a = rays.a[i];
            b = rays.b[i];
            c = rays.c[i];
        }
        double t;
        int32_t prim;
        traverse_walk(walk, walk_bytes, walk_flags, live, v3(a.x, a.y, b.x), v3(b.y, c.x, c.y), 0.001, __builtin_huge_val(), t, prim);
        if (live) {
            hit_t[i] = t;
            hit_prim[i] = prim;
        }
    }
}

// Larger scenes: one lane per ray over the same record stream (pt_device.h StreamWalk).  No LDS.
// Rays differ widely in the number of records they visit (SphereField: 40 on average, 120 for the slowest of 64), so a
// lane does not wait for its wave: whenever PT_REFILL_LANES lanes are out of work, each takes the next unassigned ray of the
// wave's region (rank among the idle lanes by ballot + mbcnt; the region is private to the wave, so no atomics).
#ifndef PT_REFILL_LANES
#define PT_REFILL_LANES 16
#endif
#ifndef PT_NODE_STEPS
#define PT_NODE_STEPS 4
#endif
#ifndef PT_LEAF_LANES
#define PT_LEAF_LANES 24
#endif
__global__ __launch_bounds__(PT_BLOCK) void trace_stream_kernel(const unsigned char* __restrict__ stream, uint32_t stream_bytes, RayStream rays,
                                                                const uint32_t* __restrict__ counts, uint32_t n_waves,
                                                                double* __restrict__ hit_t, int32_t* __restrict__ hit_prim) {
    const uint32_t wave = blockIdx.x * (blockDim.x / PT_WAVE) + (threadIdx.x >> 6);
    const uint32_t n = __builtin_amdgcn_readfirstlane(counts[wave]);
    const size_t base = (size_t)wave * counts[n_waves];
    uint32_t next = 0; // first ray of the region not yet handed to a lane (wave-uniform)
    bool busy = false;
    size_t mine = 0;
    StreamWalk w;
    w.begin(stream, v3(0, 0, 0), v3(1, 1, 1), 0.0, 0.0);
    for (;;) {
        const unsigned long long idle = __builtin_amdgcn_ballot_w64(!busy);
        const uint32_t n_idle = (uint32_t)__builtin_popcountll(idle);
        if (next < n && n_idle >= PT_REFILL_LANES) {
            const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(idle >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)idle, 0u));
            const uint32_t k = next + rank;
            if (!busy && k < n) {
                mine = base + k;
                const double2 a = rays.a[mine], b = rays.b[mine], c = rays.c[mine];
                w.begin(stream, v3(a.x, a.y, b.x), v3(b.y, c.x, c.y), 0.001, __builtin_huge_val()); // ray.rs:25
                busy = true;
            }
            next = min(n, next + n_idle);
        } else if (n_idle == 64u) {
            break; // nothing in flight, nothing left to hand out
        }
        // The wave runs box tests and primitive tests as separate phases: lanes that have reached a primitive wait (a
        // ray meets ~5 nodes per primitive) until PT_LEAF_LANES of them have gathered or nobody is at a node any more.
        const unsigned long long at_node = __builtin_amdgcn_ballot_w64(busy && w.at_node());
        const unsigned long long at_leaf = __builtin_amdgcn_ballot_w64(busy && !w.at_node());
        if (at_node != 0ull && (uint32_t)__builtin_popcountll(at_leaf) < PT_LEAF_LANES) {
            // a few box tests per round: the bookkeeping above costs as much as one of them
#pragma unroll
            for (int r = 0; r < PT_NODE_STEPS; ++r)
                if (busy && w.at_node() && w.off < stream_bytes) w.step_node(stream);
        } else {
            if (busy && !w.at_node()) w.step_leaf(stream);
        }
        if (busy && w.off >= stream_bytes) {
            hit_t[mine] = w.result_t();
            hit_prim[mine] = w.best_prim;
            busy = false;
        }
    }
}

#ifndef PT_ORDERED_WAVES
#define PT_ORDERED_WAVES 5
#endif
template <bool DEEP> // (pt_ordered.h: the wave loop's thresholds for scenes with deep trees)
__global__ __launch_bounds__(PT_BLOCK) __attribute__((amdgpu_waves_per_eu(PT_ORDERED_WAVES)))
void trace_ordered_kernel(OrderedScene sc, RayStream rays, const uint32_t* __restrict__ counts, uint32_t n_waves, double* __restrict__ hit_t,
                          int32_t* __restrict__ hit_prim) {
    __shared__ uint32_t s_stack[PT_GUIDE_STACK_SLOTS * PT_BLOCK];
    const uint32_t wave = blockIdx.x * (blockDim.x / PT_WAVE) + (threadIdx.x >> 6);
    const uint32_t n = __builtin_amdgcn_readfirstlane(counts[wave]);
    RegionRayIo io{rays, hit_t, hit_prim, (size_t)wave * counts[n_waves]};
    ordered_wave_loop<true, DEEP>(sc, s_stack + threadIdx.x, n, 0.001, __builtin_huge_val(), io); // ray.rs:25
}

void launch_trace(hipStream_t s, const LaunchGeom& g, const DevScene& sc, const Wavefront& wf, const BounceIO& io) {
    const RayStream& in = wf.rays[io.in];
    if (sc.trace_mode == PT_TRACE_ORDERED) {
        if (sc.deep_tree)
            hipLaunchKernelGGL(trace_ordered_kernel<true>, dim3(g.grid), dim3(PT_BLOCK), 0, s, ordered_scene(sc), in, io.counts_in, wf.n_waves,
                               wf.hit_t, wf.hit_prim);
        else
            hipLaunchKernelGGL(trace_ordered_kernel<false>, dim3(g.grid), dim3(PT_BLOCK), 0, s, ordered_scene(sc), in, io.counts_in, wf.n_waves,
                               wf.hit_t, wf.hit_prim);
        return;
    }
    if (sc.trace_mode == PT_TRACE_STREAM) {
        hipLaunchKernelGGL(trace_stream_kernel, dim3(g.grid), dim3(PT_BLOCK), 0, s, sc.walk, sc.walk_bytes, in, io.counts_in, wf.n_waves,
                           wf.hit_t, wf.hit_prim);
        return;
    }
    if (sc.trace_mode == PT_TRACE_LOCKSTEP) {
        hipLaunchKernelGGL(trace_walk_kernel, dim3(g.grid), dim3(PT_BLOCK), 0, s, sc.walk, sc.walk_bytes, sc.walk_flags, in, io.counts_in, wf.n_waves,
                           wf.hit_t, wf.hit_prim);
        return;
    }
    if (g.all_lds)
        hipLaunchKernelGGL(trace_kernel<true>, dim3(g.grid), dim3(PT_BLOCK), g.trace_lds_bytes, s, sc, in, io.counts_in,
                           wf.n_waves, wf.hit_t, wf.hit_prim);
    else
        hipLaunchKernelGGL(trace_kernel<false>, dim3(g.grid), dim3(PT_BLOCK), g.trace_lds_bytes, s, sc, in, io.counts_in,
                           wf.n_waves, wf.hit_t, wf.hit_prim);
}

// Parity entry: the same traversal on caller-supplied rays (AoS, as the C ABI takes them), full record out.
template <bool ALL_LDS>
__global__ __launch_bounds__(PT_BLOCK) void trace_batch_kernel(DevScene sc, const double* origins, const double* dirs,
                                                               uint32_t n, double t_min, double t_max, PtHitOut* out) {
    extern __shared__ __align__(16) unsigned char smem[];
    stage_scene(sc, smem);
    const DevNode* lds_nodes = reinterpret_cast<const DevNode*>(smem);
    const DevPrim* lds_prims = reinterpret_cast<const DevPrim*>(smem + (size_t)sc.lds_nodes * sizeof(DevNode));
    uint32_t* stack = reinterpret_cast<uint32_t*>(smem + (size_t)sc.lds_nodes * sizeof(DevNode) +
                                                  (size_t)sc.lds_prims * sizeof(DevPrim)) + threadIdx.x;
    const uint32_t stride = gridDim.x * blockDim.x;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        V3 o = v3(origins[3 * (size_t)i], origins[3 * (size_t)i + 1], origins[3 * (size_t)i + 2]);
        V3 d = v3(dirs[3 * (size_t)i], dirs[3 * (size_t)i + 1], dirs[3 * (size_t)i + 2]);
        double t;
        int32_t prim;
        traverse<ALL_LDS>(sc, lds_nodes, lds_prims, stack, o, d, t_min, t_max, t, prim);
        PtHitOut h;
        h.prim = prim;
        if (prim >= 0) {
            const DevPrim* pr = sc.prims + prim;
            V3 p = o + t * d; // ray.rs:16-18
            V3 n_out = outward_normal(sc, pr, (uint32_t)prim, p);
            bool front = dot(d, n_out) < 0.0; // geometry.rs:29
            V3 nn = front ? n_out : neg(n_out);
            h.front_face = front ? 1u : 0u;
            h.t = t;
            h.normal[0] = nn.x; h.normal[1] = nn.y; h.normal[2] = nn.z;
        } else {
            h.front_face = 0u;
            h.t = 0.0;
            h.normal[0] = h.normal[1] = h.normal[2] = 0.0;
        }
        out[i] = h;
    }
}

__device__ __forceinline__ void write_hit_record(const DevScene& sc, V3 o, V3 d, double t, int32_t prim, PtHitOut* out) {
    PtHitOut h;
    if (prim >= (int32_t)sc.n_prims) prim = -1; // cannot happen; keeps the table reads below in range regardless
    h.prim = prim;
    if (prim >= 0) {
        const DevPrim* pr = sc.prims + prim;
        V3 p = o + t * d; // ray.rs:16-18
        V3 n_out = outward_normal(sc, pr, (uint32_t)prim, p);
        bool front = dot(d, n_out) < 0.0; // geometry.rs:29
        V3 nn = front ? n_out : neg(n_out);
        h.front_face = front ? 1u : 0u;
        h.t = t;
        h.normal[0] = nn.x; h.normal[1] = nn.y; h.normal[2] = nn.z;
    } else {
        h.front_face = 0u;
        h.t = 0.0;
        h.normal[0] = h.normal[1] = h.normal[2] = 0.0;
    }
    *out = h;
}

__global__ __launch_bounds__(PT_BLOCK) void trace_batch_walk_kernel(DevScene sc, const double* __restrict__ origins,
                                                                    const double* __restrict__ dirs, uint32_t n, double t_min,
                                                                    double t_max, PtHitOut* __restrict__ out) {
    const uint32_t stride = gridDim.x * blockDim.x;
    for (uint32_t base = blockIdx.x * blockDim.x + (threadIdx.x & ~63u); base < n; base += stride) {
        const uint32_t i = base + (threadIdx.x & 63u);
        const bool live = i < n;
        V3 o = v3(0x1p40, 0x1p40, 0x1p40), d = v3(1, 1, 1); // idle lanes: an ordinary ray (see trace_walk_kernel)
        if (live) {
            o = v3(origins[3 * (size_t)i], origins[3 * (size_t)i + 1], origins[3 * (size_t)i + 2]);
            d = v3(dirs[3 * (size_t)i], dirs[3 * (size_t)i + 1], dirs[3 * (size_t)i + 2]);
        }
        double t;
        int32_t prim;
        traverse_walk(sc.walk, sc.walk_bytes, sc.walk_flags, live, o, d, t_min, t_max, t, prim);
        if (live) write_hit_record(sc, o, d, t, prim, out + i);
    }
}

__global__ __launch_bounds__(PT_BLOCK) void trace_batch_stream_kernel(DevScene sc, const double* __restrict__ origins,
                                                                      const double* __restrict__ dirs, uint32_t n, double t_min,
                                                                      double t_max, PtHitOut* __restrict__ out) {
    const uint32_t stride = gridDim.x * blockDim.x;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        V3 o = v3(origins[3 * (size_t)i], origins[3 * (size_t)i + 1], origins[3 * (size_t)i + 2]);
        V3 d = v3(dirs[3 * (size_t)i], dirs[3 * (size_t)i + 1], dirs[3 * (size_t)i + 2]);
        StreamWalk w;
        w.begin(sc.walk, o, d, t_min, t_max);
        while (w.off < sc.walk_bytes) {
            if (w.at_node()) w.step_node(sc.walk);
            else w.step_leaf(sc.walk);
        }
        write_hit_record(sc, o, d, w.result_t(), w.best_prim, out + i);
    }
}

struct BatchRayIo { // a slice of the caller's arrays (AoS, as the C ABI takes them), full hit record out
    const DevScene* sc;
    const double* origins;
    const double* dirs;
    PtHitOut* out;
    size_t first;
    __device__ __forceinline__ void load(uint32_t k, V3& o, V3& d) const {
        const size_t i = first + k;
        o = v3(origins[3 * i], origins[3 * i + 1], origins[3 * i + 2]);
        d = v3(dirs[3 * i], dirs[3 * i + 1], dirs[3 * i + 2]);
    }
    __device__ __forceinline__ void store(uint32_t k, V3 o, V3 d, double t, int32_t prim) const { write_hit_record(*sc, o, d, t, prim, out + first + k); }
};

__global__ __launch_bounds__(PT_BLOCK) void trace_batch_ordered_kernel(DevScene sc, const double* __restrict__ origins,
                                                                       const double* __restrict__ dirs, uint32_t n, double t_min,
                                                                       double t_max, PtHitOut* __restrict__ out) {
    __shared__ uint32_t s_stack[PT_GUIDE_STACK_SLOTS * PT_BLOCK];
    const OrderedScene os = ordered_scene(sc);
    const uint32_t waves = gridDim.x * (blockDim.x / PT_WAVE), wave = blockIdx.x * (blockDim.x / PT_WAVE) + (threadIdx.x >> 6);
    const uint32_t per_wave = (n + waves - 1u) / waves; // rays [wave * per_wave, ...) belong to this wave
    const uint64_t first = (uint64_t)wave * per_wave;
    const uint32_t mine = first >= n ? 0u : (uint32_t)min((uint64_t)per_wave, (uint64_t)n - first);
    BatchRayIo io{&sc, origins, dirs, out, (size_t)first};
    ordered_wave_loop(os, s_stack + threadIdx.x, mine, t_min, t_max, io);
}

void launch_trace_batch(hipStream_t s, const LaunchGeom& g, const DevScene& sc, const double* origins, const double* dirs,
                        uint32_t n, double t_min, double t_max, PtHitOut* out) {
    uint32_t blocks = (n + PT_BLOCK - 1) / PT_BLOCK;
    if (blocks > 2048u) blocks = 2048u;
    if (blocks == 0) blocks = 1;
    if (sc.trace_mode == PT_TRACE_ORDERED) {
        hipLaunchKernelGGL(trace_batch_ordered_kernel, dim3(blocks), dim3(PT_BLOCK), 0, s, sc, origins, dirs, n, t_min, t_max, out);
        return;
    }
    if (sc.trace_mode == PT_TRACE_STREAM) {
        hipLaunchKernelGGL(trace_batch_stream_kernel, dim3(blocks), dim3(PT_BLOCK), 0, s, sc, origins, dirs, n, t_min, t_max, out);
        return;
    }
    if (sc.trace_mode == PT_TRACE_LOCKSTEP) {
        hipLaunchKernelGGL(trace_batch_walk_kernel, dim3(blocks), dim3(PT_BLOCK), 0, s, sc, origins, dirs, n, t_min, t_max, out);
        return;
    }
    if (g.all_lds)
        hipLaunchKernelGGL(trace_batch_kernel<true>, dim3(blocks), dim3(PT_BLOCK), g.trace_lds_bytes, s, sc, origins, dirs, n,
                           t_min, t_max, out);
    else
        hipLaunchKernelGGL(trace_batch_kernel<false>, dim3(blocks), dim3(PT_BLOCK), g.trace_lds_bytes, s, sc, origins, dirs, n,
                           t_min, t_max, out);
}

#ifdef PT_DEBUG_PHASES
extern "C" int pt_debug_read_phases(unsigned long long* out, int reset) { // 8 x (rounds, active lanes): refill, verify, guide step 0..3, leaf, literal; then guide steps with no child hit, with both hit, taken with a hit in hand, both
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(pt_debug_phase), sizeof(pt_debug_phase)) != hipSuccess) return -1;
    if (reset) {
        unsigned long long zero[32] = {0};
        if (hipMemcpyToSymbol(HIP_SYMBOL(pt_debug_phase), zero, sizeof(zero)) != hipSuccess) return -1;
    }
    return 0;
}
#endif
