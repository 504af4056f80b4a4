// ptgpu.hip -- implementation of the C ABI in include/ptgpu.h: context, scene upload, the wavefront driver.
#include <dlfcn.h>

#include <algorithm>
#include <cstdio>
#include <cstring>
#include <limits>
#include <map>
#include <mutex>
#include <string>
#include <vector>

#include <rccl/rccl.h> // types and prototypes only: the library is loaded on the first pt_accum_reduce (dlopen)

#include "guide_tree.h"
#include "pt_launch.h"

// guide_build.hip: the guide tree built on the device (option "guide_build")
hipError_t guide_build_on_device(hipStream_t stream, const std::vector<GuideLeaf>& leaves, GuideNode* d_nodes, const char** reason);

#define PT_MAX_LANES 8
#ifndef PT_TAIL_FROM_DEFAULT
#define PT_TAIL_FROM_DEFAULT 4u
#endif
#ifndef PT_TAIL_FROM_LOCKSTEP
#define PT_TAIL_FROM_LOCKSTEP 4096u /* lockstep scenes: never, unless "tail_from" says so */
#endif
#ifndef PT_GUIDE_SPLIT_DEPTH
#define PT_GUIDE_SPLIT_DEPTH 5 /* at most 32 references per triangle */
#endif
#ifndef PT_TAIL_MIN_PRIMS
#define PT_TAIL_MIN_PRIMS 65536u
#endif

// Staging area of pt_trace_batch_static (ptgpu.h): the device copies of one batch's rays and hit records.  Every GPU the
// library runs on holds its own copy of this array at its own address for the life of the process.  Host code never uses
// the array's name as an address: scratch_for_device() asks the runtime for this device's address once and keeps it.
#define PT_STATIC_BYTES_PER_RAY 88u /* origin 24 + direction 24 + PtHitOut 40 */
static __device__ __align__(256) unsigned char scratch[(size_t)PT_TRACE_STATIC_MAX_RAYS * PT_STATIC_BYTES_PER_RAY];

namespace {
thread_local std::string g_last_error;

struct DevBuf {
    void* p = nullptr;
    size_t bytes = 0;
};
}  // namespace

struct PtContext {
    int device = 0;
    hipStream_t stream = nullptr;   // lane 0's stream; also every non-render call
    hipStream_t extra_streams[PT_MAX_LANES - 1] = {};  // the other lanes' streams
    std::string err;
    int cu_count = 256;

    // scene
    bool has_scene = false;
    DevScene sc{};
    DevBuf d_nodes, d_prims, d_aux, d_mats, d_texs, d_perlin, d_walk, d_shade, d_guide;
    LaunchGeom geom{};

    // accumulation
    uint32_t width = 0, height = 0;
    DevBuf d_accum_own;
    float* d_accum = nullptr;
    DevBuf d_rgb;

    // wavefront streams: two independent lanes (own buffers, own HIP stream) so that one lane's VALU-bound trace
    // overlaps the other lane's memory-bound shade
    struct Lane {
        hipStream_t stream = nullptr;
        size_t capacity = 0;
        size_t paths = 0;
        uint32_t counts_cap = 0;
        Wavefront wf{};
        std::vector<DevBuf> bufs;
    };
    Lane lanes[PT_MAX_LANES];
    int n_lanes = 0;  // 0 = chosen per call from the amount of work
    int lane_sync = 0;
    int lane_full_grid = 0;

    // options
    int64_t paths_per_pass = (int64_t)1 << 29;
    int profile = 1;
    int blocks_per_cu_override = 0;
    int trace_mode_override = -1; // PT_TRACE_* or -1 = by scene size
    int guide_split_ratio = 4;    // guide tree: a triangle whose box exceeds this many times its own projected areas is referenced per piece; 0 = never
    int tail_waves_div = 2;       // ... over n_waves / this many of them (half: 1M mesh 64 spp 34.1 -> 32.9 ms; a third or a quarter: slower)
    int deep_tree = -1;           // ordered walk's wave loop: thresholds for thin waves (next upload); -1 = by scene size
    int tail_deal = 1;            // the rays that enter the per-wave tail launch are first dealt out evenly over the waves
    int tail_from = -1;           // first bounce of the per-wave tail launch (ordered walk); -1 = automatic, >= max_bounces = never
    unsigned char* scratch_ptr = nullptr; // this device's address of `scratch` (hipGetSymbolAddress at context creation)
    int guide_build_device = 0;   // 1: the ordered walk's guide tree is built on the device (guide_build.hip) instead of on the host
    int fused = 0;                // one path kernel per pass where it applies (measured: not faster, see path.hip)
    int repack_every = 0, repack_first = 8;  // measured on Cornell 1080p: the copy costs more than the fuller waves save
    std::vector<hipEvent_t> events;

    int fail(int code, const std::string& msg) {
        err = msg;
        g_last_error = msg;
        return code;
    }
    int fail_hip(const char* what, hipError_t e) {
        return fail(e == hipErrorOutOfMemory ? PT_ERR_OOM : PT_ERR_HIP, std::string(what) + ": " + hipGetErrorString(e));
    }
};

#define PT_HIP(ctx, call)                                       \
    do {                                                        \
        hipError_t e_ = (call);                                 \
        if (e_ != hipSuccess) return (ctx)->fail_hip(#call, e_); \
    } while (0)

namespace {

// Address of `scratch` on `device`, which must be the current device: looked up once per device, then remembered.
std::mutex g_scratch_mutex;
unsigned char* g_scratch_ptr[256] = {};
// One call at a time per GPU may use that GPU's array (pt_trace_batch_static holds this for its whole duration): two host
// threads driving two contexts on one GPU then take turns instead of overwriting each other's rays and hit records.
std::mutex g_scratch_in_use[256];
hipError_t scratch_for_device(int device, unsigned char** out) {
    std::lock_guard<std::mutex> lock(g_scratch_mutex);
    const bool cached = device >= 0 && device < 256;
    if (cached && g_scratch_ptr[device]) {
        *out = g_scratch_ptr[device];
        return hipSuccess;
    }
    void* p = nullptr;
    hipError_t e = hipGetSymbolAddress(&p, HIP_SYMBOL(scratch));
    if (e != hipSuccess) return e;
    if (!p) return hipErrorInvalidSymbol;
    if (cached) g_scratch_ptr[device] = (unsigned char*)p;
    *out = (unsigned char*)p;
    return hipSuccess;
}

void release_cliques_of(PtContext* ctx); // below, with pt_accum_reduce: the RCCL communicators made for ctx

void free_buf(DevBuf& b) {
    if (b.p) (void)hipFree(b.p);
    b.p = nullptr;
    b.bytes = 0;
}
hipError_t ensure_buf(DevBuf& b, size_t bytes) {
    if (b.bytes >= bytes && b.p) return hipSuccess;
    free_buf(b);
    if (bytes == 0) bytes = 16;
    hipError_t e = hipMalloc(&b.p, bytes);
    if (e == hipSuccess) b.bytes = bytes;
    return e;
}

void free_lane(PtContext::Lane& ln) {
    for (auto& b : ln.bufs) free_buf(b);
    ln.bufs.clear();
    ln.wf = Wavefront{};
    ln.capacity = 0;
    ln.paths = 0;
    ln.counts_cap = 0;
}
void free_wavefront(PtContext* ctx) {
    for (auto& ln : ctx->lanes) free_lane(ln);
}

// Streams of one lane for n_paths paths in flight over a persistent grid of `grid` blocks, and per-wave counts for
// n_bounces + 1 bounces.
int ensure_lane(PtContext* ctx, PtContext::Lane& ln, uint32_t grid, size_t n_paths, uint32_t n_bounces) {
    const uint32_t n_waves = grid * (PT_BLOCK / PT_WAVE);
    const size_t chunks = (n_paths + 63) / 64;
    const size_t region = ((chunks + n_waves - 1) / n_waves) * 64;
    const size_t capacity = region * n_waves;
    const uint32_t n_counts = n_bounces + 1;
    if (region > 0xFFFFFFFFull) return ctx->fail(PT_ERR_BAD_ARG, "too many paths per pass");
    if (ln.capacity >= capacity && ln.paths >= n_paths && ln.counts_cap >= n_counts && ln.wf.n_waves == n_waves) {
        ln.wf.region = (uint32_t)region;
        return PT_OK;
    }
    free_lane(ln);
    const size_t sizes[] = {
        capacity * sizeof(double2), capacity * sizeof(double2), capacity * sizeof(double2),  // rays[0]
        capacity * sizeof(double2), capacity * sizeof(double2), capacity * sizeof(double2),  // rays[1]
        capacity * sizeof(uint4),   capacity * sizeof(uint4),                               // state
        capacity * sizeof(double),  capacity * sizeof(int32_t),                             // hits
        n_paths * sizeof(float4),                                                           // plane (by path slot)
        (size_t)n_counts * (n_waves + 1) * sizeof(uint32_t), (size_t)n_counts * sizeof(unsigned long long),  // counts, totals
        (size_t)n_counts * (n_waves + 1) * sizeof(uint32_t)                                                  // repacked counts
    };
    const size_t nb = sizeof(sizes) / sizeof(sizes[0]);
    ln.bufs.resize(nb);
    for (size_t i = 0; i < nb; ++i) {
        hipError_t e = ensure_buf(ln.bufs[i], sizes[i]);
        if (e != hipSuccess) {
            free_lane(ln);
            return ctx->fail_hip("hipMalloc(wavefront streams)", e);
        }
    }
    Wavefront& w = ln.wf;
    w.rays[0] = RayStream{(double2*)ln.bufs[0].p, (double2*)ln.bufs[1].p, (double2*)ln.bufs[2].p};
    w.rays[1] = RayStream{(double2*)ln.bufs[3].p, (double2*)ln.bufs[4].p, (double2*)ln.bufs[5].p};
    w.state[0] = (uint4*)ln.bufs[6].p;
    w.state[1] = (uint4*)ln.bufs[7].p;
    w.hit_t = (double*)ln.bufs[8].p;
    w.hit_prim = (int32_t*)ln.bufs[9].p;
    w.plane = (float4*)ln.bufs[10].p;
    w.counts = (uint32_t*)ln.bufs[11].p;
    w.totals = (unsigned long long*)ln.bufs[12].p;
    w.counts_repacked = (uint32_t*)ln.bufs[13].p;
    w.n_waves = n_waves;
    w.region = (uint32_t)region;
    ln.capacity = capacity;
    ln.paths = n_paths;
    ln.counts_cap = n_counts;
    return PT_OK;
}

// Walks the tree the way the reference recursion does, checking that it is a tree over valid indices and
// that the primitives appear in depth-first order.  Returns the deepest nesting of nodes.
struct TreeCheck {
    const PtSceneDesc* s;
    std::vector<uint8_t> seen;
    uint32_t next_prim = 0;
    uint32_t max_depth = 0;
    std::string error;
    bool walk() {
        // explicit stack: (ref, depth); children pushed right first so the left is visited first
        std::vector<std::pair<uint32_t, uint32_t>> st;
        st.emplace_back(s->root, 1u);
        while (!st.empty()) {
            auto [ref, depth] = st.back();
            st.pop_back();
            if (ref & PT_REF_LEAF) {
                uint32_t first = ref & PT_REF_INDEX_MASK;
                uint32_t count = (ref & PT_REF_GROUP6) ? 6u : 1u;
                if (first != next_prim) {
                    error = "primitives are not listed in depth-first order";
                    return false;
                }
                if ((uint64_t)first + count > s->n_prims) {
                    error = "leaf reference past the primitive table";
                    return false;
                }
                if (count == 6u) {
                    for (uint32_t k = 0; k < 6; ++k) {
                        uint32_t t = s->prims[first + k].type;
                        if (t != PT_PRIM_RECT_XY && t != PT_PRIM_RECT_XZ && t != PT_PRIM_RECT_YZ) {
                            error = "a group of six must hold rectangles (AABox sides)";
                            return false;
                        }
                    }
                }
                next_prim = first + count;
                continue;
            }
            if (ref >= s->n_nodes) {
                error = "node reference out of range";
                return false;
            }
            if (seen[ref]) {
                error = "node referenced twice: not a tree";
                return false;
            }
            seen[ref] = 1;
            max_depth = std::max(max_depth, depth);
            st.emplace_back(s->nodes[ref].right, depth + 1);
            st.emplace_back(s->nodes[ref].left, depth + 1);
        }
        if (next_prim != s->n_prims) {
            error = "tree does not reference every primitive";
            return false;
        }
        return true;
    }
};

}  // namespace

extern "C" {

int pt_device_count(void) {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) {
        g_last_error = std::string("hipGetDeviceCount: ") + hipGetErrorString(e);
        return e == hipErrorNoDevice ? 0 : PT_ERR_HIP;
    }
    return n;
}

int pt_context_create(int device, PtContext** out) {
    if (!out) {
        g_last_error = "pt_context_create: out is NULL";
        return PT_ERR_BAD_ARG;
    }
    *out = nullptr;
    int n = pt_device_count();
    if (n <= 0) {
        if (n == 0) g_last_error = "no HIP device visible";
        return PT_ERR_NO_DEVICE;
    }
    if (device < 0 || device >= n) {
        g_last_error = "pt_context_create: device index out of range";
        return PT_ERR_BAD_ARG;
    }
    hipError_t e = hipSetDevice(device);
    if (e != hipSuccess) {
        g_last_error = std::string("hipSetDevice: ") + hipGetErrorString(e);
        return PT_ERR_HIP;
    }
    PtContext* ctx = new PtContext();
    ctx->device = device;
    e = hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking);
    if (e != hipSuccess) {
        g_last_error = std::string("hipStreamCreate: ") + hipGetErrorString(e);
        delete ctx;
        return PT_ERR_HIP;
    }
    ctx->lanes[0].stream = ctx->stream;
    for (int l = 1; l < PT_MAX_LANES; ++l) {
        e = hipStreamCreateWithFlags(&ctx->extra_streams[l - 1], hipStreamNonBlocking);
        if (e != hipSuccess) {
            g_last_error = std::string("hipStreamCreate: ") + hipGetErrorString(e);
            for (int k = 0; k < l - 1; ++k) (void)hipStreamDestroy(ctx->extra_streams[k]);
            (void)hipStreamDestroy(ctx->stream);
            delete ctx;
            return PT_ERR_HIP;
        }
        ctx->lanes[l].stream = ctx->extra_streams[l - 1];
    }
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) == hipSuccess && prop.multiProcessorCount > 0) ctx->cu_count = prop.multiProcessorCount;
    e = scratch_for_device(device, &ctx->scratch_ptr); // (this device is current: hipSetDevice above)
    if (e != hipSuccess) {
        g_last_error = std::string("hipGetSymbolAddress(scratch): ") + hipGetErrorString(e);
        pt_context_destroy(ctx);
        return PT_ERR_HIP;
    }
    *out = ctx;
    return PT_OK;
}

void pt_context_destroy(PtContext* ctx) {
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
    for (hipStream_t st : ctx->extra_streams)
        if (st) (void)hipStreamSynchronize(st);
    release_cliques_of(ctx); // communicators of every pt_accum_reduce clique this context was part of (its stream has drained)
    for (hipEvent_t ev : ctx->events) (void)hipEventDestroy(ev);
    free_wavefront(ctx);
    free_buf(ctx->d_nodes);
    free_buf(ctx->d_prims);
    free_buf(ctx->d_aux);
    free_buf(ctx->d_mats);
    free_buf(ctx->d_texs);
    free_buf(ctx->d_perlin);
    free_buf(ctx->d_walk);
    free_buf(ctx->d_guide);
    free_buf(ctx->d_shade);
    free_buf(ctx->d_accum_own);
    free_buf(ctx->d_rgb);
    if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
    for (hipStream_t st : ctx->extra_streams)
        if (st) (void)hipStreamDestroy(st);
    delete ctx;
}

const char* pt_last_error(const PtContext* ctx) { return ctx ? ctx->err.c_str() : g_last_error.c_str(); }

int pt_context_set_option(PtContext* ctx, const char* key, int64_t value) {
    if (!ctx || !key) return PT_ERR_BAD_ARG;
    std::string k(key);
    if (k == "paths_per_pass") {
        if (value < 1 || value > ((int64_t)1 << 31)) return ctx->fail(PT_ERR_BAD_ARG, "paths_per_pass out of range [1, 2^31]");
        ctx->paths_per_pass = value;
    } else if (k == "profile") {
        ctx->profile = value ? 1 : 0;
    } else if (k == "lanes") {  // 2: two half-size sub-passes on two streams, trace of one overlapping shade of the other; 1: serial
        if (value < 0 || value > PT_MAX_LANES) return ctx->fail(PT_ERR_BAD_ARG, "lanes must be 0 (automatic) or 1..8");
        ctx->n_lanes = (int)value;
    } else if (k == "lane_sync") {  // 1: a lane's step waits for the other lane's previous step (lock-step pairing); 0: free-running
        ctx->lane_sync = value ? 1 : 0;
    } else if (k == "lane_full_grid") {  // 1: each lane launches the full persistent grid instead of half of it
        ctx->lane_full_grid = value ? 1 : 0;
    } else if (k == "repack_every") {  // re-deal the surviving rays over the waves every N bounces (0 = never)
        if (value < 0 || value > 4096) return ctx->fail(PT_ERR_BAD_ARG, "repack_every out of range");
        ctx->repack_every = (int)value;
    } else if (k == "repack_first") {  // first bounce before which a repack runs
        if (value < 1 || value > 4096) return ctx->fail(PT_ERR_BAD_ARG, "repack_first out of range");
        ctx->repack_first = (int)value;
    } else if (k == "fused") {  // 1: the whole bounce loop in one launch where the lockstep walk applies; 0: a trace and a shade launch per bounce
        ctx->fused = value ? 1 : 0;
    } else if (k == "guide_split") {  // guide tree: split references for thin diagonal triangles (guide_tree.h); 0 off, else the area ratio that triggers a cut
        if (value < 0 || value > 1000000) return ctx->fail(PT_ERR_BAD_ARG, "guide_split out of range [0, 1000000]");
        ctx->guide_split_ratio = (int)value;
    } else if (k == "tail_waves_div") {
        if (value < 1 || value > 64) return ctx->fail(PT_ERR_BAD_ARG, "tail_waves_div out of range [1, 64]");
        ctx->tail_waves_div = (int)value;
    } else if (k == "deep_tree") {
        if (value < -1 || value > 1) return ctx->fail(PT_ERR_BAD_ARG, "deep_tree must be -1 (by scene size), 0 or 1");
        ctx->deep_tree = (int)value;
    } else if (k == "tail_deal") {
        if (value < 0 || value > 1) return ctx->fail(PT_ERR_BAD_ARG, "tail_deal must be 0 or 1");
        ctx->tail_deal = (int)value;
    } else if (k == "tail_from") {  // ordered walk: bounces from this one on run in one per-wave launch (path.hip); -1 automatic
        if (value < -1 || value > 4096) return ctx->fail(PT_ERR_BAD_ARG, "tail_from out of range [-1, 4096]");
        ctx->tail_from = (int)value;
    } else if (k == "trace_mode") {  // -1 automatic, 0 lockstep walk, 1 per-lane stack (LDS), 2 per-lane record stream, 3 ordered; next upload
        if (value < -1 || value > 3) return ctx->fail(PT_ERR_BAD_ARG, "trace_mode out of range [-1, 3]");
        ctx->trace_mode_override = (int)value;
    } else if (k == "guide_build") {  // 0: host, binned surface-area splits; 1: device, Morton sort + Karras hierarchy; next upload
        if (value < 0 || value > 1) return ctx->fail(PT_ERR_BAD_ARG, "guide_build must be 0 (host) or 1 (device)");
        ctx->guide_build_device = (int)value;
    } else if (k == "blocks_per_cu") {  // tuning knob: persistent-grid blocks (of 4 waves) per CU; takes effect at the next upload
        if (value < 0 || value > 16) return ctx->fail(PT_ERR_BAD_ARG, "blocks_per_cu out of range [0, 16]");
        ctx->blocks_per_cu_override = (int)value;
    } else {
        return ctx->fail(PT_ERR_BAD_ARG, "unknown option " + k);
    }
    return PT_OK;
}

int pt_scene_upload(PtContext* ctx, const PtSceneDesc* s) {
    if (!ctx) return PT_ERR_BAD_ARG;
    if (!s || !s->nodes || !s->prims || !s->materials) return ctx->fail(PT_ERR_BAD_ARG, "pt_scene_upload: null table");
    // bvh.rs:76: BvhNode::new panics on fewer than two objects, so every world has a root node
    if (s->n_nodes < 1 || s->n_prims < 2) return ctx->fail(PT_ERR_BAD_ARG, "creating BVH node from empty object list");
    if (s->n_prims > PT_REF_INDEX_MASK || s->n_nodes > PT_REF_INDEX_MASK) return ctx->fail(PT_ERR_BAD_ARG, "scene too large");
    if (s->root & PT_REF_LEAF) return ctx->fail(PT_ERR_BAD_ARG, "root must be a node");
    if (s->n_materials < 1) return ctx->fail(PT_ERR_BAD_ARG, "no materials");
    if (s->n_textures > 0 && !s->textures) return ctx->fail(PT_ERR_BAD_ARG, "null texture table");
    if (s->n_perlin_tables > 0 && !s->perlin_tables) return ctx->fail(PT_ERR_BAD_ARG, "null perlin tables");

    TreeCheck tc{s, std::vector<uint8_t>(s->n_nodes, 0)};
    if (!tc.walk()) return ctx->fail(PT_ERR_BAD_ARG, "pt_scene_upload: " + tc.error);
    for (uint32_t i = 0; i < s->n_nodes; ++i)
        if (!tc.seen[i]) return ctx->fail(PT_ERR_BAD_ARG, "pt_scene_upload: unreachable node");
    if (tc.max_depth > 56) return ctx->fail(PT_ERR_BAD_ARG, "pt_scene_upload: tree deeper than 56 levels");

    // textures: types, children, and no cycles among checkers
    for (uint32_t i = 0; i < s->n_textures; ++i) {
        const PtTexture& t = s->textures[i];
        if (t.type > PT_TEX_PERLIN) return ctx->fail(PT_ERR_BAD_ARG, "bad texture type");
        if (t.type == PT_TEX_CHECKER && (t.odd >= s->n_textures || t.even >= s->n_textures))
            return ctx->fail(PT_ERR_BAD_ARG, "checker child out of range");
        if (t.type == PT_TEX_PERLIN && t.perlin_table >= s->n_perlin_tables)
            return ctx->fail(PT_ERR_BAD_ARG, "perlin table out of range");
    }
    {
        std::vector<int> state(s->n_textures, 0);  // 0 new, 1 on path, 2 done
        for (uint32_t root = 0; root < s->n_textures; ++root) {
            if (state[root]) continue;
            std::vector<std::pair<uint32_t, int>> st{{root, 0}};
            state[root] = 1;
            while (!st.empty()) {
                auto& [t, step] = st.back();
                const PtTexture& tx = s->textures[t];
                if (tx.type != PT_TEX_CHECKER || step == 2) {
                    state[t] = 2;
                    st.pop_back();
                    continue;
                }
                uint32_t child = step == 0 ? tx.odd : tx.even;
                ++step;
                if (state[child] == 1) return ctx->fail(PT_ERR_BAD_ARG, "texture graph has a cycle");
                if (state[child] == 0) {
                    state[child] = 1;
                    st.emplace_back(child, 0);
                }
            }
        }
    }
    for (uint32_t i = 0; i < s->n_materials; ++i) {
        const PtMaterial& m = s->materials[i];
        if (m.type > PT_MAT_DIFFUSE_LIGHT) return ctx->fail(PT_ERR_BAD_ARG, "bad material type");
        if (m.type != PT_MAT_DIELECTRIC && m.texture >= s->n_textures) return ctx->fail(PT_ERR_BAD_ARG, "material texture out of range");
    }

    // device primitive records
    std::vector<DevPrim> prims(s->n_prims);
    std::vector<double> aux((size_t)s->n_prims * 3, 0.0);
    for (uint32_t i = 0; i < s->n_prims; ++i) {
        const PtPrim& p = s->prims[i];
        DevPrim& q = prims[i];
        std::memset(&q, 0, sizeof(q));
        if (p.material >= s->n_materials) return ctx->fail(PT_ERR_BAD_ARG, "primitive material out of range");
        q.type = p.type;
        q.material = p.material;
        switch (p.type) {
            case PT_PRIM_SPHERE:
                for (int k = 0; k < 4; ++k) q.d[k] = p.d[k];
                break;
            case PT_PRIM_RECT_XY:
            case PT_PRIM_RECT_XZ:
            case PT_PRIM_RECT_YZ: {
                for (int k = 0; k < 5; ++k) q.d[k] = p.d[k];
                int axis = p.type == PT_PRIM_RECT_XY ? 2 : (p.type == PT_PRIM_RECT_XZ ? 1 : 0);
                aux[(size_t)i * 3 + axis] = p.d[5];  // geometry.rs:227 / :290 / :353
                break;
            }
            case PT_PRIM_TRIANGLE:
                for (int k = 0; k < 3; ++k) {
                    q.d[k] = p.d[k];
                    q.d[3 + k] = p.d[3 + k] - p.d[k];  // point2 - point1, geometry.rs:521
                    q.d[6 + k] = p.d[6 + k] - p.d[k];  // point3 - point1, geometry.rs:522
                    aux[(size_t)i * 3 + k] = p.d[9 + k];
                }
                break;
            default:
                return ctx->fail(PT_ERR_BAD_ARG, "bad primitive type");
        }
    }

    // breadth-first renumbering: the top levels become a prefix of the node table (staged in LDS by trace)
    std::vector<uint32_t> order;  // order[new] = old
    order.reserve(s->n_nodes);
    std::vector<uint32_t> remap(s->n_nodes, 0);
    order.push_back(s->root);
    for (size_t head = 0; head < order.size(); ++head) {
        const PtNode& nd = s->nodes[order[head]];
        if (!(nd.left & PT_REF_LEAF)) order.push_back(nd.left);
        if (!(nd.right & PT_REF_LEAF)) order.push_back(nd.right);
    }
    for (uint32_t i = 0; i < (uint32_t)order.size(); ++i) remap[order[i]] = i;
    std::vector<DevNode> nodes(s->n_nodes);
    for (uint32_t i = 0; i < s->n_nodes; ++i) {
        const PtNode& src = s->nodes[order[i]];
        DevNode& dst = nodes[i];
        std::memset(&dst, 0, sizeof(dst));
        for (int k = 0; k < 3; ++k) {
            dst.bmin[k] = src.bmin[k];
            dst.bmax[k] = src.bmax[k];
        }
        dst.left = (src.left & PT_REF_LEAF) ? src.left : remap[src.left];
        dst.right = (src.right & PT_REF_LEAF) ? src.right : remap[src.right];
    }

    // per-primitive shading records (pt_device.h ShadePrim)
    std::vector<ShadePrim> shade_prims(s->n_prims);
    for (uint32_t i = 0; i < s->n_prims; ++i) {
        const PtPrim& p = s->prims[i];
        const PtMaterial& m = s->materials[p.material];
        ShadePrim& q = shade_prims[i];
        std::memset(&q, 0, sizeof(q));
        if (p.type == PT_PRIM_SPHERE) {
            for (int k = 0; k < 4; ++k) q.g[k] = p.d[k];
        } else {
            for (int k = 0; k < 3; ++k) q.g[k] = aux[(size_t)i * 3 + k];
        }
        q.param = m.param;
        q.prim_type = p.type;
        q.mat_type = m.type;
        q.texture = m.texture;
        if (m.type != PT_MAT_DIELECTRIC && s->textures[m.texture].type == PT_TEX_SOLID) {
            q.tex_solid = 1;
            for (int k = 0; k < 3; ++k) q.color[k] = (float)s->textures[m.texture].color[k];
        } else if (m.type != PT_MAT_DIELECTRIC && s->textures[m.texture].type == PT_TEX_CHECKER) {
            const PtTexture& ck = s->textures[m.texture];  // children in range: checked above
            if (s->textures[ck.odd].type == PT_TEX_SOLID && s->textures[ck.even].type == PT_TEX_SOLID) {
                q.tex_solid = 2;
                for (int k = 0; k < 3; ++k) {
                    q.color[k] = (float)s->textures[ck.even].color[k];
                    q.color_odd[k] = (float)s->textures[ck.odd].color[k];
                }
            }
        }
    }

    // small scenes: the pre-order record stream of the wave-uniform walk (pt_device.h traverse_walk)
    std::vector<uint64_t> walk; // 8-byte words; a record = {kind | aux << 32} + payload doubles
    uint32_t walk_flags = PT_WALK_PLANES_IN_RANGE;
    std::vector<GuideLeaf> guide_leaves; // one per primitive record, for the ordered walk's guide tree (guide_tree.h)
    const bool large = (uint64_t)s->n_nodes + s->n_prims > PT_WALK_MAX_ITEMS;
    const bool want_guide = ctx->trace_mode_override == PT_TRACE_ORDERED || (ctx->trace_mode_override < 0 && large);
    bool boxes_nested = true;
    // Which rays the ordered walk may take (pt_device.h OrderedWalk::begin).  Its argument needs "a primitive test that
    // reports a hit is crossed inside its padded guide box" for EVERY primitive of the scene, tested or not; a ray whose
    // origin is far from a primitive relative to the primitive's size loses that to cancellation (a sphere's discriminant,
    // a triangle's barycentrics) while the padding is only 1e-5 of the leaf's largest coordinate m.  Per kind, with the
    // rounding of the tests bounded generously (constants leave a factor of 2 or more):
    //   sphere    false-hit distance ~ 4e-16 |o - c|^2 / (2|r|) <= 1e-5 |r|  if  |o|inf <= 5e4 |r| - |c|inf
    //   rectangle in-plane error ~ 2^-51 (|o| + extent) <= 1e-5 m             if  |o|inf <= 2e10 m - extent
    //   triangle  barycentric error <= 1e-15 |o - p1| |d| |e| / det with det >= 1e-4 (geometry.rs:526), times the other edge:
    //             <= 1e-5 m  if  (|o|inf + extent of the triangles) |d|inf <= 8e4 / max(|e1| |e2| / m)
    // The guide tree's box tests run in f32 (pt_device.h guide_slab): they shift a plane by up to 1.1e-14 |o|, which the
    // padding of EVERY leaf must cover: |o|inf <= 5e8 * (smallest m over the leaves); and no slab product may overflow f32:
    // coordinates and origins within 1e18 (with |1/d| <= 2^60).
    double ow_origin_limit = 1e18, ow_tri_ratio = 0.0, ow_tri_extent = 0.0, ow_rect_m = INFINITY, ow_rect_extent = 0.0;
    double ow_leaf_m_min = INFINITY, ow_leaf_m_max = 0.0;
    {
        // (corners: a triangle's three points -- a thin diagonal one is referenced once per piece, guide_tree.h split_pieces; every
        // piece is padded like the whole, by 1e-5 of the TRIANGLE's largest coordinate)
        std::vector<double> pieces;
        const size_t split_budget = want_guide && ctx->guide_split_ratio > 0 ? (size_t)s->n_prims / 2u + 4096u : 0u;
        size_t split_extra = 0;
        auto leaf_box = [&](const double* lo, const double* hi, const double* corners = nullptr) {
            GuideLeaf l;
            for (int q = 0; q < 3; ++q) { l.lo[q] = lo[q]; l.hi[q] = hi[q]; }
            l.offset = (uint32_t)std::min<size_t>(walk.size() * 8u, 0xFFFFFFFFu);
            guide::pad(l);
            double m = 0.0;
            for (int q = 0; q < 3; ++q) m = std::max(m, std::max(std::fabs(lo[q]), std::fabs(hi[q])));
            pieces.clear();
            size_t n_pieces = 1;
            if (corners && split_extra < split_budget && std::isfinite(m))
                n_pieces = guide::split_pieces(guide::split_poly_of(corners), (double)ctx->guide_split_ratio, PT_GUIDE_SPLIT_DEPTH, &pieces);
            if (n_pieces > 1) {
                const double eps = 1e-5 * m + 1e-30; // guide::pad's
                for (size_t i = 0; i < n_pieces; ++i) {
                    GuideLeaf piece = l;
                    for (int q = 0; q < 3; ++q) { // (never beyond the whole triangle's padded box)
                        piece.lo[q] = std::max(l.lo[q], pieces[6 * i + q] - eps);
                        piece.hi[q] = std::min(l.hi[q], pieces[6 * i + 3 + q] + eps);
                    }
                    guide_leaves.push_back(piece);
                }
                split_extra += n_pieces - 1;
            } else {
                guide_leaves.push_back(l);
            }
            ow_leaf_m_min = std::min(ow_leaf_m_min, m);
            ow_leaf_m_max = std::max(ow_leaf_m_max, m);
            return m; // the leaf's largest coordinate: what its padding is proportional to
        };
        // (parent_word: for primitive records of a stream built for the ordered walk, where the parent node's record starts)
        auto put_header = [&](uint32_t kind, uint32_t aux_word, uint32_t parent_word = 0) {
            if (want_guide && parent_word >= (1u << (32 - PT_WK_LINK_SHIFT))) boxes_nested = false; // no room for the link: no ordered walk
            walk.push_back((uint64_t)(kind | (want_guide ? parent_word << PT_WK_LINK_SHIFT : 0u)) | ((uint64_t)aux_word << 32));
        };
        auto put = [&](double v) {
            uint64_t bits;
            std::memcpy(&bits, &v, 8);
            walk.push_back(bits);
        };
        auto plane_ok = [](double k) { k = std::fabs(k); return k == 0.0 || (k >= 0x1p-800 && k <= 0x1p400); };
        auto same = [](double a, double b) { return std::memcmp(&a, &b, 8) == 0; };
        // iterative pre-order: (ref, stage, word index of the node's header to patch when its subtree ends)
        struct Frame { uint32_t ref; int stage; uint32_t item; };
        std::vector<Frame> st;
        st.push_back({s->root, 0, 0});
        while (!st.empty()) {
            Frame& fr = st.back();
            if (fr.ref & PT_REF_LEAF) {
                const uint32_t first = fr.ref & PT_REF_INDEX_MASK, count = (fr.ref & PT_REF_GROUP6) ? 6u : 1u;
                const uint32_t parent = st.size() >= 2 ? st[st.size() - 2].item : 0u; // the root is a node: every leaf has one
                bool as_box = false;
                if (count == 6u) {
                    // the six sides of an AABox as built by geometry.rs:391-446: all from one (min, max) pair
                    const DevPrim* r = &prims[first];
                    const double x0 = r[4].d[0], x1 = r[5].d[0], y0 = r[2].d[0], y1 = r[3].d[0], z0 = r[0].d[0], z1 = r[1].d[0];
                    as_box = r[0].type == PT_PRIM_RECT_XY && r[1].type == PT_PRIM_RECT_XY && r[2].type == PT_PRIM_RECT_XZ &&
                             r[3].type == PT_PRIM_RECT_XZ && r[4].type == PT_PRIM_RECT_YZ && r[5].type == PT_PRIM_RECT_YZ;
                    for (int k = 0; k < 2 && as_box; ++k) {
                        as_box = same(r[k].d[1], x0) && same(r[k].d[2], x1) && same(r[k].d[3], y0) && same(r[k].d[4], y1) &&
                                 same(r[2 + k].d[1], x0) && same(r[2 + k].d[2], x1) && same(r[2 + k].d[3], z0) && same(r[2 + k].d[4], z1) &&
                                 same(r[4 + k].d[1], y0) && same(r[4 + k].d[2], y1) && same(r[4 + k].d[3], z0) && same(r[4 + k].d[4], z1);
                    }
                    if (as_box) {
                        const double lo[3] = {x0, y0, z0}, hi[3] = {x1, y1, z1};
                        const double m = leaf_box(lo, hi);
                        ow_rect_m = std::min(ow_rect_m, m);
                        ow_rect_extent = std::max(ow_rect_extent, m);
                        put_header(PT_WK_BOX6, first, parent);
                        for (double v : {x0, y0, z0, x1, y1, z1}) {
                            put(v);
                            if (!plane_ok(v)) walk_flags = 0;
                        }
                    }
                }
                for (uint32_t k = 0; k < count && !as_box; ++k) {
                    const DevPrim& pr = prims[first + k];
                    {
                        double lo[3], hi[3];
                        if (pr.type == PT_PRIM_SPHERE) {
                            for (int q = 0; q < 3; ++q) { lo[q] = pr.d[q] - std::fabs(pr.d[3]); hi[q] = pr.d[q] + std::fabs(pr.d[3]); }
                        } else if (pr.type == PT_PRIM_TRIANGLE) {
                            const double* v = s->prims[first + k].d; // the three corners
                            for (int q = 0; q < 3; ++q) {
                                lo[q] = std::min(v[q], std::min(v[3 + q], v[6 + q]));
                                hi[q] = std::max(v[q], std::max(v[3 + q], v[6 + q]));
                            }
                        } else { // k on the normal axis, [a_lo, a_hi] x [b_lo, b_hi] on the other two in x<y<z order
                            const int na = pr.type == PT_PRIM_RECT_XY ? 2 : (pr.type == PT_PRIM_RECT_XZ ? 1 : 0);
                            const int aa = na == 0 ? 1 : 0, ba = na == 2 ? 1 : 2;
                            lo[na] = hi[na] = pr.d[0];
                            lo[aa] = pr.d[1]; hi[aa] = pr.d[2];
                            lo[ba] = pr.d[3]; hi[ba] = pr.d[4];
                        }
                        const double m = leaf_box(lo, hi, pr.type == PT_PRIM_TRIANGLE ? s->prims[first + k].d : nullptr);
                        if (pr.type == PT_PRIM_SPHERE) {
                            const double c = std::max(std::fabs(pr.d[0]), std::max(std::fabs(pr.d[1]), std::fabs(pr.d[2])));
                            ow_origin_limit = std::min(ow_origin_limit, 5e4 * std::fabs(pr.d[3]) - c);
                        } else if (pr.type == PT_PRIM_TRIANGLE) {
                            const double e1 = std::sqrt(pr.d[3] * pr.d[3] + pr.d[4] * pr.d[4] + pr.d[5] * pr.d[5]);
                            const double e2 = std::sqrt(pr.d[6] * pr.d[6] + pr.d[7] * pr.d[7] + pr.d[8] * pr.d[8]);
                            ow_tri_ratio = std::max(ow_tri_ratio, m > 0.0 ? e1 * e2 / m : INFINITY);
                            ow_tri_extent = std::max(ow_tri_extent, m);
                        } else {
                            ow_rect_m = std::min(ow_rect_m, m);
                            ow_rect_extent = std::max(ow_rect_extent, m);
                        }
                    }
                    if (pr.type == PT_PRIM_SPHERE) {
                        put_header(PT_WK_SPHERE, first + k, parent);
                        for (int q = 0; q < 4; ++q) put(pr.d[q]);
                    } else if (pr.type == PT_PRIM_TRIANGLE) {
                        put_header(PT_WK_TRIANGLE, first + k, parent);
                        for (int q = 0; q < 9; ++q) put(pr.d[q]);
                    } else {
                        put_header(PT_WK_RECT_XY + (pr.type - PT_PRIM_RECT_XY), first + k, parent);
                        for (int q = 0; q < 5; ++q) put(pr.d[q]);
                        if (!plane_ok(pr.d[0])) walk_flags = 0;
                    }
                }
                st.pop_back();
                continue;
            }
            const PtNode& nd = s->nodes[fr.ref];
            if (fr.stage == 0) {
                fr.item = (uint32_t)walk.size();
                put_header(PT_WK_NODE, 0);
                for (int q = 0; q < 3; ++q) put(nd.bmin[q]);
                for (int q = 0; q < 3; ++q) put(nd.bmax[q]);
                for (int q = 0; q < 3; ++q)
                    if (!(std::isfinite(nd.bmin[q]) && std::isfinite(nd.bmax[q]) && nd.bmin[q] <= nd.bmax[q])) { walk_flags = 0; boxes_nested = false; }
                for (uint32_t child : {nd.left, nd.right}) {
                    if (child & PT_REF_LEAF) continue;
                    const PtNode& c = s->nodes[child];
                    for (int q = 0; q < 3; ++q)
                        if (!(nd.bmin[q] <= c.bmin[q] && c.bmax[q] <= nd.bmax[q])) boxes_nested = false;
                }
                fr.stage = 1;
                const uint32_t child = nd.left;
                st.push_back({child, 0, 0});
            } else if (fr.stage == 1) {
                fr.stage = 2;
                const uint32_t child = nd.right;
                st.push_back({child, 0, 0});
            } else {
                walk[fr.item] |= (uint64_t)(walk.size() * 8u) << 32;  // byte offset of the first record after this subtree
                st.pop_back();
            }
        }
    }

    PT_HIP(ctx, hipSetDevice(ctx->device));
    PT_HIP(ctx, hipStreamSynchronize(ctx->stream));
    ctx->has_scene = false;
    PT_HIP(ctx, ensure_buf(ctx->d_shade, shade_prims.size() * sizeof(ShadePrim)));
    PT_HIP(ctx, hipMemcpy(ctx->d_shade.p, shade_prims.data(), shade_prims.size() * sizeof(ShadePrim), hipMemcpyHostToDevice));
    if (walk.size() * sizeof(uint64_t) >= 0xFFFFFF00ull) return ctx->fail(PT_ERR_BAD_ARG, "pt_scene_upload: scene too large (record stream over 4 GiB)");
    const size_t walk_words = walk.size();
    for (int k = 0; k < 16; ++k) walk.push_back(0); // the per-lane walks fetch 64 bytes (+16 for a triangle) per record whatever its kind
    {
        PT_HIP(ctx, ensure_buf(ctx->d_walk, walk.size() * sizeof(uint64_t)));
        PT_HIP(ctx, hipMemcpy(ctx->d_walk.p, walk.data(), walk.size() * sizeof(uint64_t), hipMemcpyHostToDevice));
    }
    std::vector<GuideNode> guide_nodes;
    if (boxes_nested) walk_flags |= PT_WALK_BOXES_NESTED;
    // The guide tree's allocation holds the nodes and, right behind them, a second copy of the (padded) record stream: the
    // ordered walk then reaches everything it fetches with a 32-bit offset from one base (pt_lean.h).  n leaves => n - 1 nodes.
    const size_t walk_padded_bytes = walk.size() * sizeof(uint64_t);
    const size_t guide_bytes_max = guide_leaves.size() * sizeof(GuideNode);
    const bool guide_possible = want_guide && boxes_nested && ow_leaf_m_max <= 1e18 && walk_padded_bytes < 0x80000000ull &&
                                guide_bytes_max + walk_padded_bytes < 0xFFFFFF00ull;
    bool has_guide = false;
    size_t n_guide_nodes = 0;
    if (guide_possible && ctx->guide_build_device && guide::usable(guide_leaves)) {
        n_guide_nodes = guide_leaves.size() - 1;
        PT_HIP(ctx, ensure_buf(ctx->d_guide, n_guide_nodes * sizeof(GuideNode) + walk_padded_bytes));
        const char* why = "";
        hipError_t e = guide_build_on_device(ctx->stream, guide_leaves, (GuideNode*)ctx->d_guide.p, &why);
        if (e != hipSuccess) {
            (void)hipStreamSynchronize(ctx->stream);
            return ctx->fail_hip(why, e);
        }
        has_guide = true;
    } else if (guide_possible && guide::build(guide_leaves, guide_nodes)) {
        n_guide_nodes = guide_nodes.size();
        PT_HIP(ctx, ensure_buf(ctx->d_guide, guide_nodes.size() * sizeof(GuideNode) + walk_padded_bytes));
        PT_HIP(ctx, hipMemcpy(ctx->d_guide.p, guide_nodes.data(), guide_nodes.size() * sizeof(GuideNode), hipMemcpyHostToDevice));
        has_guide = true;
    }
    if (has_guide)
        PT_HIP(ctx, hipMemcpy((unsigned char*)ctx->d_guide.p + n_guide_nodes * sizeof(GuideNode), walk.data(), walk_padded_bytes, hipMemcpyHostToDevice));
    PT_HIP(ctx, ensure_buf(ctx->d_nodes, nodes.size() * sizeof(DevNode)));
    PT_HIP(ctx, ensure_buf(ctx->d_prims, prims.size() * sizeof(DevPrim)));
    PT_HIP(ctx, ensure_buf(ctx->d_aux, aux.size() * sizeof(double)));
    PT_HIP(ctx, ensure_buf(ctx->d_mats, (size_t)s->n_materials * sizeof(PtMaterial)));
    PT_HIP(ctx, ensure_buf(ctx->d_texs, std::max<size_t>(1, s->n_textures) * sizeof(PtTexture)));
    PT_HIP(ctx, ensure_buf(ctx->d_perlin, std::max<size_t>(1, s->n_perlin_tables) * 256));
    PT_HIP(ctx, hipMemcpy(ctx->d_nodes.p, nodes.data(), nodes.size() * sizeof(DevNode), hipMemcpyHostToDevice));
    PT_HIP(ctx, hipMemcpy(ctx->d_prims.p, prims.data(), prims.size() * sizeof(DevPrim), hipMemcpyHostToDevice));
    PT_HIP(ctx, hipMemcpy(ctx->d_aux.p, aux.data(), aux.size() * sizeof(double), hipMemcpyHostToDevice));
    PT_HIP(ctx, hipMemcpy(ctx->d_mats.p, s->materials, (size_t)s->n_materials * sizeof(PtMaterial), hipMemcpyHostToDevice));
    if (s->n_textures)
        PT_HIP(ctx, hipMemcpy(ctx->d_texs.p, s->textures, (size_t)s->n_textures * sizeof(PtTexture), hipMemcpyHostToDevice));
    if (s->n_perlin_tables)
        PT_HIP(ctx, hipMemcpy(ctx->d_perlin.p, s->perlin_tables, (size_t)s->n_perlin_tables * 256, hipMemcpyHostToDevice));

    DevScene& sc = ctx->sc;
    sc.nodes = (const DevNode*)ctx->d_nodes.p;
    sc.prims = (const DevPrim*)ctx->d_prims.p;
    sc.aux = (const double*)ctx->d_aux.p;
    sc.materials = (const PtMaterial*)ctx->d_mats.p;
    sc.textures = (const PtTexture*)ctx->d_texs.p;
    sc.perlin = (const uint8_t*)ctx->d_perlin.p;
    sc.n_perlin = s->n_perlin_tables;
    sc.n_nodes = s->n_nodes;
    sc.n_prims = s->n_prims;
    sc.n_materials = s->n_materials;
    sc.n_textures = s->n_textures;
    sc.root = 0;  // the root is first in breadth-first order
    sc.stack_depth = tc.max_depth;
    sc.walk = (const unsigned char*)ctx->d_walk.p;
    sc.walk_bytes = (uint32_t)(walk_words * sizeof(uint64_t));
    // which kernel walks it: the wave-uniform walk while the stream fits the scalar cache, one lane per ray beyond
    // and, beyond that, in nearest-first order over the guide tree where one could be built
    sc.guide = has_guide ? (const unsigned char*)ctx->d_guide.p : nullptr;
    sc.guide_nodes = has_guide ? (uint32_t)n_guide_nodes : 0u;
    sc.deep_tree = ctx->deep_tree >= 0 ? (uint32_t)ctx->deep_tree : (s->n_prims >= PT_TAIL_MIN_PRIMS ? 1u : 0u);
    sc.trace_mode = !large ? PT_TRACE_LOCKSTEP : (has_guide ? PT_TRACE_ORDERED : PT_TRACE_STREAM);
    if (ctx->trace_mode_override >= 0) sc.trace_mode = (uint32_t)ctx->trace_mode_override;
    if (sc.trace_mode == PT_TRACE_ORDERED && !has_guide) sc.trace_mode = PT_TRACE_STREAM; // e.g. non-finite primitive bounds
    sc.walk_flags = walk_flags;
    if (std::isfinite(ow_rect_m)) ow_origin_limit = std::min(ow_origin_limit, 2e10 * ow_rect_m - ow_rect_extent);
    if (std::isfinite(ow_leaf_m_min)) ow_origin_limit = std::min(ow_origin_limit, 5e8 * ow_leaf_m_min);
    sc.ow_origin_limit = ow_origin_limit == ow_origin_limit ? ow_origin_limit : -1.0; // (negative or NaN: no ray qualifies)
    sc.ow_tri_extent = ow_tri_extent;
    sc.ow_od_limit = ow_tri_ratio > 0.0 ? 8e4 / ow_tri_ratio : __builtin_huge_val();
    sc.shade_prims = (const ShadePrim*)ctx->d_shade.p;
    sc.has_procedural = PT_PROC_NONE;
    for (uint32_t i = 0; i < s->n_textures; ++i) {
        if (s->textures[i].type == PT_TEX_CHECKER && sc.has_procedural == PT_PROC_NONE) sc.has_procedural = PT_PROC_CHECKER;
        if (s->textures[i].type == PT_TEX_PERLIN) sc.has_procedural = PT_PROC_MARBLE;
    }

    // LDS plan for the trace kernel: [staged nodes][staged primitives][per-lane stack]
    const size_t stack_bytes = (size_t)sc.stack_depth * PT_BLOCK * sizeof(uint32_t);
    const size_t scene_bytes = (size_t)s->n_nodes * sizeof(DevNode) + (size_t)s->n_prims * sizeof(DevPrim);
    const size_t lds_limit = 64 * 1024;
    LaunchGeom& g = ctx->geom;
    if (scene_bytes + stack_bytes <= 40 * 1024) {
        g.all_lds = true;
        sc.lds_nodes = s->n_nodes;
        sc.lds_prims = s->n_prims;
    } else {
        g.all_lds = false;
        size_t room = lds_limit > stack_bytes ? (lds_limit - stack_bytes) / 2 : 0;  // half the remainder: keep 2+ blocks/CU
        sc.lds_nodes = (uint32_t)std::min<size_t>(s->n_nodes, room / sizeof(DevNode));
        sc.lds_prims = 0;
    }
    g.trace_lds_bytes = (size_t)sc.lds_nodes * sizeof(DevNode) + (size_t)sc.lds_prims * sizeof(DevPrim) + stack_bytes;
    // Waves per CU of the persistent grid: 32 (8 per SIMD) when every kernel of the pass fits 8 waves per SIMD (small
    // scene, solid textures); otherwise 24, a whole number of rounds for kernels that fit 6, 3 or 2 (no ragged last round).
    // (the ordered walk: 32 as well, with four lanes (pt_render_accumulate) -- SphereField 1080p 128 spp 85.3 -> 83.6 ms, the 1M mesh
    // unchanged; profiles/r03_lanes_blocks_sweep.txt)
    uint32_t blocks_per_cu = (sc.trace_mode == PT_TRACE_ORDERED || (sc.trace_mode != PT_TRACE_STACK && !sc.has_procedural)) ? 8u : 6u;
    if (ctx->blocks_per_cu_override > 0) blocks_per_cu = (uint32_t)ctx->blocks_per_cu_override;
    g.grid = (uint32_t)ctx->cu_count * blocks_per_cu;
    ctx->has_scene = true;
    return PT_OK;
}

int pt_trace_batch(PtContext* ctx, const double* origins, const double* dirs, size_t n, double t_min, double t_max,
                   PtHitOut* out) {
    if (!ctx) return PT_ERR_BAD_ARG;
    if (!ctx->has_scene) return ctx->fail(PT_ERR_NO_SCENE, "pt_trace_batch: no scene uploaded");
    if (n == 0) return PT_OK;
    if (!origins || !dirs || !out) return ctx->fail(PT_ERR_BAD_ARG, "pt_trace_batch: null array");
    PT_HIP(ctx, hipSetDevice(ctx->device));
    const size_t chunk_max = (size_t)1 << 22;
    const size_t cap = std::min(n, chunk_max);
    DevBuf d_o, d_d, d_out;
    hipError_t e = ensure_buf(d_o, cap * 3 * sizeof(double));
    if (e == hipSuccess) e = ensure_buf(d_d, cap * 3 * sizeof(double));
    if (e == hipSuccess) e = ensure_buf(d_out, cap * sizeof(PtHitOut));
    int rc = PT_OK;
    if (e != hipSuccess) rc = ctx->fail_hip("hipMalloc(trace batch)", e);
    for (size_t done = 0; rc == PT_OK && done < n; done += chunk_max) {
        const size_t m = std::min(chunk_max, n - done);
        e = hipMemcpyAsync(d_o.p, origins + done * 3, m * 3 * sizeof(double), hipMemcpyHostToDevice, ctx->stream);
        if (e == hipSuccess) e = hipMemcpyAsync(d_d.p, dirs + done * 3, m * 3 * sizeof(double), hipMemcpyHostToDevice, ctx->stream);
        if (e == hipSuccess) {
            launch_trace_batch(ctx->stream, ctx->geom, ctx->sc, (const double*)d_o.p, (const double*)d_d.p, (uint32_t)m, t_min, t_max,
                               (PtHitOut*)d_out.p);
            e = hipGetLastError();
        }
        if (e == hipSuccess) e = hipMemcpyAsync(out + done, d_out.p, m * sizeof(PtHitOut), hipMemcpyDeviceToHost, ctx->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
        if (e != hipSuccess) rc = ctx->fail_hip("pt_trace_batch", e);
    }
    free_buf(d_o);
    free_buf(d_d);
    free_buf(d_out);
    return rc;
}

// pt_trace_batch without a single allocation: the rays and the hit records are staged in this device's copy of `scratch`
// (reached through ctx->scratch_ptr, never through the array's name).  Nothing in there has to start at zero: the copies
// fill [0, n) of the two input areas, the kernel reads and writes only [0, n), and the copy back takes only [0, n).
int pt_trace_batch_static(PtContext* ctx, const double* origins, const double* dirs, size_t n, double t_min, double t_max,
                          PtHitOut* out) {
    if (!ctx) return PT_ERR_BAD_ARG;
    if (!ctx->has_scene) return ctx->fail(PT_ERR_NO_SCENE, "pt_trace_batch_static: no scene uploaded");
    if (n > PT_TRACE_STATIC_MAX_RAYS)
        return ctx->fail(PT_ERR_BAD_ARG, "pt_trace_batch_static: n exceeds PT_TRACE_STATIC_MAX_RAYS (" + std::to_string(PT_TRACE_STATIC_MAX_RAYS) +
                                             " rays fit the library's fixed scratch array); split the batch or use pt_trace_batch");
    if (n == 0) return PT_OK;
    if (!origins || !dirs || !out) return ctx->fail(PT_ERR_BAD_ARG, "pt_trace_batch_static: null array");
    if (!ctx->scratch_ptr) return ctx->fail(PT_ERR_STATE, "pt_trace_batch_static: no scratch address for this device");
    std::lock_guard<std::mutex> in_use(g_scratch_in_use[(unsigned)ctx->device & 255u]); // calls of this PROCESS on one GPU serialize
    PT_HIP(ctx, hipSetDevice(ctx->device));
    static_assert(sizeof(PtHitOut) == 40, "scratch layout assumes 40-byte hit records");
    double* d_o = reinterpret_cast<double*>(ctx->scratch_ptr);
    double* d_d = reinterpret_cast<double*>(ctx->scratch_ptr + (size_t)PT_TRACE_STATIC_MAX_RAYS * 24u);
    PtHitOut* d_out = reinterpret_cast<PtHitOut*>(ctx->scratch_ptr + (size_t)PT_TRACE_STATIC_MAX_RAYS * 48u);
    hipError_t e = hipMemcpyAsync(d_o, origins, n * 3 * sizeof(double), hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(d_d, dirs, n * 3 * sizeof(double), hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) {
        launch_trace_batch(ctx->stream, ctx->geom, ctx->sc, d_o, d_d, (uint32_t)n, t_min, t_max, d_out);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpyAsync(out, d_out, n * sizeof(PtHitOut), hipMemcpyDeviceToHost, ctx->stream);
    const hipError_t drained = hipStreamSynchronize(ctx->stream); // whatever was enqueued has finished before the call returns
    if (e == hipSuccess) e = drained;
    if (e != hipSuccess) return ctx->fail_hip("pt_trace_batch_static", e);
    return PT_OK;
}

int pt_accum_reset(PtContext* ctx, uint32_t width, uint32_t height) {
    if (!ctx) return PT_ERR_BAD_ARG;
    if (width < 2 || height < 2) return ctx->fail(PT_ERR_BAD_ARG, "pt_accum_reset: width and height must be >= 2");  // (w-1), (h-1) divide
    if ((uint64_t)width * height > ((uint64_t)1 << 28)) return ctx->fail(PT_ERR_BAD_ARG, "pt_accum_reset: image too large");
    PT_HIP(ctx, hipSetDevice(ctx->device));
    const size_t bytes = (size_t)width * height * 3 * sizeof(float);
    PT_HIP(ctx, ensure_buf(ctx->d_accum_own, bytes));
    PT_HIP(ctx, ensure_buf(ctx->d_rgb, (size_t)width * height * 3));
    ctx->d_accum = (float*)ctx->d_accum_own.p;
    ctx->width = width;
    ctx->height = height;
    PT_HIP(ctx, hipMemsetAsync(ctx->d_accum, 0, bytes, ctx->stream));
    PT_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return PT_OK;
}

int pt_accum_bind(PtContext* ctx, void* device_ptr) {
    if (!ctx) return PT_ERR_BAD_ARG;
    if (ctx->width == 0) return ctx->fail(PT_ERR_STATE, "pt_accum_bind before pt_accum_reset");
    ctx->d_accum = device_ptr ? (float*)device_ptr : (float*)ctx->d_accum_own.p;
    return PT_OK;
}

void* pt_accum_device_ptr(PtContext* ctx) { return ctx ? (void*)ctx->d_accum : nullptr; }

int pt_render_accumulate(PtContext* ctx, const PtCamera* camera, uint32_t width, uint32_t height, const double background[3],
                         uint32_t sample_begin, uint32_t sample_count, uint32_t max_bounces, uint64_t seed, PtStats* stats) {
    if (!ctx) return PT_ERR_BAD_ARG;
    if (stats) std::memset(stats, 0, sizeof(*stats));
    if (!ctx->has_scene) return ctx->fail(PT_ERR_NO_SCENE, "pt_render_accumulate: no scene uploaded");
    if (!camera || !background) return ctx->fail(PT_ERR_BAD_ARG, "pt_render_accumulate: null argument");
    if (ctx->width == 0 || !ctx->d_accum) return ctx->fail(PT_ERR_STATE, "pt_render_accumulate before pt_accum_reset");
    if (width != ctx->width || height != ctx->height)
        return ctx->fail(PT_ERR_BAD_ARG, "pt_render_accumulate: size differs from pt_accum_reset");
    if (max_bounces > 4096) return ctx->fail(PT_ERR_BAD_ARG, "pt_render_accumulate: max_bounces > 4096");
    if ((uint64_t)sample_begin + sample_count > 0xFFFFFFFFull) return ctx->fail(PT_ERR_BAD_ARG, "sample range overflows 32 bits");
    if (sample_count == 0 || max_bounces == 0) return PT_OK;  // ray.rs:21-23: depth 0 contributes black
    PT_HIP(ctx, hipSetDevice(ctx->device));

    const uint32_t npix = width * height;
    uint64_t slots64 = std::max<uint64_t>(1, (uint64_t)ctx->paths_per_pass / npix);
    slots64 = std::min<uint64_t>(slots64, sample_count);
    slots64 = std::min<uint64_t>(slots64, 0xFFFFFFFFull / npix);
    uint32_t slots = (uint32_t)slots64;
    const uint32_t n_counts = max_bounces + 1;
    // Lanes share a pass: the pass's sample slots are dealt out to up to n_lanes independent sub-passes, each on its
    // own stream with its own buffers, started one pipeline step apart and then free-running, so that the VALU-bound
    // trace of one lane shares the machine with the memory-bound shade of another.
    int n_lanes = 1;
    LaunchGeom geom = ctx->geom;
    auto lane_share = [](uint32_t total, int lanes, int l) { return total / (uint32_t)lanes + ((uint32_t)l < total % (uint32_t)lanes ? 1u : 0u); };
    for (;;) {  // a pass as large as allowed; when the device cannot hold its streams, half as large
        // automatic: three overlapped sub-passes, each on a third of the persistent grid (measured on the Cornell box at 1080p,
        // ms/frame at 32 / 64 / 128 / 256 spp: 20.5 / 37.3 / 70.4 / 133.3; two lanes on the full grid each: 21.0 / 38.0 / 72.4 /
        // 138; four lanes on a quarter each: 20.8 / 37.5 / 70.4 / 133.7; four on the full grid: 22.2 / 38.9 / 73.7 / 136-141)
        uint32_t want = (uint32_t)ctx->n_lanes;
        if (want == 0) want = ctx->sc.trace_mode == PT_TRACE_ORDERED ? 4u : 3u; // (ordered walk: four on a quarter each, see pt_scene_upload)
        n_lanes = (int)std::min<uint32_t>(want, slots);
        geom = ctx->geom;
        if (n_lanes > 1 && !ctx->lane_full_grid) geom.grid = std::max<uint32_t>(1u, ctx->geom.grid / (uint32_t)n_lanes);
        int rc = PT_OK;
        for (int l = 0; l < n_lanes && rc == PT_OK; ++l)
            rc = ensure_lane(ctx, ctx->lanes[l], geom.grid, (size_t)lane_share(slots, n_lanes, l) * npix, max_bounces);
        if (rc == PT_OK) break;
        free_wavefront(ctx);
        (void)hipGetLastError();
        if (slots == 1) return rc;
        slots = (slots + 1) / 2;
    }

    PassParams base{};
    base.width = width;
    base.height = height;
    base.npix = npix;
    base.max_bounces = max_bounces;
    base.seed = seed;
    for (int k = 0; k < 3; ++k) base.bg[k] = (float)background[k];
    base.tiled = (width % 8 == 0 && height % 8 == 0) ? 1 : 0;

    // Event bookkeeping.  Every launch is bracketed by two events on its own stream; the stop event doubles as the
    // hand-off the other lane waits on.
    enum { K_RAYGEN = 0, K_TRACE = 1, K_SHADE = 2, K_FOLD = 3, K_TOTALS = 4, K_REPACK = 5, K_PATH = 6, K_TAIL = 7, K_DEAL = 8 };
    // one launch for the whole bounce loop (path.hip) where the wave-uniform walk applies and textures are solid
    const bool fused = ctx->fused && ctx->sc.trace_mode == PT_TRACE_LOCKSTEP && !ctx->sc.has_procedural && ctx->repack_every == 0;
    // ordered walk: the first few bounces as a trace and a shade launch each, the thinned-out rest in one per-wave launch.
    // Automatic: only for scenes with deep trees (a million triangles: a launch of few rays lasts as long as its slowest
    // ray's hundreds of dependent fetches, 29 -> 21 ms per 8-spp frame); SphereField's late bounces are cheap launches and
    // run better on the separate kernels' higher occupancy.
    uint32_t tail_from = max_bounces;
    if (ctx->sc.trace_mode == PT_TRACE_ORDERED && ctx->repack_every == 0) {
        if (ctx->tail_from >= 0) tail_from = (uint32_t)ctx->tail_from;
        else if (ctx->sc.n_prims >= PT_TAIL_MIN_PRIMS) tail_from = PT_TAIL_FROM_DEFAULT;
    }
    // lockstep walk, solid textures: the same idea with path.hip's fused kernel ("fused" = from bounce 0)
    const bool lockstep_tail_ok = ctx->sc.trace_mode == PT_TRACE_LOCKSTEP && ctx->sc.has_procedural == PT_PROC_NONE && ctx->repack_every == 0;
    if (lockstep_tail_ok) {
        if (fused) tail_from = 0;
        else if (ctx->tail_from >= 0) tail_from = (uint32_t)ctx->tail_from;
        else tail_from = PT_TAIL_FROM_LOCKSTEP;
    }
    struct Timed { int kind; size_t start, stop; };
    std::vector<Timed> timed;
    size_t ev_used = 0;
    auto new_event = [&](size_t& idx) -> hipError_t {
        if (ev_used == ctx->events.size()) {
            hipEvent_t ev;
            hipError_t e = hipEventCreateWithFlags(&ev, hipEventDefault);
            if (e != hipSuccess) return e;
            ctx->events.push_back(ev);
        }
        idx = ev_used++;
        return hipSuccess;
    };

    // One lane's pass as a list of steps; step s of lane 1 runs in the same time slot as step s + 1 of lane 0.
    struct Step { int kind; uint32_t bounce; BounceIO io; };
    struct LanePlan {
        PassParams p;
        uint32_t slots;
        std::vector<Step> steps;
        long last_stop = -1;  // event index of the lane's previous step
    };
    auto plan_lane = [&](LanePlan& pl, const Wavefront& wf) {
        pl.steps.clear();
        pl.steps.push_back({K_RAYGEN, 0, BounceIO{0, 1, nullptr, nullptr}});
        int buf = 0;
        const uint32_t* counts_in = wf.counts;
        for (uint32_t b = 0; b < max_bounces; ++b) {
            if (b == tail_from) {
                if (!lockstep_tail_ok && ctx->tail_deal && b > 0) {
                    // the survivors are dealt out over the waves one at a time first (shade.hip repack_kernel<true>)
                    BounceIO rp{buf, buf ^ 1, counts_in, wf.counts_repacked + count_row(wf, b)};
                    pl.steps.push_back({K_DEAL, b, rp});
                    buf ^= 1;
                    counts_in = rp.counts_out;
                }
                pl.steps.push_back({lockstep_tail_ok ? K_PATH : K_TAIL, b, BounceIO{buf, buf ^ 1, counts_in, nullptr}});
                break;
            }
            BounceIO io{buf, buf ^ 1, counts_in, wf.counts + count_row(wf, b + 1)};
            pl.steps.push_back({K_TRACE, b, io});
            pl.steps.push_back({K_SHADE, b, io});
            buf ^= 1;
            counts_in = io.counts_out;
            const uint32_t every = (uint32_t)ctx->repack_every;
            if (every && b + 1 < max_bounces && b + 1 >= (uint32_t)ctx->repack_first && (b + 1 - ctx->repack_first) % every == 0) {
                BounceIO rp{buf, buf ^ 1, counts_in, wf.counts_repacked + count_row(wf, b + 1)};
                pl.steps.push_back({K_REPACK, b, rp});
                buf ^= 1;
                counts_in = rp.counts_out;
            }
        }
        pl.steps.push_back({K_FOLD, 0, BounceIO{0, 1, nullptr, nullptr}});
        pl.steps.push_back({K_TOTALS, 0, BounceIO{0, 1, nullptr, nullptr}});
    };

    size_t ev_begin = 0, ev_end = 0;
    uint32_t passes = 0, launches_trace = 0, launches_shade = 0, launches_path = 0;
    std::vector<unsigned long long> totals((size_t)n_counts * PT_MAX_LANES, 0);
    const bool profile = ctx->profile != 0;
    // Everything that enqueues work.  A failure in here returns with kernels possibly still running on the other lanes'
    // streams: the caller below drains every lane before the error leaves the library.
    auto enqueue = [&]() -> int {
    for (int l = 0; l < n_lanes; ++l)
        PT_HIP(ctx, hipMemsetAsync(ctx->lanes[l].wf.totals, 0, (size_t)n_counts * sizeof(unsigned long long), ctx->lanes[l].stream));
    PT_HIP(ctx, new_event(ev_begin));
    PT_HIP(ctx, hipEventRecord(ctx->events[ev_begin], ctx->stream));
    for (int l = 1; l < n_lanes; ++l) PT_HIP(ctx, hipStreamWaitEvent(ctx->lanes[l].stream, ctx->events[ev_begin], 0));

    long fold_guard = -1;  // stop event of the last fold: folds add to the same buffer and must not overlap
    for (uint32_t done = 0; done < sample_count; done += slots) {
        const uint32_t cur = std::min(slots, sample_count - done);
        LanePlan plan[PT_MAX_LANES];
        const int active = (int)std::min<uint32_t>((uint32_t)n_lanes, cur);
        uint32_t slot_base = 0;
        for (int l = 0; l < active; ++l) {
            plan[l].p = base;
            plan[l].slots = lane_share(cur, active, l);
            plan[l].p.n_paths = plan[l].slots * npix;
            plan[l].p.sample_base = sample_begin + done + slot_base;
            slot_base += plan[l].slots;
            plan_lane(plan[l], ctx->lanes[l].wf);
        }
        const size_t n_steps = plan[0].steps.size();
        for (size_t slot = 0; slot < n_steps + (size_t)(active - 1); ++slot) {
            long stop_of[PT_MAX_LANES];
            for (int l = 0; l < active; ++l) stop_of[l] = plan[l].last_stop;
            for (int l = 0; l < active; ++l) {
                const long si = (long)slot - l;  // lane l starts l steps behind lane 0
                if (si < 0 || si >= (long)plan[l].steps.size()) continue;
                const Step& st = plan[l].steps[(size_t)si];
                PtContext::Lane& ln = ctx->lanes[l];
                if (active > 1 && (ctx->lane_sync || si == 0)) {
                    // stagger: a lane's first step (always) and, with lane_sync, every step waits for the previous
                    // lane's previous step
                    const int prev = (l + active - 1) % active;
                    if (plan[prev].last_stop >= 0)
                        PT_HIP(ctx, hipStreamWaitEvent(ln.stream, ctx->events[(size_t)plan[prev].last_stop], 0));
                }
                if (st.kind == K_FOLD && fold_guard >= 0) PT_HIP(ctx, hipStreamWaitEvent(ln.stream, ctx->events[(size_t)fold_guard], 0));
                // Without profiling only the hand-offs are recorded: a raygen's end (the next lane's start waits for it) and
                // a fold's (folds add to one buffer and must not overlap); with lane_sync every step is a hand-off.
                const bool want_stop = profile || ctx->lane_sync || st.kind == K_RAYGEN || st.kind == K_FOLD;
                Timed t{st.kind, 0, 0};
                if (profile) {
                    PT_HIP(ctx, new_event(t.start));
                    PT_HIP(ctx, hipEventRecord(ctx->events[t.start], ln.stream));
                }
                if (want_stop) PT_HIP(ctx, new_event(t.stop));
                switch (st.kind) {
                    case K_RAYGEN: launch_raygen(ln.stream, geom, *camera, plan[l].p, ln.wf); break;
                    case K_PATH: launch_path(ln.stream, geom, ctx->sc, plan[l].p, ln.wf, st.io, st.bounce); ++launches_path; break;
                    case K_TAIL: launch_tail(ln.stream, geom, ctx->sc, plan[l].p, ln.wf, st.io, st.bounce); ++launches_path; break;
                    case K_TRACE: launch_trace(ln.stream, geom, ctx->sc, ln.wf, st.io); ++launches_trace; break;
                    case K_SHADE: launch_shade(ln.stream, geom, ctx->sc, plan[l].p, ln.wf, st.io, st.bounce); ++launches_shade; break;
                    case K_REPACK: launch_repack(ln.stream, geom, ln.wf, st.io); break;
                    case K_DEAL: launch_repack(ln.stream, geom, ln.wf, st.io, std::max<uint32_t>(1u, ln.wf.n_waves / (uint32_t)ctx->tail_waves_div)); break;
                    case K_FOLD: launch_fold(ln.stream, plan[l].p, ln.wf, plan[l].slots, ctx->d_accum); break;
                    default: launch_count_totals(ln.stream, ln.wf, max_bounces); break;
                }
                if (want_stop) {
                    PT_HIP(ctx, hipEventRecord(ctx->events[t.stop], ln.stream));
                    if (st.kind == K_FOLD) fold_guard = (long)t.stop;
                    stop_of[l] = (long)t.stop;
                }
                if (profile) timed.push_back(t);
            }
            for (int l = 0; l < active; ++l) plan[l].last_stop = stop_of[l];
        }
        PT_HIP(ctx, hipGetLastError());
        ++passes;
    }
    // join: lane 0's stream waits for lane 1, then the end marker
    for (int l = 1; l < n_lanes; ++l) {
        size_t ej = 0;
        PT_HIP(ctx, new_event(ej));
        PT_HIP(ctx, hipEventRecord(ctx->events[ej], ctx->lanes[l].stream));
        PT_HIP(ctx, hipStreamWaitEvent(ctx->stream, ctx->events[ej], 0));
    }
    PT_HIP(ctx, new_event(ev_end));
    PT_HIP(ctx, hipEventRecord(ctx->events[ev_end], ctx->stream));
    for (int l = 0; l < n_lanes; ++l)
        PT_HIP(ctx, hipMemcpyAsync(totals.data() + (size_t)l * n_counts, ctx->lanes[l].wf.totals,
                                   (size_t)n_counts * sizeof(unsigned long long), hipMemcpyDeviceToHost, ctx->stream));
    return PT_OK;
    };
    const int enqueue_rc = enqueue();
    // drain every lane, whatever happened above (ptgpu.h: calls return after the context's streams have drained)
    hipError_t drain_err = hipSuccess;
    for (int l = 0; l < PT_MAX_LANES; ++l) {
        if (!ctx->lanes[l].stream) continue;
        hipError_t e = hipStreamSynchronize(ctx->lanes[l].stream);
        if (e != hipSuccess && drain_err == hipSuccess) drain_err = e;
    }
    if (enqueue_rc != PT_OK) return enqueue_rc; // (its message stands)
    if (drain_err != hipSuccess) return ctx->fail_hip("pt_render_accumulate: hipStreamSynchronize", drain_err);

    if (stats) {
        stats->paths = (uint64_t)npix * sample_count;
        stats->passes = passes;
        stats->launches_trace = launches_trace;
        stats->launches_shade = launches_shade;
        stats->launches_path = launches_path;
        stats->tail_from = tail_from < max_bounces ? tail_from : max_bounces;
        for (uint32_t b = 0; b < max_bounces; ++b) {
            unsigned long long v = 0;
            for (int l = 0; l < n_lanes; ++l) v += totals[(size_t)l * n_counts + b];
            stats->rays += v;
            stats->rays_per_bounce[std::min<uint32_t>(b, 63)] += v;
        }
        // Timing is a by-product: an event that cannot be read zeroes the times, it does not fail a finished render.
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, ctx->events[ev_begin], ctx->events[ev_end]) == hipSuccess) stats->ms_total = ms;
        else (void)hipGetLastError();
        if (profile) {
            bool ok = true;
            std::vector<std::pair<float, float>> spans[3], all; // [trace, shade, path] launch intervals since ev_begin
            all.reserve(timed.size());
            for (const Timed& t : timed) {
                float t0 = 0.f, t1 = 0.f;
                if (hipEventElapsedTime(&t0, ctx->events[ev_begin], ctx->events[t.start]) != hipSuccess ||
                    hipEventElapsedTime(&t1, ctx->events[ev_begin], ctx->events[t.stop]) != hipSuccess) {
                    ok = false;
                    break;
                }
                ms = t1 - t0;
                switch (t.kind) {
                    case K_RAYGEN: stats->ms_raygen += ms; break;
                    case K_TRACE: stats->ms_trace += ms; spans[0].emplace_back(t0, t1); break;
                    case K_SHADE: stats->ms_shade += ms; spans[1].emplace_back(t0, t1); break;
                    case K_FOLD: stats->ms_fold += ms; break;
                    case K_REPACK: case K_DEAL: stats->ms_repack += ms; break;
                    case K_PATH: stats->ms_path += ms; spans[2].emplace_back(t0, t1); break;
                    case K_TAIL: stats->ms_path += ms; spans[2].emplace_back(t0, t1); break;
                    default: break;
                }
                if (t.kind != K_TOTALS) all.emplace_back(t0, t1);
            }
            auto union_ms = [](std::vector<std::pair<float, float>>& v) {
                std::sort(v.begin(), v.end());
                double sum = 0.0;
                float lo = 0.f, hi = -1.f;
                for (const auto& iv : v) {
                    if (hi < lo || iv.first > hi) {
                        if (hi >= lo) sum += (double)hi - lo;
                        lo = iv.first;
                        hi = iv.second;
                    } else {
                        hi = std::max(hi, iv.second);
                    }
                }
                if (hi >= lo) sum += (double)hi - lo;
                return sum;
            };
            if (ok) {
                stats->ms_busy_trace = union_ms(spans[0]);
                stats->ms_busy_shade = union_ms(spans[1]);
                stats->ms_busy_path = union_ms(spans[2]);
                stats->ms_busy_all = union_ms(all);
            } else {
                (void)hipGetLastError();
                stats->ms_raygen = stats->ms_trace = stats->ms_shade = stats->ms_fold = stats->ms_repack = stats->ms_path = 0.0;
            }
        }
    }
    return PT_OK;
}

int pt_resolve(PtContext* ctx, uint32_t total_spp, uint8_t* rgb_out) {
    if (!ctx) return PT_ERR_BAD_ARG;
    if (ctx->width == 0 || !ctx->d_accum) return ctx->fail(PT_ERR_STATE, "pt_resolve before pt_accum_reset");
    if (!rgb_out || total_spp == 0) return ctx->fail(PT_ERR_BAD_ARG, "pt_resolve: null output or zero spp");
    PT_HIP(ctx, hipSetDevice(ctx->device));
    const uint32_t n = ctx->width * ctx->height * 3;
    launch_resolve(ctx->stream, ctx->d_accum, n, total_spp, (uint8_t*)ctx->d_rgb.p);
    PT_HIP(ctx, hipGetLastError());
    PT_HIP(ctx, hipMemcpyAsync(rgb_out, ctx->d_rgb.p, n, hipMemcpyDeviceToHost, ctx->stream));
    PT_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return PT_OK;
}

int pt_accum_download(PtContext* ctx, float* rgb_sum) {
    if (!ctx) return PT_ERR_BAD_ARG;
    if (ctx->width == 0 || !ctx->d_accum) return ctx->fail(PT_ERR_STATE, "pt_accum_download before pt_accum_reset");
    if (!rgb_sum) return ctx->fail(PT_ERR_BAD_ARG, "pt_accum_download: null output");
    PT_HIP(ctx, hipSetDevice(ctx->device));
    PT_HIP(ctx, hipMemcpyAsync(rgb_sum, ctx->d_accum, (size_t)ctx->width * ctx->height * 3 * sizeof(float), hipMemcpyDeviceToHost,
                               ctx->stream));
    PT_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return PT_OK;
}

int pt_accum_upload(PtContext* ctx, const float* rgb_sum) {
    if (!ctx) return PT_ERR_BAD_ARG;
    if (ctx->width == 0 || !ctx->d_accum) return ctx->fail(PT_ERR_STATE, "pt_accum_upload before pt_accum_reset");
    if (!rgb_sum) return ctx->fail(PT_ERR_BAD_ARG, "pt_accum_upload: null input");
    PT_HIP(ctx, hipSetDevice(ctx->device));
    PT_HIP(ctx, hipMemcpyAsync(ctx->d_accum, rgb_sum, (size_t)ctx->width * ctx->height * 3 * sizeof(float), hipMemcpyHostToDevice,
                               ctx->stream));
    PT_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return PT_OK;
}

}  // extern "C"

// ------------------------------------------------------------------ the multi-GPU exchange
namespace {
// librccl.so.1, resolved on first use: a host that never shards carries no RCCL dependency (the static archive links without it).
struct Rccl {
    void* lib = nullptr;
    decltype(&ncclCommInitAll) CommInitAll = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclReduce) Reduce = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    std::string error; // why loading failed
    bool load() {
        if (lib) return true;
        if (!error.empty()) return false;
        for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
            lib = dlopen(name, RTLD_NOW | RTLD_LOCAL);
            if (lib) break;
        }
        if (!lib) {
            const char* why = dlerror();
            error = std::string("cannot load librccl.so.1: ") + (why ? why : "unknown error");
            return false;
        }
        auto sym = [&](const char* n) {
            void* f = dlsym(lib, n);
            if (!f && error.empty()) error = std::string("librccl lacks ") + n;
            return f;
        };
        CommInitAll = reinterpret_cast<decltype(CommInitAll)>(sym("ncclCommInitAll"));
        CommDestroy = reinterpret_cast<decltype(CommDestroy)>(sym("ncclCommDestroy"));
        Reduce = reinterpret_cast<decltype(Reduce)>(sym("ncclReduce"));
        GroupStart = reinterpret_cast<decltype(GroupStart)>(sym("ncclGroupStart"));
        GroupEnd = reinterpret_cast<decltype(GroupEnd)>(sym("ncclGroupEnd"));
        GetErrorString = reinterpret_cast<decltype(GetErrorString)>(sym("ncclGetErrorString"));
        if (!error.empty()) {
            dlclose(lib);
            lib = nullptr;
            return false;
        }
        return true;
    }
};
std::mutex g_rccl_mutex;                                   // one reduce at a time per process
Rccl g_rccl;
// A clique = the communicators ncclCommInitAll made for one list of contexts (rank order), kept until one of those contexts
// is destroyed (release_cliques_of): a host that renders frame after frame on the same contexts pays for the set-up once.
struct Clique {
    std::vector<PtContext*> members;
    std::vector<ncclComm_t> comms;
};
std::vector<Clique> g_cliques;

void release_cliques_of(PtContext* ctx) {
    std::lock_guard<std::mutex> lock(g_rccl_mutex);
    for (size_t i = 0; i < g_cliques.size();) {
        Clique& c = g_cliques[i];
        if (std::find(c.members.begin(), c.members.end(), ctx) == c.members.end()) {
            ++i;
            continue;
        }
        // every member's last reduce has drained: pt_accum_reduce returns only after synchronising all their streams
        for (size_t k = 0; k < c.comms.size(); ++k) {
            if (!c.comms[k]) continue;
            (void)hipSetDevice(c.members[k]->device);
            (void)g_rccl.CommDestroy(c.comms[k]);
        }
        g_cliques.erase(g_cliques.begin() + (long)i);
    }
    (void)hipSetDevice(ctx->device);
}
}  // namespace

extern "C" int pt_accum_reduce(PtContext** ctxs, int n) {
    if (!ctxs || n < 1) {
        g_last_error = "pt_accum_reduce: no contexts";
        return PT_ERR_BAD_ARG;
    }
    PtContext* root = ctxs[0];
    if (!root) {
        g_last_error = "pt_accum_reduce: null context";
        return PT_ERR_BAD_ARG;
    }
    auto fail_all = [&](int code, const std::string& msg) { // the message is readable from every context of the call
        for (int i = 0; i < n; ++i)
            if (ctxs[i]) ctxs[i]->err = msg;
        g_last_error = msg;
        return code;
    };
    std::vector<int> devices((size_t)n);
    for (int i = 0; i < n; ++i) {
        if (!ctxs[i]) return fail_all(PT_ERR_BAD_ARG, "pt_accum_reduce: null context");
        if (ctxs[i]->width == 0 || !ctxs[i]->d_accum) return fail_all(PT_ERR_STATE, "pt_accum_reduce before pt_accum_reset");
        if (ctxs[i]->width != root->width || ctxs[i]->height != root->height)
            return fail_all(PT_ERR_BAD_ARG, "pt_accum_reduce: accumulation buffers differ in size");
        devices[(size_t)i] = ctxs[i]->device;
        for (int k = 0; k < i; ++k)
            if (ctxs[k] == ctxs[i] || ctxs[k]->device == ctxs[i]->device)
                return fail_all(PT_ERR_BAD_ARG, "pt_accum_reduce: one context per GPU (a context or a device is listed twice)");
    }
    std::lock_guard<std::mutex> lock(g_rccl_mutex);
    Rccl& r = g_rccl;
    if (!r.load()) return fail_all(PT_ERR_RCCL, "pt_accum_reduce: " + r.error);
    auto rccl_fail = [&](const char* what, ncclResult_t e) { return fail_all(PT_ERR_RCCL, std::string(what) + ": " + r.GetErrorString(e)); };
    auto hip_fail = [&](const char* what, hipError_t e) { return fail_all(PT_ERR_HIP, std::string(what) + ": " + hipGetErrorString(e)); };

    const std::vector<PtContext*> members(ctxs, ctxs + n);
    const Clique* clique = nullptr;
    for (const Clique& c : g_cliques)
        if (c.members == members) clique = &c;
    if (!clique) {
        std::vector<ncclComm_t> made((size_t)n, nullptr);
        ncclResult_t e = r.CommInitAll(made.data(), n, devices.data());
        if (e != ncclSuccess) return rccl_fail("ncclCommInitAll", e);
        g_cliques.push_back(Clique{members, std::move(made)});
        clique = &g_cliques.back();
    }
    const std::vector<ncclComm_t>& comms = clique->comms;
    const size_t count = (size_t)root->width * root->height * 3;
    // every context's earlier work (its render) runs on its own stream: the reduce is ordered behind it there
    ncclResult_t e = r.GroupStart();
    if (e != ncclSuccess) return rccl_fail("ncclGroupStart", e);
    ncclResult_t first_bad = ncclSuccess;
    for (int i = 0; i < n; ++i) {
        hipError_t he = hipSetDevice(ctxs[i]->device);
        if (he != hipSuccess) {
            (void)r.GroupEnd();
            return hip_fail("hipSetDevice", he);
        }
        e = r.Reduce(ctxs[i]->d_accum, ctxs[i]->d_accum, count, ncclFloat32, ncclSum, 0, comms[(size_t)i], ctxs[i]->stream);
        if (e != ncclSuccess && first_bad == ncclSuccess) first_bad = e;
    }
    e = r.GroupEnd();
    if (first_bad != ncclSuccess) return rccl_fail("ncclReduce", first_bad);
    if (e != ncclSuccess) return rccl_fail("ncclGroupEnd", e);
    for (int i = 0; i < n; ++i) {
        hipError_t he = hipSetDevice(ctxs[i]->device);
        if (he == hipSuccess) he = hipStreamSynchronize(ctxs[i]->stream);
        if (he != hipSuccess) return hip_fail("pt_accum_reduce: hipStreamSynchronize", he);
    }
    return PT_OK;
}

extern "C" {

static int selftest_arith(PtContext* ctx, const double* a, const double* b, size_t n, double* quot, double* root, int via_reciprocal) {
    if (!ctx) return PT_ERR_BAD_ARG;
    if (!a || !b || !quot || !root || n == 0 || n > ((size_t)1 << 26)) return ctx->fail(PT_ERR_BAD_ARG, "pt_selftest_arith: bad argument");
    PT_HIP(ctx, hipSetDevice(ctx->device));
    DevBuf da, db, dq, dr;
    hipError_t e = ensure_buf(da, n * 8);
    if (e == hipSuccess) e = ensure_buf(db, n * 8);
    if (e == hipSuccess) e = ensure_buf(dq, n * 8);
    if (e == hipSuccess) e = ensure_buf(dr, n * 8);
    if (e == hipSuccess) e = hipMemcpyAsync(da.p, a, n * 8, hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(db.p, b, n * 8, hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) {
        launch_selftest_arith(ctx->stream, (const double*)da.p, (const double*)db.p, (uint32_t)n, (double*)dq.p, (double*)dr.p,
                              via_reciprocal);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpyAsync(quot, dq.p, n * 8, hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(root, dr.p, n * 8, hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    free_buf(da);
    free_buf(db);
    free_buf(dq);
    free_buf(dr);
    if (e != hipSuccess) return ctx->fail_hip("pt_selftest_arith", e);
    return PT_OK;
}
int pt_selftest_arith(PtContext* ctx, const double* a, const double* b, size_t n, double* quot, double* root) {
    return selftest_arith(ctx, a, b, n, quot, root, 0);
}
int pt_selftest_division(PtContext* ctx, const double* a, const double* b, size_t n, double* quot) {
    if (!ctx) return PT_ERR_BAD_ARG;
    if (!quot || n == 0 || n > ((size_t)1 << 26)) return ctx->fail(PT_ERR_BAD_ARG, "pt_selftest_division: bad argument");
    std::vector<double> root(n);
    return selftest_arith(ctx, a, b, n, quot, root.data(), 1);
}

int pt_selftest_rng(PtContext* ctx, uint64_t seed, uint32_t pixel, uint32_t sample, uint32_t stream, uint32_t n, double* out) {
    if (!ctx) return PT_ERR_BAD_ARG;
    if (!out || n == 0 || n > 65536) return ctx->fail(PT_ERR_BAD_ARG, "pt_selftest_rng: bad argument");
    PT_HIP(ctx, hipSetDevice(ctx->device));
    DevBuf d;
    hipError_t e = ensure_buf(d, (size_t)n * 8);
    if (e == hipSuccess) {
        launch_selftest_rng(ctx->stream, seed, pixel, sample, stream, n, (double*)d.p);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpyAsync(out, d.p, (size_t)n * 8, hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    free_buf(d);
    if (e != hipSuccess) return ctx->fail_hip("pt_selftest_rng", e);
    return PT_OK;
}

int pt_selftest_texture(PtContext* ctx, uint32_t tex, const double* points, size_t n, double* rgb_out) {
    if (!ctx) return PT_ERR_BAD_ARG;
    if (!ctx->has_scene) return ctx->fail(PT_ERR_NO_SCENE, "pt_selftest_texture: no scene uploaded");
    if (!points || !rgb_out || n == 0 || n > ((size_t)1 << 24) || tex >= ctx->sc.n_textures)
        return ctx->fail(PT_ERR_BAD_ARG, "pt_selftest_texture: bad argument");
    PT_HIP(ctx, hipSetDevice(ctx->device));
    DevBuf dp, dc;
    hipError_t e = ensure_buf(dp, n * 24);
    if (e == hipSuccess) e = ensure_buf(dc, n * 24);
    if (e == hipSuccess) e = hipMemcpyAsync(dp.p, points, n * 24, hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) {
        launch_selftest_texture(ctx->stream, ctx->sc, tex, (const double*)dp.p, (uint32_t)n, (double*)dc.p);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpyAsync(rgb_out, dc.p, n * 24, hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    free_buf(dp);
    free_buf(dc);
    if (e != hipSuccess) return ctx->fail_hip("pt_selftest_texture", e);
    return PT_OK;
}

// Checks of the guide-tree builders (guide_tree.h on the host, guide_build.hip on the device): the tree over n boxes (lo.xyz,
// hi.xyz each; box i stands for the primitive record at stream offset 8 * i) is verified on the host: every box is a leaf
// exactly once, every child box stored in a node contains, after the outward rounding to f32, all the boxes below it.
static int verify_guide_tree(const std::vector<GuideNode>& nodes, const double* boxes, size_t n, uint32_t* n_nodes, uint32_t* max_depth) {
    if (nodes.size() != n - 1) return PT_ERR_STATE;
    // depth-first: each entry carries the box its parent stored for it
    struct Item { uint32_t ref; float box[6]; uint32_t depth; }; // box: lo.x, hi.x, lo.y, hi.y, lo.z, hi.z (GuideNode's layout)
    std::vector<Item> todo;
    std::vector<uint8_t> seen_leaf(n, 0), seen_node(nodes.size(), 0);
    const float inf = std::numeric_limits<float>::infinity();
    todo.push_back({0u, {-inf, inf, -inf, inf, -inf, inf}, 1u});
    uint32_t deepest = 0;
    while (!todo.empty()) {
        const Item it = todo.back();
        todo.pop_back();
        deepest = std::max(deepest, it.depth);
        if (it.ref & PT_GUIDE_LEAF) {
            const size_t i = (it.ref & ~PT_GUIDE_LEAF) / 8u;
            if (i >= n || seen_leaf[i]) return PT_ERR_STATE;
            seen_leaf[i] = 1;
            for (int k = 0; k < 3; ++k)  // the box lies inside what the parent stored for it
                if (!((double)it.box[2 * k] <= boxes[6 * i + k] && (double)it.box[2 * k + 1] >= boxes[6 * i + 3 + k])) return PT_ERR_STATE;
            continue;
        }
        if (it.ref >= nodes.size() || seen_node[it.ref]) return PT_ERR_STATE;
        seen_node[it.ref] = 1;
        const GuideNode& nd = nodes[it.ref];
        for (int c = 0; c < 2; ++c) {
            Item ch{nd.child[c], {}, it.depth + 1u};
            for (int k = 0; k < 6; ++k) ch.box[k] = nd.box[c][k];
            for (int k = 0; k < 3; ++k)  // nested: a child's box never sticks out of its parent's
                if (ch.box[2 * k] < it.box[2 * k] || ch.box[2 * k + 1] > it.box[2 * k + 1]) return PT_ERR_STATE;
            todo.push_back(ch);
        }
    }
    for (size_t i = 0; i < n; ++i)
        if (!seen_leaf[i]) return PT_ERR_STATE;
    if (n_nodes) *n_nodes = (uint32_t)nodes.size();
    if (max_depth) *max_depth = deepest;
    return PT_OK;
}
static bool selftest_leaves(const double* boxes, size_t n, std::vector<GuideLeaf>& leaves) {
    if (!boxes || n < 2 || n > ((size_t)1 << 26)) return false;
    leaves.resize(n);
    for (size_t i = 0; i < n; ++i) {
        for (int k = 0; k < 3; ++k) { leaves[i].lo[k] = boxes[6 * i + k]; leaves[i].hi[k] = boxes[6 * i + 3 + k]; }
        leaves[i].offset = (uint32_t)(8u * i);
        guide::pad(leaves[i]);
    }
    return true;
}
int pt_selftest_guide_tree(const double* boxes, size_t n, uint32_t* n_nodes, uint32_t* max_depth) {
    std::vector<GuideLeaf> leaves;
    if (!selftest_leaves(boxes, n, leaves)) return PT_ERR_BAD_ARG;
    std::vector<GuideNode> nodes;
    if (!guide::build(leaves, nodes)) return PT_ERR_BAD_ARG;
    return verify_guide_tree(nodes, boxes, n, n_nodes, max_depth);
}
// The pieces a triangle is referenced by in the guide tree (guide_tree.h split_pieces): their unpadded boxes, 6 doubles each.
int pt_selftest_split_triangle(const double corners[9], int ratio, double* boxes, size_t capacity, size_t* n_pieces) {
    if (!corners || !n_pieces || ratio < 1) return PT_ERR_BAD_ARG;
    for (int k = 0; k < 9; ++k)
        if (!std::isfinite(corners[k])) return PT_ERR_BAD_ARG;
    std::vector<double> out;
    *n_pieces = guide::split_pieces(guide::split_poly_of(corners), (double)ratio, PT_GUIDE_SPLIT_DEPTH, &out);
    if (boxes) std::memcpy(boxes, out.data(), std::min(capacity, *n_pieces) * 6 * sizeof(double));
    return PT_OK;
}
int pt_selftest_guide_tree_device(PtContext* ctx, const double* boxes, size_t n, uint32_t* n_nodes, uint32_t* max_depth) {
    if (!ctx) return PT_ERR_BAD_ARG;
    std::vector<GuideLeaf> leaves;
    if (!selftest_leaves(boxes, n, leaves) || !guide::usable(leaves)) return ctx->fail(PT_ERR_BAD_ARG, "pt_selftest_guide_tree_device: unusable boxes");
    PT_HIP(ctx, hipSetDevice(ctx->device));
    DevBuf d;
    hipError_t e = ensure_buf(d, (n - 1) * sizeof(GuideNode));
    const char* why = "hipMalloc(guide nodes)";
    std::vector<GuideNode> nodes(n - 1);
    if (e == hipSuccess) e = guide_build_on_device(ctx->stream, leaves, (GuideNode*)d.p, &why);
    if (e == hipSuccess) {
        why = "hipMemcpy(guide nodes)";
        e = hipMemcpy(nodes.data(), d.p, (n - 1) * sizeof(GuideNode), hipMemcpyDeviceToHost);
    }
    free_buf(d);
    if (e != hipSuccess) {
        (void)hipStreamSynchronize(ctx->stream);
        return ctx->fail_hip(why, e);
    }
    const int rc = verify_guide_tree(nodes, boxes, n, n_nodes, max_depth);
    if (rc != PT_OK) return ctx->fail(rc, "pt_selftest_guide_tree_device: the device-built tree breaks an invariant");
    return PT_OK;
}

}  // extern "C"
