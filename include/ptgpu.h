/* ptgpu.h -- C ABI of the MI355X path-tracing backend (libptgpu.so / libptgpu.a).
 *
 * The reference (ArcticXWolf/rust-pathtracer) has no FFI of its own: its seam is the Rust function
 *   renderer::render(world, camera, width, height, background, samples_per_pixel, max_bounces) -> Vec<u8>
 * (src/renderer.rs:5-13, one call site at src/main.rs:56-64) over the Hittable / Material / Texture trait
 * objects.  A GPU backend cannot introspect trait objects, so the host flattens the world into the POD tables
 * below and hands them over once (the reference builds the world once per run, src/main.rs:23), then renders
 * frames from it.  Every entry point here cites the reference code whose results it reproduces.
 *
 * Conventions: plain C, POD structs with fixed-width fields, no exceptions across the boundary.
 * Return 0 = success, negative = PtStatus error; the message is available from pt_last_error().
 * Host pointers are borrowed for the duration of the call only; the library owns all device memory
 * (except a buffer bound with pt_accum_bind) and frees it in pt_context_destroy().
 * One context per GPU; one host thread drives a context at a time; contexts are independent.
 * Calls return after the context's stream has drained.
 */
#ifndef PTGPU_H
#define PTGPU_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum PtStatus {
    PT_OK = 0,
    PT_ERR_BAD_ARG = -1,     /* null pointer, out-of-range index, malformed scene */
    PT_ERR_HIP = -2,         /* a HIP runtime call failed (message carries hipGetErrorString) */
    PT_ERR_OOM = -3,         /* device allocation failed */
    PT_ERR_NO_SCENE = -4,    /* render / trace before pt_scene_upload */
    PT_ERR_NO_DEVICE = -5,   /* no usable GPU */
    PT_ERR_STATE = -6,       /* call out of order (e.g. resolve before pt_accum_reset) */
    PT_ERR_RCCL = -7         /* librccl could not be loaded, or an RCCL call failed (message carries ncclGetErrorString) */
} PtStatus;

/* ---- primitives: geometry.rs Sphere (:104-171), RectangleXY/XZ/YZ (:182-369), Triangle (:466-589).
 * An AABox (:371-464) is six consecutive rectangles in the order XY-,XY+,XZ-,XZ+,YZ-,YZ+ referenced as a group.
 * d[] by type:
 *   SPHERE    d[0..2] center, d[3] radius (may be negative: hollow glass, normal flips)
 *   RECT_XY   d[0] plane z, d[1] x_lo, d[2] x_hi, d[3] y_lo, d[4] y_hi, d[5] direction (+1/-1, already signum'd)
 *   RECT_XZ   d[0] plane y, d[1] x_lo, d[2] x_hi, d[3] z_lo, d[4] z_hi, d[5] direction
 *   RECT_YZ   d[0] plane x, d[1] y_lo, d[2] y_hi, d[3] z_lo, d[4] z_hi, d[5] direction
 *   TRIANGLE  d[0..2] point1, d[3..5] point2, d[6..8] point3, d[9..11] stored normal (returned verbatim)
 * Primitives MUST be listed in the reference's depth-first (left before right, list order inside a group)
 * order: the index of a primitive in this table is the "hit primitive index" of the parity contract. */
enum { PT_PRIM_SPHERE = 0, PT_PRIM_RECT_XY = 1, PT_PRIM_RECT_XZ = 2, PT_PRIM_RECT_YZ = 3, PT_PRIM_TRIANGLE = 4 };
typedef struct PtPrim {
    uint32_t type;
    uint32_t material;
    double d[12];
} PtPrim;

/* ---- BVH: bvh.rs BvhNode (:57-126).  A child reference is either a node index or, with PT_REF_LEAF set,
 * a primitive: bits 0..29 = index of the first primitive, PT_REF_GROUP6 set = the six rectangles of an AABox
 * starting there (scanned in order, geometry.rs:70-83).  A nested ObjModel tree (obj_model.rs:119-121) is
 * spliced in as an ordinary child node.  Only node boxes are ever tested (bvh.rs:108). */
#define PT_REF_LEAF 0x80000000u
#define PT_REF_GROUP6 0x40000000u
#define PT_REF_INDEX_MASK 0x3FFFFFFFu
typedef struct PtNode {
    double bmin[3];
    double bmax[3];
    uint32_t left;
    uint32_t right;
} PtNode;

/* ---- materials: material.rs Lambertian (:24-56), Metal (:58-94), Dielectric (:96-144), DiffuseLight (:146-167) */
enum { PT_MAT_LAMBERTIAN = 0, PT_MAT_METAL = 1, PT_MAT_DIELECTRIC = 2, PT_MAT_DIFFUSE_LIGHT = 3 };
typedef struct PtMaterial {
    uint32_t type;
    uint32_t texture; /* albedo (lambertian, metal) or emit (diffuse light); ignored for dielectric */
    double param;     /* metal: fuzz (unclamped); dielectric: index of refraction */
} PtMaterial;

/* ---- textures: texture.rs SolidColor (:9-23), Checker (:25-45), PerlinNoise marble (:47-81) */
enum { PT_TEX_SOLID = 0, PT_TEX_CHECKER = 1, PT_TEX_PERLIN = 2 };
typedef struct PtTexture {
    uint32_t type;
    uint32_t odd;          /* checker: texture used where the sine product is < 0 */
    uint32_t even;         /* checker: texture used otherwise */
    uint32_t perlin_table; /* perlin: index of a 256-byte permutation in PtSceneDesc.perlin_tables */
    double color[3];       /* solid */
    double scale;          /* perlin */
} PtTexture;

typedef struct PtSceneDesc {
    const PtNode* nodes;
    uint32_t n_nodes;
    uint32_t root; /* child reference of the root (a node for every world BvhNode::new accepts) */
    const PtPrim* prims;
    uint32_t n_prims;
    const PtMaterial* materials;
    uint32_t n_materials;
    const PtTexture* textures;
    uint32_t n_textures;
    const uint8_t* perlin_tables; /* n_perlin_tables * 256 bytes */
    uint32_t n_perlin_tables;
} PtSceneDesc;

/* ---- camera.rs Camera fields (:3-11), as computed by Camera::new (:14-44) on the host */
typedef struct PtCamera {
    double origin[3];
    double lower_left_corner[3];
    double horizontal[3];
    double vertical[3];
    double u[3];
    double v[3];
    double lens_radius;
} PtCamera;

/* ---- result of one BvhNode::hit query (bvh.rs:107-121 -> geometry.rs HitRecord :9-44) */
typedef struct PtHitOut {
    int32_t prim;        /* index into PtSceneDesc.prims, -1 = miss */
    uint32_t front_face; /* geometry.rs:29 */
    double t;
    double normal[3];    /* flipped to oppose the ray, geometry.rs:33-37 */
} PtHitOut;

/* Per-call statistics of pt_render_accumulate.  Kernel times are HIP-event times on the stream each kernel is launched on.
 * ms_<kernel> is the SUM of that kernel's launch durations; with several lanes (sub-passes on separate streams) launches
 * of one kernel overlap each other and the other kernels, so the sums can exceed ms_total.  ms_busy_<kernel> is the length
 * of the UNION of the same intervals: the time during which at least one launch of that kernel was running. */
typedef struct PtStats {
    uint64_t rays;            /* hit queries issued (= Ray::color calls with bounces_left > 0, ray.rs:25) */
    uint64_t paths;           /* primary samples started */
    uint32_t passes;          /* wavefront passes the call was split into */
    uint32_t launches_trace;  /* trace kernel launches */
    uint32_t launches_shade;
    uint32_t launches_path;   /* fused path-kernel / tail-kernel launches (at most one per sub-pass) */
    double ms_total;          /* first launch to last completion */
    double ms_raygen;
    double ms_trace;
    double ms_shade;
    double ms_fold;
    double ms_repack;
    double ms_path;           /* the fused path kernel / the tail kernel (trace + shade of several bounces in one launch) */
    double ms_busy_trace;     /* union of the trace launches' intervals */
    double ms_busy_shade;
    double ms_busy_path;
    double ms_busy_all;       /* union over every kernel of the call (<= ms_total; the rest is launch gaps) */
    uint64_t rays_per_bounce[64]; /* live rays entering bounce b (b >= 64 folded into the last slot) */
    uint32_t tail_from;       /* bounces [0, tail_from) ran as one trace and one shade launch each, bounces [tail_from, max_bounces) inside
                                 the fused path / tail launch (= max_bounces when no such launch ran): what splits rays_per_bounce
                                 between the kernels */
    uint32_t reserved_;
} PtStats;

typedef struct PtContext PtContext;

/* Number of HIP devices visible to the process; negative PtStatus on failure. */
int pt_device_count(void);

/* Creates a context on `device`: its stream, and nothing else until a scene is uploaded. */
int pt_context_create(int device, PtContext** out);
void pt_context_destroy(PtContext* ctx);

/* Message of the last failing call on ctx (or of the last failing pt_context_create when ctx is NULL). */
const char* pt_last_error(const PtContext* ctx);

/* Options: "paths_per_pass" (upper bound on paths in flight, 156 B each; default 512 Mi = 84 GB, halved until the streams fit the device),
 * "profile" (1 = time every kernel with HIP events and fill PtStats.ms_*; default 1),
 * "repack_every" / "repack_first" (re-deal surviving rays over the waves every N bounces from bounce F on;
 *   default 0 = never / 8),
 * "lanes" (1..8: split every pass into that many sub-passes on separate HIP streams, one pipeline step apart, so one's
 *   trace overlaps another's shade; default 0 = automatic: 4 for scenes on the ordered walk, 3 otherwise), "lane_sync" (1 = lanes advance in lock step through events; default 0
 *   = free-running), "lane_full_grid" (1 = every lane launches the whole persistent grid, 0 = its share of it; default 0),
 * "fused" (1 = scenes that take the lockstep walk and have only solid textures run the whole bounce loop of a sub-pass in one
 *   launch, each wave on its own region; 0 = one trace and one shade launch per bounce; default 0 -- the fused kernel needs 87
 *   VGPRs, runs 5 waves per SIMD instead of 8 and measures 10-15 % slower on the Cornell box; results are identical),
 * "trace_mode" (which kernel answers the hit queries, applied at the next pt_scene_upload: -1 = by scene size (0 up to 192
 *   nodes + primitives, 3 beyond), 0 = wave-uniform lockstep walk, 1 = per-lane walk with an LDS stack, 2 = per-lane stackless
 *   walk of the record stream, 3 = ordered walk: nearest-first over a surface-area guide tree built at upload, then a
 *   proof per ray that the reference's walk returns the same primitive, else the walk of mode 2 for that ray; falls back
 *   to 2 when the scene's boxes are not finite and nested; all four give identical results),
 * "tail_from" (bounces from this one on run in ONE launch per sub-pass, each wave looping over its own rays -- for scenes
 *   on the ordered walk, and for scenes on the lockstep walk with solid textures ("fused" = from bounce 0); -1 = automatic:
 *   4 for ordered-walk scenes of 65536 primitives or more, never otherwise (measured on the Cornell box: 20.5 / 21.3 / 22.2 ms
 *   per 32-spp frame for never / 16 / 12); results are identical),
 * "guide_build" (who builds the ordered walk's guide tree at the next pt_scene_upload: 0 = the host, binned surface-area splits
 *   (0.45 s for a million triangles; the default: its trees take ~12 % fewer steps per ray); 1 = the device, linear BVH
 *   (7 ms of kernels for a million triangles: for worlds rebuilt every frame); results are identical either way),
 * "tail_deal" (1, the default: the rays that enter that one launch are first dealt out over the waves one at a time -- the long-lived
 *   paths of a frame sit in a few image tiles, so in a few waves, and a tail wave's cost per bounce is that of its slowest ray; 0 = each
 *   wave keeps its own) and "tail_waves_div" (dealt over 1 / this many of the waves; default 2),
 * "guide_split" (guide tree, at the next pt_scene_upload: a triangle whose box is more than this many times its own projected areas is
 *   referenced once per PIECE, cut by axis-aligned planes, at most 32 pieces and n_prims / 2 extra references per scene -- the
 *   slivers around the pole of a latitude-longitude mesh; default 4, 0 = never; results are identical),
 * "deep_tree" (ordered walk, at the next pt_scene_upload: 1 = the wave loop hands out rays earlier and scales its waiting thresholds with
 *   the lanes that hold a ray -- for trees in which most rays leave after a few steps and a few take hundreds; -1, the default = for
 *   scenes of 65536 primitives or more; results are identical),
 * "blocks_per_cu" (persistent-grid size, applied at the next pt_scene_upload; 0 = automatic: 8, or 6 for procedural textures off the
 *   ordered walk). */
int pt_context_set_option(PtContext* ctx, const char* key, int64_t value);

/* Replaces scene.get_world()'s result (scene.rs:38-42): validates and uploads the flattened world.
 * Errors (PT_ERR_BAD_ARG) mirror the reference's panics: fewer than two objects / empty tree (bvh.rs:76). */
int pt_scene_upload(PtContext* ctx, const PtSceneDesc* scene);

/* BvhNode::hit (bvh.rs:107-121) for n rays given as host arrays origins[n*3], dirs[n*3]; out[n].
 * This is the parity entry point: same device traversal code as the render path's trace kernel. */
int pt_trace_batch(PtContext* ctx, const double* origins, const double* dirs, size_t n, double t_min, double t_max,
                   PtHitOut* out);

/* Sizes (and zeroes) the accumulation buffer: width*height*3 f32 sums, rows top-first (renderer.rs:14-18). */
int pt_accum_reset(PtContext* ctx, uint32_t width, uint32_t height);

/* Uses caller-owned device memory (width*height*3 floats on this context's device) as the accumulation
 * buffer, e.g. a torch tensor that torch.distributed will reduce.  NULL returns to the internal buffer.
 * Must follow pt_accum_reset; the buffer is NOT cleared by this call. */
int pt_accum_bind(PtContext* ctx, void* device_ptr);

/* The spp loop of renderer::render (renderer.rs:19-28) for samples [sample_begin, sample_begin+sample_count)
 * of every pixel: adds each sample's colour to the accumulation buffer.  Sample s of pixel p draws the same
 * random numbers wherever and whenever it is computed (counter-based RNG keyed by seed, p, s, bounce), so
 * disjoint sample ranges rendered on different GPUs sum to the single-GPU result.  stats may be NULL. */
int pt_render_accumulate(PtContext* ctx, const PtCamera* camera, uint32_t width, uint32_t height, const double background[3],
                         uint32_t sample_begin, uint32_t sample_count, uint32_t max_bounces, uint64_t seed, PtStats* stats);

/* renderer.rs:30-33 + vec3.rs:278-290: sum/total_spp -> sqrt -> (c*255.999) saturating u8; rgb_out = host
 * buffer of width*height*3 bytes, top row first, RGB. */
int pt_resolve(PtContext* ctx, uint32_t total_spp, uint8_t* rgb_out);

/* Copies the accumulation buffer to / from host memory (width*height*3 floats). */
int pt_accum_download(PtContext* ctx, float* rgb_sum);
int pt_accum_upload(PtContext* ctx, const float* rgb_sum);

/* Device address of the current accumulation buffer (internal or bound); NULL before pt_accum_reset. */
void* pt_accum_device_ptr(PtContext* ctx);

/* The exchange step of a sample-sharded frame (the loop being sharded is renderer.rs:19-28): sums the
 * accumulation buffers of n contexts -- one per GPU, all of one size, each after its own pt_render_accumulate calls
 * on its own sample range -- into ctxs[0]'s buffer with ONE RCCL ncclReduce(f32, sum, root 0) over xGMI
 * (width*height*3 floats: 24.9 MB at 1080p, 99.5 MB at 4K), each rank's part enqueued on that context's stream
 * inside one ncclGroupStart/End.  The other contexts' buffers are left as they were.  Returns after every stream has
 * drained; ctxs[0] can then be resolved with the TOTAL spp.  n == 1 is the identity (and still goes through RCCL).
 * The communicators (ncclCommInitAll over the contexts' devices, in the order given) are created on the first call with a
 * given list of contexts and kept until one of those contexts is destroyed (pt_context_destroy calls ncclCommDestroy).  librccl.so.1 is loaded on the first call, not at link time: a host that never shards needs no
 * RCCL.  One thread calls this for all n contexts, after the threads that drove them have returned from their calls.
 * PT_ERR_BAD_ARG: null / repeated context, two contexts on one device, sizes differ; PT_ERR_STATE: before
 * pt_accum_reset; PT_ERR_RCCL: library missing or an RCCL call failed. */
int pt_accum_reduce(PtContext** ctxs, int n);

/* pt_trace_batch for callers that cannot allocate device memory (tight serving loops, harnesses that forbid hipMalloc, code
 * that runs between the captures of a HIP graph and must not disturb the allocator).  No counterpart in the reference.  Same
 * arguments, same kernels and bit-identical results, but the device copies of the rays and of the hit records live in a fixed
 * scratch array inside the library (a file-scope __device__ array, PT_TRACE_STATIC_MAX_RAYS * 88 bytes = 5.5 MiB held per GPU
 * for the life of the process) instead of buffers allocated and freed per call: the call makes no hipMalloc / hipFree and no
 * other allocating call.  It is NOT itself capturable into a HIP graph: like pt_trace_batch it copies from and to pageable
 * host memory and returns after its stream has drained.  n > PT_TRACE_STATIC_MAX_RAYS is refused with PT_ERR_BAD_ARG.  All
 * calls on one GPU share that GPU's array, whatever the context: within one process the library serializes them (a per-GPU
 * mutex held for the call); calls from DIFFERENT processes cannot overlap on the array (each process has its own copy). */
#define PT_TRACE_STATIC_MAX_RAYS 65536u
int pt_trace_batch_static(PtContext* ctx, const double* origins, const double* dirs, size_t n, double t_min, double t_max,
                          PtHitOut* out);

/* Device-side checks of the arithmetic the parity contract leans on (IEEE division and sqrt, the Philox
 * stream): computes a[i]/b[i] and sqrt(|a[i]|) on the GPU for host arrays of n doubles. */
int pt_selftest_arith(PtContext* ctx, const double* a, const double* b, size_t n, double* quot, double* root);
/* a[i]/b[i] as the lockstep trace kernel forms its rectangle quotients (from the correctly rounded reciprocal, one
 * FMA residual and one correction, with the IEEE division as the fallback): must equal pt_selftest_arith's quot. */
int pt_selftest_division(PtContext* ctx, const double* a, const double* b, size_t n, double* quot);
/* First n draws of stream (seed, pixel, sample, stream) as produced on the GPU. */
int pt_selftest_rng(PtContext* ctx, uint64_t seed, uint32_t pixel, uint32_t sample, uint32_t stream, uint32_t n, double* out);
/* Texture.value (texture.rs:5-7) of texture `tex` of the uploaded scene at n points (host arrays). */
int pt_selftest_texture(PtContext* ctx, uint32_t tex, const double* points, size_t n, double* rgb_out);
/* Host-only check of the guide-tree builder of the ordered walk ("trace_mode" 3): builds the tree over n >= 2 boxes
 * (boxes[6*i .. 6*i+5] = lo.xyz, hi.xyz) and verifies that every box is a leaf exactly once and lies inside every box
 * stored on the way down to it.  Touches no device.  PT_ERR_BAD_ARG for unusable boxes, PT_ERR_STATE for a broken tree. */
int pt_selftest_guide_tree(const double* boxes, size_t n, uint32_t* n_nodes, uint32_t* max_depth);
/* The same check of the DEVICE builder ("guide_build" 1: Morton sort + Karras hierarchy + bottom-up refit on ctx's GPU): the
 * tree is built on the device, copied back and verified on the host. */
int pt_selftest_guide_tree_device(PtContext* ctx, const double* boxes, size_t n, uint32_t* n_nodes, uint32_t* max_depth);
/* Host-only: the pieces by which the guide tree references one triangle (corners: 3 x xyz; option "guide_split" = ratio): cut by
 * axis-aligned planes until a piece's box is at most `ratio` times the piece's own projected areas (at most 32 pieces).  Writes
 * up to `capacity` unpadded boxes (lo.xyz, hi.xyz) to `boxes` (may be null) and the number of pieces to *n_pieces. */
int pt_selftest_split_triangle(const double corners[9], int ratio, double* boxes, size_t capacity, size_t* n_pieces);

#ifdef __cplusplus
}
#endif
#endif /* PTGPU_H */
