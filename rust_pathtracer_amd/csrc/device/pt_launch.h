// pt_launch.h -- host-callable launchers of the wavefront kernels (one .hip file per stage).
#pragma once
#include "pt_device.h"

// All streams of one wavefront pass.
//
// WAVE-PRIVATE REGIONS.  Every kernel of a pass runs the same persistent grid of n_waves wavefronts.  Wave w owns
// elements [w*region, (w+1)*region) of every stream for the whole pass: raygen fills its region (with every
// n_waves-th 64-path chunk of the image, so all waves see the same mix of pixels), trace and shade walk it in
// 64-ray chunks, and shade appends the survivors of each chunk -- ranked with __ballot + mbcnt -- at the wave's own
// running offset in the other ping-pong buffer.  Compaction therefore needs no atomics, no barriers and no scan;
// the order of rays in every stream is fixed, so a pass is reproducible down to the memory layout.
struct Wavefront {
    RayStream rays[2];   // ping-pong: shade reads rays[b&1], writes the compacted survivors to rays[(b+1)&1]
    uint4* state[2];     // x = path slot, yzw = throughput (f32 bits)
    double* hit_t;       // written by trace, read by shade
    int32_t* hit_prim;
    float4* plane;       // per-path radiance sum of the pass (one owner per slot: no atomics, bit-reproducible)
    // Per-bounce rows of n_waves + 1 words: row[w] = live rays of wave w entering the bounce, row[n_waves] = the
    // region size (elements per wave, a multiple of 64) of the layout those rays are stored in.  `counts` rows are
    // written by raygen / shade; `counts_repacked` rows by the repack kernel, which re-deals the survivors evenly.
    uint32_t* counts;
    uint32_t* counts_repacked;
    unsigned long long* totals; // totals[b] = running sum over waves and passes (ray statistics)
    uint32_t n_waves;
    uint32_t region;     // region size at bounce 0
};
inline size_t count_row(const Wavefront& wf, uint32_t bounce) { return (size_t)bounce * (wf.n_waves + 1u); }

// Which ping-pong buffer and which count row a bounce reads / writes (the host tracks them across repacks).
struct BounceIO {
    int in, out;
    const uint32_t* counts_in;
    uint32_t* counts_out;
};

struct PassParams {
    uint32_t width, height, npix;
    uint32_t n_paths;     // sample_slots * npix
    uint32_t sample_base; // absolute index of sample slot 0
    uint32_t max_bounces;
    uint64_t seed;
    float bg[3];
    int tiled; // 8x8 pixel tiles per wave in raygen (width and height multiples of 8)
};

struct LaunchGeom {
    uint32_t grid; // persistent grid of the pass kernels: grid * 4 = Wavefront::n_waves
    size_t trace_lds_bytes;
    bool all_lds;
};

void launch_raygen(hipStream_t s, const LaunchGeom& g, const PtCamera& cam, const PassParams& p, const Wavefront& wf);
void launch_trace(hipStream_t s, const LaunchGeom& g, const DevScene& sc, const Wavefront& wf, const BounceIO& io);
void launch_shade(hipStream_t s, const LaunchGeom& g, const DevScene& sc, const PassParams& p, const Wavefront& wf,
                  const BounceIO& io, uint32_t bounce);
// The whole bounce loop in one launch (path.hip); the count rows 1.. of wf.counts must have been zeroed.
void launch_path(hipStream_t s, const LaunchGeom& g, const DevScene& sc, const PassParams& p, const Wavefront& wf, const BounceIO& io,
                 uint32_t first_bounce);
// Bounces first_bounce.. of the pass in one launch, each wave on its own region (ordered walk only); io.in / io.counts_in: where
// first_bounce's rays and counts are.
void launch_tail(hipStream_t s, const LaunchGeom& g, const DevScene& sc, const PassParams& p, const Wavefront& wf, const BounceIO& io,
                 uint32_t first_bounce);
struct ShadeEnv;
ShadeEnv make_shade_env(const PassParams& p, const Wavefront& wf, uint32_t bounce);
// Re-deals the rays of buffer io.in evenly over the waves into buffer io.out (counts_in -> counts_out).
void launch_repack(hipStream_t s, const LaunchGeom& g, const Wavefront& wf, const BounceIO& io, uint32_t deal_waves = 0); // deal_waves > 0: dealt out one by one over that many waves
void launch_fold(hipStream_t s, const PassParams& p, const Wavefront& wf, uint32_t sample_slots, float* accum);
void launch_count_totals(hipStream_t s, const Wavefront& wf, uint32_t n_bounces);
void launch_resolve(hipStream_t s, const float* accum, uint32_t n_values, uint32_t total_spp, uint8_t* rgb);
void launch_trace_batch(hipStream_t s, const LaunchGeom& g, const DevScene& sc, const double* origins, const double* dirs,
                        uint32_t n, double t_min, double t_max, PtHitOut* out);
void launch_selftest_arith(hipStream_t s, const double* a, const double* b, uint32_t n, double* quot, double* root, int via_reciprocal);
void launch_selftest_rng(hipStream_t s, uint64_t seed, uint32_t pixel, uint32_t sample, uint32_t stream, uint32_t n, double* out);
void launch_selftest_texture(hipStream_t s, const DevScene& sc, uint32_t tex, const double* points, uint32_t n, double* rgb);
