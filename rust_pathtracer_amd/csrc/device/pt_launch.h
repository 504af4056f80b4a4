// pt_launch.h -- host-callable launchers of the wavefront kernels (one .hip file per stage).
#pragma once
#include "pt_device.h"

// All streams of one wavefront pass.  Capacity = paths in flight.
struct Wavefront {
    RayStream rays[2];   // ping-pong: shade reads rays[b&1], writes the compacted survivors to rays[(b+1)&1]
    uint4* state[2];     // x = path slot, yzw = throughput (f32 bits)
    double* hit_t;       // written by trace, read by shade
    int32_t* hit_prim;
    float4* plane;       // per-path radiance sum of the pass (one owner per slot: no atomics, bit-reproducible)
    uint32_t* counts;    // counts[b] = live rays entering bounce b; zeroed per pass, bumped by shade's compaction
    unsigned long long* totals; // running sum of counts[] over passes (ray statistics)
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
    uint32_t grid; // persistent grid: blocks loop over the live rays
    size_t trace_lds_bytes;
    bool all_lds;
};

void launch_raygen(hipStream_t s, const LaunchGeom& g, const PtCamera& cam, const PassParams& p, const Wavefront& wf);
void launch_trace(hipStream_t s, const LaunchGeom& g, const DevScene& sc, const Wavefront& wf, uint32_t bounce);
void launch_shade(hipStream_t s, const LaunchGeom& g, const DevScene& sc, const PassParams& p, const Wavefront& wf,
                  uint32_t bounce);
void launch_fold(hipStream_t s, const PassParams& p, const Wavefront& wf, uint32_t sample_slots, float* accum);
void launch_count_totals(hipStream_t s, const Wavefront& wf, uint32_t n_counts);
void launch_resolve(hipStream_t s, const float* accum, uint32_t n_values, uint32_t total_spp, uint8_t* rgb);
void launch_trace_batch(hipStream_t s, const LaunchGeom& g, const DevScene& sc, const double* origins, const double* dirs,
                        uint32_t n, double t_min, double t_max, PtHitOut* out);
void launch_selftest_arith(hipStream_t s, const double* a, const double* b, uint32_t n, double* quot, double* root);
void launch_selftest_rng(hipStream_t s, uint64_t seed, uint32_t pixel, uint32_t sample, uint32_t stream, uint32_t n, double* out);
void launch_selftest_texture(hipStream_t s, const DevScene& sc, uint32_t tex, const double* points, uint32_t n, double* rgb);
