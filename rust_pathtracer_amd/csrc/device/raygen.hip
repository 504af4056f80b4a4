// raygen.hip -- primary-ray generation: renderer.rs:22-26 + camera.rs:46-56.
// One lane per (sample slot, pixel).  Writes the wave's region of the ray streams, the path state (slot,
// throughput = 1), zeroes the path's radiance slot and publishes the wave's ray count for bounce 0.
#include "pt_launch.h"

__global__ __launch_bounds__(PT_BLOCK) void raygen_kernel(PtCamera cam, PassParams p, RayStream out, uint4* state,
                                                          float4* plane, uint32_t* counts, uint32_t n_waves, uint32_t region) {
    const uint32_t wave = blockIdx.x * (PT_BLOCK / PT_WAVE) + (threadIdx.x >> 6);
    const uint32_t lane = threadIdx.x & 63u;
    const V3 c_origin = v3(cam.origin[0], cam.origin[1], cam.origin[2]);
    const V3 c_llc = v3(cam.lower_left_corner[0], cam.lower_left_corner[1], cam.lower_left_corner[2]);
    const V3 c_hor = v3(cam.horizontal[0], cam.horizontal[1], cam.horizontal[2]);
    const V3 c_ver = v3(cam.vertical[0], cam.vertical[1], cam.vertical[2]);
    const V3 c_u = v3(cam.u[0], cam.u[1], cam.u[2]);
    const V3 c_v = v3(cam.v[0], cam.v[1], cam.v[2]);
    // chunk c of wave w is the (c * n_waves + w)-th group of 64 consecutive paths: every wave samples the whole image
    uint32_t mine = 0;
    for (uint32_t chunk = 0;; ++chunk) {
        const uint64_t first = ((uint64_t)chunk * n_waves + wave) * 64u;
        if (first >= p.n_paths) break;
        const uint32_t i = (uint32_t)first + lane;
        if (i < p.n_paths) {
            uint32_t slot = i / p.npix;
            uint32_t j = i - slot * p.npix;
            uint32_t x, row; // row counts from the top of the image (output order, renderer.rs:16 `.rev()`)
            if (p.tiled) {   // 64 consecutive paths = one 8x8 pixel tile: coherent primary rays
                uint32_t tile = j >> 6, tl = j & 63u;
                uint32_t tiles_x = p.width >> 3;
                uint32_t ty = tile / tiles_x, tx = tile - ty * tiles_x;
                x = tx * 8u + (tl & 7u);
                row = ty * 8u + (tl >> 3);
            } else {
                row = j / p.width;
                x = j - row * p.width;
            }
            uint32_t pixel = row * p.width + x;
            uint32_t y = p.height - 1u - row; // the reference's y counts from the bottom
            SampleRng rng;
            rng.init(p.seed, pixel, p.sample_base + slot, 0u);
            double xi1 = rng.f64();
            double xi2 = rng.f64();
            double u = ((double)x + xi1) / ((double)p.width - 1.0);  // renderer.rs:23
            double v = ((double)y + xi2) / ((double)p.height - 1.0); // renderer.rs:24
            // camera.rs:46-56 -- the reference draws the lens sample even when the aperture is 0.  Its draws are the last of stream 0
            // and the sample is multiplied by the lens radius: at radius 0 the offset is a vector of zeros whatever was drawn, so
            // the rejection loop (three quarters of this kernel's instructions) is skipped -- a wave-uniform branch.  (Only the SIGN
            // of those zeros is not reproduced; it could reach a result only through a camera origin component or an unblurred
            // direction component that is exactly -0.0, which no difference x - x and no look-at camera produces.)
            V3 rd = v3(0.0, 0.0, 0.0);
            if (cam.lens_radius != 0.0) rd = cam.lens_radius * random_in_unitdisk_xy(rng);
            V3 blur_offset = c_u * rd.x + c_v * rd.y;
            V3 o = c_origin + blur_offset;
            V3 d = c_llc + u * c_hor + v * c_ver - c_origin - blur_offset;
            const size_t e = (size_t)wave * region + (size_t)chunk * 64u + lane;
            out.a[e] = make_double2(o.x, o.y);
            out.b[e] = make_double2(o.z, d.x);
            out.c[e] = make_double2(d.y, d.z);
            uint32_t path = slot * p.npix + pixel;
            state[e] = make_uint4(path, __float_as_uint(1.0f), __float_as_uint(1.0f), __float_as_uint(1.0f));
            plane[path] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
        const uint64_t left = (uint64_t)p.n_paths - first;
        mine += left < 64u ? (uint32_t)left : 64u;
    }
    if (lane == 0) counts[wave] = mine;
    if (wave == 0 && lane == 0) counts[n_waves] = region;
}

void launch_raygen(hipStream_t s, const LaunchGeom& g, const PtCamera& cam, const PassParams& p, const Wavefront& wf) {
    hipLaunchKernelGGL(raygen_kernel, dim3(g.grid), dim3(PT_BLOCK), 0, s, cam, p, wf.rays[0], wf.state[0], wf.plane, wf.counts,
                       wf.n_waves, wf.region);
}
