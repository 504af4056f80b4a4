// resolve.hip -- the per-pixel tail of renderer::render (renderer.rs:27-33) and small utility kernels.
#include "pt_launch.h"

// color_sampling += sample (renderer.rs:27): the pass's per-path radiance slots are summed per pixel in
// sample-slot order (a fixed order: the result is bit-reproducible), then added to the running sum.
__global__ __launch_bounds__(PT_BLOCK) void fold_kernel(const float4* plane, uint32_t npix, uint32_t sample_slots, float* accum) {
    const uint32_t stride = gridDim.x * blockDim.x;
    for (uint32_t pix = blockIdx.x * blockDim.x + threadIdx.x; pix < npix; pix += stride) {
        float r = 0.f, g = 0.f, b = 0.f;
        for (uint32_t s = 0; s < sample_slots; ++s) {
            float4 v = plane[(size_t)s * npix + pix];
            r += v.x;
            g += v.y;
            b += v.z;
        }
        float* dst = accum + (size_t)pix * 3;
        dst[0] += r;
        dst[1] += g;
        dst[2] += b;
    }
}

void launch_fold(hipStream_t s, const PassParams& p, const Wavefront& wf, uint32_t sample_slots, float* accum) {
    uint32_t blocks = (p.npix + PT_BLOCK - 1) / PT_BLOCK;
    if (blocks > 8192u) blocks = 8192u;
    hipLaunchKernelGGL(fold_kernel, dim3(blocks), dim3(PT_BLOCK), 0, s, wf.plane, p.npix, sample_slots, accum);
}

// totals[b] += sum over waves of counts[b][w]: one block per bounce.
__global__ __launch_bounds__(PT_BLOCK) void count_totals_kernel(const uint32_t* counts, unsigned long long* totals, uint32_t n_waves) {
    __shared__ unsigned long long s_sum[PT_BLOCK];
    const uint32_t* row = counts + (size_t)blockIdx.x * (n_waves + 1u);
    unsigned long long acc = 0;
    for (uint32_t w = threadIdx.x; w < n_waves; w += PT_BLOCK) acc += row[w];
    s_sum[threadIdx.x] = acc;
    __syncthreads();
    for (uint32_t half = PT_BLOCK / 2; half > 0; half >>= 1) {
        if (threadIdx.x < half) s_sum[threadIdx.x] += s_sum[threadIdx.x + half];
        __syncthreads();
    }
    if (threadIdx.x == 0) totals[blockIdx.x] += s_sum[0];
}
void launch_count_totals(hipStream_t s, const Wavefront& wf, uint32_t n_bounces) {
    hipLaunchKernelGGL(count_totals_kernel, dim3(n_bounces), dim3(PT_BLOCK), 0, s, wf.counts, wf.totals, wf.n_waves);
}

// (sum / spp).sqrt() then `(c * 255.999) as u8` (renderer.rs:30-33, vec3.rs:278-290): Rust's float->int
// cast saturates and maps NaN to 0.
__global__ __launch_bounds__(PT_BLOCK) void resolve_kernel(const float* accum, uint32_t n_values, double spp, uint8_t* rgb) {
    const uint32_t stride = gridDim.x * blockDim.x;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n_values; i += stride) {
        double c = sqrt((double)accum[i] / spp);
        double v = c * 255.999;
        uint8_t q;
        if (!(v == v)) q = 0;
        else if (v <= 0.0) q = 0;
        else if (v >= 255.0) q = 255;
        else q = (uint8_t)v;
        rgb[i] = q;
    }
}
void launch_resolve(hipStream_t s, const float* accum, uint32_t n_values, uint32_t total_spp, uint8_t* rgb) {
    uint32_t blocks = (n_values + PT_BLOCK - 1) / PT_BLOCK;
    if (blocks > 8192u) blocks = 8192u;
    if (blocks == 0) blocks = 1;
    hipLaunchKernelGGL(resolve_kernel, dim3(blocks), dim3(PT_BLOCK), 0, s, accum, n_values, (double)total_spp, rgb);
}

// ---- device-side checks of the arithmetic the parity contract leans on
// via_reciprocal: the quotient as the lockstep walk forms it (div_given_reciprocal); n is padded to whole waves by the
// caller's grid, and every lane of a wave takes part in the wave-wide fallback vote.
__global__ void selftest_arith_kernel(const double* a, const double* b, uint32_t n, double* quot, double* root, int via_reciprocal) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    const bool live = i < n;
    const double x = live ? a[i] : 1.0, y = live ? b[i] : 1.0;
    const double q = via_reciprocal ? div_given_reciprocal(x, y, 1.0 / y) : x / y;
    if (live) {
        quot[i] = q;
        root[i] = sqrt(fabs(x));
    }
}
void launch_selftest_arith(hipStream_t s, const double* a, const double* b, uint32_t n, double* quot, double* root, int via_reciprocal) {
    hipLaunchKernelGGL(selftest_arith_kernel, dim3((n + 255) / 256), dim3(256), 0, s, a, b, n, quot, root, via_reciprocal);
}
__global__ void selftest_rng_kernel(uint64_t seed, uint32_t pixel, uint32_t sample, uint32_t stream, uint32_t n, double* out) {
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        SampleRng rng;
        rng.init(seed, pixel, sample, stream);
        for (uint32_t i = 0; i < n; ++i) out[i] = rng.f64();
    }
}
void launch_selftest_rng(hipStream_t s, uint64_t seed, uint32_t pixel, uint32_t sample, uint32_t stream, uint32_t n, double* out) {
    hipLaunchKernelGGL(selftest_rng_kernel, dim3(1), dim3(64), 0, s, seed, pixel, sample, stream, n, out);
}
__global__ void selftest_texture_kernel(DevScene sc, uint32_t tex, const double* points, uint32_t n, double* rgb) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        V3 c = texture_value<true>(sc, tex, v3(points[3 * (size_t)i], points[3 * (size_t)i + 1], points[3 * (size_t)i + 2]));
        rgb[3 * (size_t)i] = c.x;
        rgb[3 * (size_t)i + 1] = c.y;
        rgb[3 * (size_t)i + 2] = c.z;
    }
}
void launch_selftest_texture(hipStream_t s, const DevScene& sc, uint32_t tex, const double* points, uint32_t n, double* rgb) {
    hipLaunchKernelGGL(selftest_texture_kernel, dim3((n + 255) / 256), dim3(256), 0, s, sc, tex, points, n, rgb);
}
