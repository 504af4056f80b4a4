// Stream-layout bandwidth test for gfx950: the wavefront kernels' access pattern (6 input streams, 4 output streams,
// one 64-lane chunk per wave per step) under two layouts of a wave's chunks:
//   private     : wave w owns [w*region, (w+1)*region)             (chunks of one wave are contiguous)
//   interleaved : chunk k of wave w sits at (k*n_waves + w)*64     (chunks of neighbouring waves are contiguous)
// Build: hipcc --offload-arch=gfx950 -O3 scratch/membench.hip -o scratch/membench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

template <bool INTERLEAVED>
__global__ __launch_bounds__(256) void k_shade_like(const double2* __restrict__ a, const double2* __restrict__ b, const double2* __restrict__ c,
                                                    const uint4* __restrict__ st, const double* __restrict__ t, const int* __restrict__ prim,
                                                    double2* __restrict__ oa, double2* __restrict__ ob, double2* __restrict__ oc, uint4* __restrict__ ost,
                                                    unsigned n_waves, unsigned chunks_per_wave, int write) {
    const unsigned wave = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    for (unsigned k = 0; k < chunks_per_wave; ++k) {
        const size_t i = INTERLEAVED ? ((size_t)k * n_waves + wave) * 64 + lane : ((size_t)wave * chunks_per_wave + k) * 64 + lane;
        double2 va = a[i], vb = b[i], vc = c[i];
        uint4 vs = st[i];
        double vt = t[i];
        int vp = prim[i];
        va.x += vt; vs.x += vp;
        if (write) {
            oa[i] = va; ob[i] = vb; oc[i] = vc; ost[i] = vs;
        } else if (va.x == 1.2345 && vb.x == 2.0 && vc.y == 3.0 && vs.y == 77) {
            oa[i] = va;
        }
    }
}

int main() {
    CHECK(hipSetDevice(0));
    hipDeviceProp_t p;
    CHECK(hipGetDeviceProperties(&p, 0));
    const unsigned cus = p.multiProcessorCount;
    const unsigned grid = cus * 8, n_waves = grid * 4;
    const unsigned chunks = 128;                       // 8192 rays per wave
    const size_t n = (size_t)n_waves * chunks * 64;    // 67 M rays
    double2 *a, *b, *c, *oa, *ob, *oc; uint4 *st, *ost; double* t; int* prim;
    CHECK(hipMalloc(&a, n * 16)); CHECK(hipMalloc(&b, n * 16)); CHECK(hipMalloc(&c, n * 16)); CHECK(hipMalloc(&st, n * 16));
    CHECK(hipMalloc(&oa, n * 16)); CHECK(hipMalloc(&ob, n * 16)); CHECK(hipMalloc(&oc, n * 16)); CHECK(hipMalloc(&ost, n * 16));
    CHECK(hipMalloc(&t, n * 8)); CHECK(hipMalloc(&prim, n * 4));
    CHECK(hipMemset(a, 0, n * 16)); CHECK(hipMemset(b, 0, n * 16)); CHECK(hipMemset(c, 0, n * 16)); CHECK(hipMemset(st, 0, n * 16));
    CHECK(hipMemset(t, 0, n * 8)); CHECK(hipMemset(prim, 0, n * 4));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    for (int write = 0; write < 2; ++write)
        for (int inter = 0; inter < 2; ++inter) {
            float best = 1e9f;
            for (int rep = 0; rep < 4; ++rep) {
                CHECK(hipEventRecord(e0, 0));
                if (inter) hipLaunchKernelGGL(k_shade_like<true>, dim3(grid), dim3(256), 0, 0, a, b, c, st, t, prim, oa, ob, oc, ost, n_waves, chunks, write);
                else hipLaunchKernelGGL(k_shade_like<false>, dim3(grid), dim3(256), 0, 0, a, b, c, st, t, prim, oa, ob, oc, ost, n_waves, chunks, write);
                CHECK(hipEventRecord(e1, 0));
                CHECK(hipEventSynchronize(e1));
                float ms;
                CHECK(hipEventElapsedTime(&ms, e0, e1));
                if (ms < best) best = ms;
            }
            const double bytes = (double)n * (76 + (write ? 64 : 0));
            printf("%-12s %-10s %8.3f ms  %8.1f GB/s\n", inter ? "interleaved" : "private", write ? "read+write" : "read", best, bytes / best * 1e-6);
        }
    return 0;
}
