// Instruction-rate microbenchmark, part 3 (f16 -> f32 paths of the guide tree's box test).  Build: hipcc --offload-arch=gfx950 -O2 scratch/ubench3.hip -o scratch/ubench3
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
constexpr int UNROLL = 8, REPS = 16;
#define DEF_KERNEL_U(NAME, ASM)                                                                          \
    __global__ void NAME(double* out, int iters, double seed) {                                          \
        unsigned a[UNROLL];                                                                              \
        for (int k = 0; k < UNROLL; ++k) a[k] = 0x3c003c00u + k + threadIdx.x;                            \
        unsigned b = 0x3f800000u, c = 0x3c003c00u;                                                       \
        for (int i = 0; i < iters; ++i) {                                                                \
            _Pragma("unroll") for (int r = 0; r < REPS; ++r) {                                           \
                _Pragma("unroll") for (int k = 0; k < UNROLL; ++k) { asm volatile(ASM : "+v"(a[k]) : "v"(b), "v"(c)); } \
            }                                                                                            \
        }                                                                                                \
        unsigned s = 0;                                                                                  \
        for (int k = 0; k < UNROLL; ++k) s += a[k];                                                      \
        if (s == 123456789u) out[threadIdx.x] = s;                                                       \
    }
#define DEF_KERNEL_D(NAME, ASM)                                                                          \
    __global__ void NAME(double* out, int iters, double seed) {                                          \
        double a[UNROLL];                                                                                \
        for (int k = 0; k < UNROLL; ++k) a[k] = seed + k + threadIdx.x * 1e-3;                            \
        double b = seed * 0.999, c = seed * 1.0001;                                                      \
        for (int i = 0; i < iters; ++i) {                                                                \
            _Pragma("unroll") for (int r = 0; r < REPS; ++r) {                                           \
                _Pragma("unroll") for (int k = 0; k < UNROLL; ++k) { asm volatile(ASM : "+v"(a[k]) : "v"(b), "v"(c)); } \
            }                                                                                            \
        }                                                                                                \
        double s = 0;                                                                                    \
        for (int k = 0; k < UNROLL; ++k) s += a[k];                                                      \
        if (s == 12345.678) out[threadIdx.x] = s;                                                        \
    }
DEF_KERNEL_U(k_fma_f32, "v_fma_f32 %0, %0, %1, %2")
DEF_KERNEL_U(k_fma_mix_lo, "v_fma_mix_f32 %0, %2, %1, %0 op_sel:[0,0,0] op_sel_hi:[1,0,0]")
DEF_KERNEL_U(k_fma_mix_hi, "v_fma_mix_f32 %0, %2, %1, %0 op_sel:[1,0,0] op_sel_hi:[1,0,0]")
DEF_KERNEL_U(k_fma_mix_f32only, "v_fma_mix_f32 %0, %2, %1, %0 op_sel:[0,0,0] op_sel_hi:[0,0,0]")
DEF_KERNEL_U(k_cvt_f32_f16, "v_cvt_f32_f16 %0, %2")
DEF_KERNEL_U(k_cvt_f32_f16_hi, "v_cvt_f32_f16_sdwa %0, %2 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1")
DEF_KERNEL_U(k_cvt_ubyte, "v_cvt_f32_ubyte1 %0, %2")
DEF_KERNEL_U(k_max3, "v_max3_f32 %0, %0, %1, %2")
DEF_KERNEL_U(k_min, "v_min_f32 %0, %0, %1")
DEF_KERNEL_U(k_pk_fma_f16, "v_pk_fma_f16 %0, %0, %1, %2")
DEF_KERNEL_D(k_pk_fma_f32, "v_pk_fma_f32 %0, %0, %1, %2")
DEF_KERNEL_D(k_pk_add_f32, "v_pk_add_f32 %0, %0, %1")
DEF_KERNEL_D(k_pk_mul_f32, "v_pk_mul_f32 %0, %0, %1")
typedef void (*Kern)(double*, int, double);
struct Entry { const char* name; Kern k; };
int main() {
    CHECK(hipSetDevice(0));
    hipDeviceProp_t p;
    CHECK(hipGetDeviceProperties(&p, 0));
    const int cus = p.multiProcessorCount;
    const double mhz = p.clockRate / 1000.0;
    printf("device %s, %d CUs, clock %.0f MHz\n", p.name, cus, mhz);
    double* out;
    CHECK(hipMalloc(&out, 4096));
    Entry es[] = {{"v_fma_f32", k_fma_f32}, {"v_fma_mix_f32 (f16 lo)", k_fma_mix_lo}, {"v_fma_mix_f32 (f16 hi)", k_fma_mix_hi}, {"v_fma_mix_f32 (all f32)", k_fma_mix_f32only},
                  {"v_cvt_f32_f16", k_cvt_f32_f16}, {"v_cvt_f32_f16 sdwa WORD_1", k_cvt_f32_f16_hi}, {"v_cvt_f32_ubyte1", k_cvt_ubyte}, {"v_max3_f32", k_max3}, {"v_min_f32", k_min},
                  {"v_pk_fma_f16", k_pk_fma_f16}, {"v_pk_fma_f32", k_pk_fma_f32}, {"v_pk_add_f32", k_pk_add_f32}, {"v_pk_mul_f32", k_pk_mul_f32}};
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    const int iters = 2000;
    for (int waves_per_simd : {1, 4, 8}) {
        printf("---- %d wave(s) per SIMD: cycles per wave64 instruction per SIMD (at the reported clock)\n", waves_per_simd);
        for (auto& e : es) {
            dim3 grid(cus * waves_per_simd), block(256);
            hipLaunchKernelGGL(e.k, grid, block, 0, 0, out, 10, 1.5);
            CHECK(hipDeviceSynchronize());
            CHECK(hipEventRecord(e0, 0));
            hipLaunchKernelGGL(e.k, grid, block, 0, 0, out, iters, 1.5);
            CHECK(hipEventRecord(e1, 0));
            CHECK(hipEventSynchronize(e1));
            float ms;
            CHECK(hipEventElapsedTime(&ms, e0, e1));
            const double cyc = ms * 1e-3 * mhz * 1e6 / ((double)iters * REPS * UNROLL * waves_per_simd);
            printf("%-28s %8.2f\n", e.name, cyc);
        }
    }
    return 0;
}
