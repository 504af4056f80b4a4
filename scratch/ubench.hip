// Instruction-rate microbenchmark for gfx950: issue cost (cycles per wave64 instruction per SIMD) of the VALU
// instructions the trace / shade kernels are made of.  Build: hipcc --offload-arch=gfx950 -O2 scratch/ubench.hip -o scratch/ubench
// Run on the GPU box: ./scratch/ubench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

constexpr int UNROLL = 8;   // independent chains
constexpr int REPS = 16;    // instructions per chain per loop iteration

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

#define DEF_KERNEL_U(NAME, ASM)                                                                          \
    __global__ void NAME(double* out, int iters, double seed) {                                          \
        asm volatile("s_mov_b64 vcc, 0x5555\n s_mov_b64 s[20:21], 0x3333" ::: "vcc", "s20", "s21");        \
        unsigned a[UNROLL];                                                                              \
        for (int k = 0; k < UNROLL; ++k) a[k] = (unsigned)seed + k + threadIdx.x;                         \
        unsigned b = (unsigned)(seed * 977), c = (unsigned)(seed * 131);                                 \
        for (int i = 0; i < iters; ++i) {                                                                \
            _Pragma("unroll") for (int r = 0; r < REPS; ++r) {                                           \
                _Pragma("unroll") for (int k = 0; k < UNROLL; ++k) { asm volatile(ASM : "+v"(a[k]) : "v"(b), "v"(c)); } \
            }                                                                                            \
        }                                                                                                \
        unsigned s = 0;                                                                                  \
        for (int k = 0; k < UNROLL; ++k) s += a[k];                                                      \
        if (s == 123456789u) out[threadIdx.x] = s;                                                       \
    }

DEF_KERNEL_D(k_fma_f64, "v_fma_f64 %0, %0, %1, %2")
DEF_KERNEL_D(k_mul_f64, "v_mul_f64 %0, %0, %1")
DEF_KERNEL_D(k_add_f64, "v_add_f64 %0, %0, %1")
DEF_KERNEL_D(k_min_f64, "v_min_f64 %0, %0, %1")
DEF_KERNEL_D(k_rcp_f64, "v_rcp_f64 %0, %0")
DEF_KERNEL_D(k_sqrt_f64, "v_sqrt_f64 %0, %0")
DEF_KERNEL_D(k_rsq_f64, "v_rsq_f64 %0, %0")
DEF_KERNEL_D(k_cmp_f64, "v_cmp_lt_f64 vcc, %0, %1")
DEF_KERNEL_D(k_cmp_f64_sgpr, "v_cmp_lt_f64 s[20:21], %0, %1")
DEF_KERNEL_D(k_div_fixup_f64, "v_div_fixup_f64 %0, %0, %1, %2")
DEF_KERNEL_D(k_div_scale_f64, "v_div_scale_f64 %0, vcc, %0, %1, %2")
DEF_KERNEL_D(k_div_fmas_f64, "v_div_fmas_f64 %0, %0, %1, %2")
DEF_KERNEL_D(k_ldexp_f64, "v_ldexp_f64 %0, %0, 3")
DEF_KERNEL_D(k_mov_b64, "v_mov_b64 %0, %1")
DEF_KERNEL_U(k_mov_b32, "v_mov_b32 %0, %1")
DEF_KERNEL_U(k_add_u32, "v_add_u32 %0, %0, %1")
DEF_KERNEL_U(k_xor3, "v_xor_b32 %0, %0, %1")
DEF_KERNEL_U(k_cndmask, "v_cndmask_b32 %0, %0, %1, vcc")
DEF_KERNEL_U(k_cndmask_sgpr, "v_cndmask_b32_e64 %0, %0, %1, s[20:21]")
DEF_KERNEL_U(k_cndmask_other, "v_cndmask_b32 %0, %1, %2, vcc")
DEF_KERNEL_U(k_cmp_cndmask, "v_cmp_lt_u32 vcc, %0, %1\n v_cndmask_b32 %0, %0, %2, vcc")
DEF_KERNEL_U(k_cmp_cndmask_sgpr, "v_cmp_lt_u32 s[20:21], %0, %1\n v_cndmask_b32_e64 %0, %0, %2, s[20:21]")
DEF_KERNEL_U(k_cmp_u32, "v_cmp_lt_u32 vcc, %0, %1")
DEF_KERNEL_U(k_max_u32, "v_max_u32 %0, %0, %1")
DEF_KERNEL_U(k_mul_lo_u32, "v_mul_lo_u32 %0, %0, %1")
DEF_KERNEL_U(k_mul_hi_u32, "v_mul_hi_u32 %0, %0, %1")
DEF_KERNEL_U(k_fma_f32, "v_fma_f32 %0, %0, %1, %2")
DEF_KERNEL_U(k_sin_f32, "v_sin_f32 %0, %0")

__global__ void k_mad_u64_u32(double* out, int iters, double seed) {
    unsigned long long a[UNROLL];
    for (int k = 0; k < UNROLL; ++k) a[k] = (unsigned long long)seed + k + threadIdx.x;
    unsigned b = (unsigned)(seed * 977);
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int r = 0; r < REPS; ++r) {
#pragma unroll
            for (int k = 0; k < UNROLL; ++k) { asm volatile("v_mad_u64_u32 %0, vcc, %1, %1, %0" : "+v"(a[k]) : "v"(b) : "vcc"); }
        }
    }
    unsigned long long s = 0;
    for (int k = 0; k < UNROLL; ++k) s += a[k];
    if (s == 123456789ull) out[threadIdx.x] = (double)s;
}

// full IEEE division as the compiler emits it, and the quotient-from-reciprocal form
__global__ void k_div_ieee(double* out, int iters, double seed) {
    double a[UNROLL];
    for (int k = 0; k < UNROLL; ++k) a[k] = seed + k + threadIdx.x * 1e-3;
    double b = seed * 0.999;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int r = 0; r < REPS; ++r) {
#pragma unroll
            for (int k = 0; k < UNROLL; ++k) { a[k] = a[k] / b; asm volatile("" : "+v"(a[k])); }
        }
    }
    double s = 0;
    for (int k = 0; k < UNROLL; ++k) s += a[k];
    if (s == 12345.678) out[threadIdx.x] = s;
}

typedef void (*Kern)(double*, int, double);
struct Entry { const char* name; Kern k; };

int main(int argc, char** argv) {
    int dev = 0;
    CHECK(hipSetDevice(dev));
    hipDeviceProp_t p;
    CHECK(hipGetDeviceProperties(&p, dev));
    const int cus = p.multiProcessorCount;
    const double mhz = p.clockRate / 1000.0;
    printf("device %s, %d CUs, clock %.0f MHz\n", p.name, cus, mhz);
    double* out;
    CHECK(hipMalloc(&out, 4096));
    Entry es[] = {
        {"v_mov_b32", k_mov_b32}, {"v_add_u32", k_add_u32}, {"v_xor_b32", k_xor3}, {"v_cndmask_b32", k_cndmask}, {"v_cndmask_b32 sgpr mask", k_cndmask_sgpr}, {"v_cndmask_b32 no chain", k_cndmask_other},
        {"v_cmp_u32+v_cndmask (2)", k_cmp_cndmask}, {"v_cmp_u32+v_cndmask sgpr (2)", k_cmp_cndmask_sgpr}, {"v_cmp_lt_u32", k_cmp_u32}, {"v_max_u32", k_max_u32},
        {"v_fma_f32", k_fma_f32}, {"v_sin_f32", k_sin_f32},
        {"v_mul_lo_u32", k_mul_lo_u32}, {"v_mul_hi_u32", k_mul_hi_u32}, {"v_mad_u64_u32", k_mad_u64_u32},
        {"v_mov_b64", k_mov_b64}, {"v_fma_f64", k_fma_f64}, {"v_mul_f64", k_mul_f64}, {"v_add_f64", k_add_f64}, {"v_min_f64", k_min_f64},
        {"v_cmp_lt_f64 vcc", k_cmp_f64}, {"v_cmp_lt_f64 sgpr", k_cmp_f64_sgpr},
        {"v_rcp_f64", k_rcp_f64}, {"v_sqrt_f64", k_sqrt_f64}, {"v_rsq_f64", k_rsq_f64},
        {"v_div_scale_f64", k_div_scale_f64}, {"v_div_fmas_f64", k_div_fmas_f64}, {"v_div_fixup_f64", k_div_fixup_f64},
        {"a/b (IEEE sequence)", k_div_ieee},
    };
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    const int iters = 2000;
    for (int waves_per_simd : {1, 2, 8}) {
        printf("---- %d wave(s) per SIMD: cycles per wave64 instruction per SIMD (at the reported clock)\n", waves_per_simd);
        for (auto& e : es) {
            dim3 grid(cus * waves_per_simd), block(256); // 4 waves per block = one per SIMD
            hipLaunchKernelGGL(e.k, grid, block, 0, 0, out, 10, 1.5);
            CHECK(hipDeviceSynchronize());
            CHECK(hipEventRecord(e0, 0));
            hipLaunchKernelGGL(e.k, grid, block, 0, 0, out, iters, 1.5);
            CHECK(hipEventRecord(e1, 0));
            CHECK(hipEventSynchronize(e1));
            float ms;
            CHECK(hipEventElapsedTime(&ms, e0, e1));
            const double instr_per_wave = (double)iters * REPS * UNROLL;
            const double cyc = ms * 1e-3 * mhz * 1e6 / (instr_per_wave * waves_per_simd);
            printf("%-24s %8.2f\n", e.name, cyc);
        }
    }
    return 0;
}
