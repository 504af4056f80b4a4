// Issue-overlap microbenchmark for gfx950: how VALU (f64), SALU and branches share a SIMD's issue slots.
// Build: hipcc --offload-arch=gfx950 -O2 scratch/ubench2.hip -o scratch/ubench2
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

#define V "v_fma_f64 %0, %0, %2, %3\n"
#define V2 "v_fma_f64 %1, %1, %2, %3\n"
#define S "s_add_u32 s20, s20, 1\n"
#define S2 "s_and_b64 s[22:23], s[22:23], s[24:25]\n"
#define R4(x) x x x x
#define R16(x) R4(x) R4(x) R4(x) R4(x)

#define DEF(NAME, BODY)                                                                       \
    __global__ void NAME(double* out, int iters, double seed) {                               \
        double a = seed + threadIdx.x, a2 = seed * 2 + threadIdx.x, b = seed * 0.999, c = seed * 1.0001; \
        for (int i = 0; i < iters; ++i) {                                                     \
            asm volatile(BODY : "+v"(a), "+v"(a2) : "v"(b), "v"(c) : "s20", "s22", "s23", "s24", "s25", "scc", "vcc"); \
        }                                                                                     \
        if (a + a2 == 12345.678) out[threadIdx.x] = a;                                        \
    }

DEF(k_v, R16(V V2))                                   // 32 VALU
DEF(k_s, R16(S S2))                                   // 32 SALU
DEF(k_vs, R16(V S V2 S2))                             // 32 VALU + 32 SALU interleaved
DEF(k_vss, R16(V S S2 V2 S S2))                       // 32 VALU + 64 SALU
DEF(k_vsss, R16(V S S2 S V2 S2 S S2))                 // 32 VALU + 96 SALU
// branches: not-taken conditional branch after each pair (scc = 0 after s_cmp_eq of different values)
#define BN "s_cmp_eq_u32 s20, -1\n s_cbranch_scc1 1f\n"
DEF(k_v_bn, R16(V BN V2 BN) "1:\n")                    // 32 VALU + 32 (cmp + branch not taken)
// taken branches: jump over one instruction
#define BT "s_branch 2f\n v_fma_f64 %0, %0, %2, %3\n 2:\n"
#define BT3 "s_branch 3f\n v_fma_f64 %0, %0, %2, %3\n 3:\n"
DEF(k_v_bt, R4(V BT) R4(V2 BT3) R4(V BT) R4(V2 BT3) R4(V BT) R4(V2 BT3) R4(V BT) R4(V2 BT3))   // 32 VALU + 32 taken branches
// scalar load + wait per group of 8 VALU
#define LD "s_load_dwordx16 s[36:51], %4, 0x0\n s_waitcnt lgkmcnt(0)\n"
__global__ void k_v_ld(double* out, int iters, double seed, const double* tab) {
    double a = seed + threadIdx.x, a2 = seed * 2 + threadIdx.x, b = seed * 0.999, c = seed * 1.0001;
    for (int i = 0; i < iters; ++i) {
        asm volatile(R4(LD V V2 V V2 V V2 V V2) : "+v"(a), "+v"(a2) : "v"(b), "v"(c), "s"(tab)
                     : "s36","s37","s38","s39","s40","s41","s42","s43","s44","s45","s46","s47","s48","s49","s50","s51");
    }
    if (a + a2 == 12345.678) out[threadIdx.x] = a;
}
__global__ void k_ld(double* out, int iters, double seed, const double* tab) {
    double a = seed + threadIdx.x;
    for (int i = 0; i < iters; ++i) {
        asm volatile(R4(LD LD LD LD) : "+v"(a) : "v"(a), "v"(a), "v"(a), "s"(tab)
                     : "s36","s37","s38","s39","s40","s41","s42","s43","s44","s45","s46","s47","s48","s49","s50","s51");
    }
    if (a == 12345.678) out[threadIdx.x] = a;
}

typedef void (*Kern)(double*, int, double);
struct Entry { const char* name; Kern k; int valu, other; };

int main() {
    CHECK(hipSetDevice(0));
    hipDeviceProp_t p;
    CHECK(hipGetDeviceProperties(&p, 0));
    const int cus = p.multiProcessorCount;
    const double mhz = p.clockRate / 1000.0;
    double* out;
    CHECK(hipMalloc(&out, 65536));
    CHECK(hipMemset(out, 0, 65536));
    Entry es[] = {
        {"32 VALU(f64)", k_v, 32, 0}, {"32 SALU", k_s, 0, 32}, {"32 VALU + 32 SALU", k_vs, 32, 32}, {"32 VALU + 64 SALU", k_vss, 32, 64},
        {"32 VALU + 96 SALU", k_vsss, 32, 96}, {"32 VALU + 32 (s_cmp + branch not taken)", k_v_bn, 32, 64},
        {"32 VALU + 32 taken s_branch", k_v_bt, 32, 32},
    };
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    const int iters = 4000;
    for (int w : {1, 2, 4, 8}) {
        printf("---- %d wave(s) per SIMD: cycles per loop iteration per SIMD-wave (i.e. time * clock / (iters * waves))\n", w);
        dim3 grid(cus * w), block(256);
        for (auto& e : es) {
            hipLaunchKernelGGL(e.k, grid, block, 0, 0, out, 10, 1.5);
            CHECK(hipDeviceSynchronize());
            CHECK(hipEventRecord(e0, 0));
            hipLaunchKernelGGL(e.k, grid, block, 0, 0, out, iters, 1.5);
            CHECK(hipEventRecord(e1, 0));
            CHECK(hipEventSynchronize(e1));
            float ms;
            CHECK(hipEventElapsedTime(&ms, e0, e1));
            printf("%-44s %8.1f\n", e.name, ms * 1e-3 * mhz * 1e6 / ((double)iters * w));
        }
        for (int which = 0; which < 2; ++which) {
            auto k = which ? k_ld : k_v_ld;
            hipLaunchKernelGGL(k, grid, block, 0, 0, out, 10, 1.5, (const double*)out);
            CHECK(hipDeviceSynchronize());
            CHECK(hipEventRecord(e0, 0));
            hipLaunchKernelGGL(k, grid, block, 0, 0, out, iters, 1.5, (const double*)out);
            CHECK(hipEventRecord(e1, 0));
            CHECK(hipEventSynchronize(e1));
            float ms;
            CHECK(hipEventElapsedTime(&ms, e0, e1));
            printf("%-44s %8.1f\n", which ? "16 x (s_load_dwordx16 + wait)" : "4 x (s_load_dwordx16 + wait + 8 VALU)", ms * 1e-3 * mhz * 1e6 / ((double)iters * w));
        }
    }
    return 0;
}
