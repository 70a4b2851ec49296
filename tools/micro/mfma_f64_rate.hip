// Micro-benchmark (round 6): issue rate of v_mfma_f64_16x16x4_f64 on one MI355X, alone and beside f64 VALU FMAs --
//   mode 0: every wave MFMA only (NACC independent accumulators)
//   mode 1: every wave VALU v_fma_f64 only
//   mode 2: odd waves of a workgroup MFMA, even waves VALU (do the two pipes overlap for f64?)
//   mode 3: ONE stream per wave: 1 MFMA + NV VALU FMAs interleaved
// Build: hipcc --offload-arch=gfx950 -O3 -o mfma_f64_rate tools/micro/mfma_f64_rate.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("hip error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

typedef double d4 __attribute__((ext_vector_type(4)));

template <int MODE, int NV>
__global__ void __launch_bounds__(512) k(double *out, int iters, double a, double b)
{
    d4 acc[4];
    double r[8];
#pragma unroll
    for (int q = 0; q < 4; ++q) acc[q] = d4{0, 0, 0, 0};
#pragma unroll
    for (int q = 0; q < 8; ++q) r[q] = threadIdx.x + q;
    const int wave = threadIdx.x >> 6;
    const bool do_m = MODE == 0 || MODE == 3 || (MODE == 2 && (wave & 1));
    const bool do_v = MODE == 1 || MODE == 3 || (MODE == 2 && !(wave & 1));
    double av = a + threadIdx.x * 1e-9, bv = b;
    for (int it = 0; it < iters; ++it) {
        if (MODE == 3) {
#pragma unroll
            for (int rep = 0; rep < 8; ++rep) {
                acc[rep & 3] = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, acc[rep & 3], 0, 0, 0);
#pragma unroll
                for (int q = 0; q < NV; ++q) r[q & 7] = __builtin_fma(a, r[(q + 1) & 7], r[q & 7]);
            }
        } else {
            if (do_m) {
#pragma unroll
                for (int rep = 0; rep < 8; ++rep) acc[rep & 3] = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, acc[rep & 3], 0, 0, 0);
            }
            if (do_v) {
#pragma unroll
                for (int rep = 0; rep < 8 * NV; ++rep) r[rep & 7] = __builtin_fma(a, r[(rep + 1) & 7], r[rep & 7]);
            }
        }
    }
    double s = 0;
#pragma unroll
    for (int q = 0; q < 4; ++q) s += acc[q][0] + acc[q][1] + acc[q][2] + acc[q][3];
#pragma unroll
    for (int q = 0; q < 8; ++q) s += r[q];
    if (s == 1.234567e-300) out[0] = s;
}

template <int MODE, int NV>
static int run(const char *name, double *out, int wpb)
{
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    const int iters = 1000;
    for (int bpc : {1, 2}) {
        const int grid = 256 * bpc;
        float best = 1e9f;
        for (int rr = 0; rr < 5; ++rr) {
            CK(hipEventRecord(e0, 0));
            k<MODE, NV><<<grid, wpb * 64>>>(out, iters, 1.0000001e-3, 1e-9);
            CK(hipEventRecord(e1, 0));
            CK(hipEventSynchronize(e1));
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            if (rr) best = ms < best ? ms : best;
        }
        const double waves = (double)grid * wpb;
        double n_m = 0, n_v = 0;
        if (MODE == 0) n_m = waves * iters * 8.0;
        if (MODE == 1) n_v = waves * iters * 8.0 * NV;
        if (MODE == 2) { n_m = waves / 2 * iters * 8.0; n_v = waves / 2 * iters * 8.0 * NV; }
        if (MODE == 3) { n_m = waves * iters * 8.0; n_v = waves * iters * 8.0 * NV; }
        const double fma = n_m * 1024 + n_v * 64;
        printf("%-28s %d wg/CU x %d waves: %7.3f ms  mfma %.3g valu %.3g  -> %.2f T FMA/s (%.1f TFLOP/s); per SIMD: %.1f cyc/mfma-equivalent at 2.4 GHz\n", name, bpc, wpb, best,
               n_m, n_v, fma / (best * 1e-3) / 1e12, 2 * fma / (best * 1e-3) / 1e12, (best * 1e-3) * 2.4e9 / ((fma / 1024) / 1024.0));
    }
    return 0;
}

int main()
{
    double *out;
    CK(hipMalloc(&out, 64));
    run<0, 0>("mfma only", out, 4);
    run<0, 0>("mfma only", out, 8);
    run<1, 16>("valu only", out, 4);
    run<1, 16>("valu only", out, 8);
    run<2, 16>("odd mfma / even valu(16:1)", out, 8);
    run<2, 8>("odd mfma / even valu(8:1)", out, 8);
    run<3, 4>("one stream 1 mfma + 4 valu", out, 4);
    run<3, 8>("one stream 1 mfma + 8 valu", out, 4);
    run<3, 16>("one stream 1 mfma + 16 valu", out, 4);
    run<3, 8>("one stream 1 mfma + 8 valu", out, 8);
    run<3, 16>("one stream 1 mfma + 16 valu", out, 8);
    return 0;
}
