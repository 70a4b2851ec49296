// Micro-benchmark: issue rate of f64 FMA / MUL / ADD and f32 FMA on one MI355X (all CUs, W waves per SIMD).
// Build: hipcc --offload-arch=gfx950 -O3 -o f64_rate tools/micro/f64_rate.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("hip error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

template <int OP>
__global__ void __launch_bounds__(256) rate_kernel(double *out, int iters, double a, double b)
{
    double r[8];
    float f[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) { r[k] = threadIdx.x + k; f[k] = threadIdx.x + k; }
    const float af = (float)a, bf = (float)b;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int rep = 0; rep < 8; ++rep)
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                if (OP == 0) r[k] = __builtin_fma(r[k], a, b);
                if (OP == 1) r[k] = r[k] * a;
                if (OP == 2) r[k] = r[k] + b;
                if (OP == 3) f[k] = __builtin_fmaf(f[k], af, bf);
                if (OP == 4) r[k] = __builtin_fma(a, r[(k + 1) & 7], r[k]);       // v_fmac_f64 acc, sgpr, vgpr (the FIR form)
                if (OP == 5) r[k & 3] = __builtin_fma(a, r[4 + (k & 3)], r[k & 3]);   // 4 accumulators only
            }
    }
    double s = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) s += r[k] + f[k];
    if (s == 1.234567e-300) out[0] = s;
}

int main()
{
    double *out;
    CK(hipMalloc(&out, 64));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    const int iters = 2000;
    const char *names[] = {"v_fma_f64", "v_mul_f64", "v_add_f64", "v_fma_f32", "fmac s,v 8", "fmac s,v 4"};
    for (int op = 0; op < 6; ++op)
        for (int blocks_per_cu : {1, 2, 4}) {
            const int grid = 256 * blocks_per_cu;
            float best = 1e9f;
            for (int r = 0; r < 5; ++r) {
                CK(hipEventRecord(e0, 0));
                if (op == 0) rate_kernel<0><<<grid, 256>>>(out, iters, 1.0000001, 1e-9);
                if (op == 1) rate_kernel<1><<<grid, 256>>>(out, iters, 1.0000001, 1e-9);
                if (op == 2) rate_kernel<2><<<grid, 256>>>(out, iters, 1.0000001, 1e-9);
                if (op == 3) rate_kernel<3><<<grid, 256>>>(out, iters, 1.0000001, 1e-9);
                if (op == 4) rate_kernel<4><<<grid, 256>>>(out, iters, 1.0000001e-3, 1e-9);
                if (op == 5) rate_kernel<5><<<grid, 256>>>(out, iters, 1.0000001e-3, 1e-9);
                CK(hipEventRecord(e1, 0));
                CK(hipEventSynchronize(e1));
                float ms;
                CK(hipEventElapsedTime(&ms, e0, e1));
                if (r) best = ms < best ? ms : best;
            }
            const double insts = (double)grid * 4 /*waves*/ * iters * 64.0;   // wave-instructions
            const double lane_ops = insts * 64;
            printf("%-10s %d wave(s)/SIMD: %7.3f ms  %.2f T lane-ops/s  (%.1f cycles per wave-instruction per SIMD at 2.4 GHz)\n", names[op], blocks_per_cu,
                   best, lane_ops / (best * 1e-3) / 1e12, (best * 1e-3) * 2.4e9 / (insts / (256.0 * 4)));
        }
    return 0;
}
