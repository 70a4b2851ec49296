// Micro-benchmark: issue rate of the integer instructions the ingest kernel is made of (v_dot2_i32_i16 with a scalar operand,
// v_perm_b32, v_pk_add_u16, v_add_u32) on one MI355X, all CUs, 1 / 2 / 3 / 4 waves per SIMD.
// Build: hipcc --offload-arch=gfx950 -O3 -o valu_rate tools/micro/valu_rate.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("hip error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
typedef short s2 __attribute__((ext_vector_type(2)));
typedef unsigned short us2 __attribute__((ext_vector_type(2)));

template <int OP>
__global__ void __launch_bounds__(256) rate_kernel(int *out, int iters, const int *__restrict__ taps)
{
    int r[8], w[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) { r[k] = threadIdx.x + k; w[k] = threadIdx.x * 77 + k; }
    int c[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) c[k] = taps[k];                 // uniform: scalar registers
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int rep = 0; rep < 8; ++rep)
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                if (OP == 0) r[k] = __builtin_amdgcn_sdot2(__builtin_bit_cast(s2, w[(k + rep) & 7]), __builtin_bit_cast(s2, c[k]), r[k], false);      // 8 chains
                if (OP == 1) r[k % 6] = __builtin_amdgcn_sdot2(__builtin_bit_cast(s2, w[(k + rep) & 7]), __builtin_bit_cast(s2, c[k]), r[k % 6], false);   // 6 chains (the kernel)
                if (OP == 2) r[k] = __builtin_amdgcn_perm(r[k], w[k], 0x05040100u);
                if (OP == 3) r[k] = __builtin_bit_cast(int, (us2)(__builtin_bit_cast(us2, r[k]) + __builtin_bit_cast(us2, w[k])));
                if (OP == 4) r[k] = r[k] + w[k];
                if (OP == 5) r[k] = __builtin_amdgcn_sdot2(__builtin_bit_cast(s2, w[(k + rep) & 7]), __builtin_bit_cast(s2, w[(k + 3) & 7]), r[k], false);   // all-VGPR operands
                if (OP == 6) r[k] = __builtin_amdgcn_sdot4(w[(k + rep) & 7], c[k], r[k], false);
            }
    }
    int s = 0;
#pragma unroll
    for (int k = 0; k < 8; ++k) s += r[k];
    if (s == 0x12345678) out[0] = s;
}

int main()
{
    int *out, *taps;
    CK(hipMalloc(&out, 64));
    CK(hipMalloc(&taps, 64));
    int h[8] = {0x00010002, 0x00030004, 0x7fff8000, 0x00050006, 0x00070008, 0x0009000a, 0x000b000c, 0x000d000e};
    CK(hipMemcpy(taps, h, 32, hipMemcpyHostToDevice));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    const int iters = 2000;
    const char *names[] = {"v_dot2_i32_i16 s,v 8ch", "v_dot2_i32_i16 s,v 6ch", "v_perm_b32", "v_pk_add_u16", "v_add_u32", "v_dot2_i32_i16 v,v", "v_dot4_i32_i8 s,v"};
    for (int op = 0; op < 7; ++op)
        for (int blocks_per_cu : {1, 2, 3, 4}) {
            const int grid = 256 * blocks_per_cu;
            float best = 1e9f;
            for (int r = 0; r < 4; ++r) {
                CK(hipEventRecord(e0, 0));
                if (op == 0) rate_kernel<0><<<grid, 256>>>(out, iters, taps);
                if (op == 1) rate_kernel<1><<<grid, 256>>>(out, iters, taps);
                if (op == 2) rate_kernel<2><<<grid, 256>>>(out, iters, taps);
                if (op == 3) rate_kernel<3><<<grid, 256>>>(out, iters, taps);
                if (op == 4) rate_kernel<4><<<grid, 256>>>(out, iters, taps);
                if (op == 5) rate_kernel<5><<<grid, 256>>>(out, iters, taps);
                if (op == 6) rate_kernel<6><<<grid, 256>>>(out, iters, taps);
                CK(hipEventRecord(e1, 0));
                CK(hipEventSynchronize(e1));
                float ms;
                CK(hipEventElapsedTime(&ms, e0, e1));
                if (r) best = ms < best ? ms : best;
            }
            // wave-instructions per SIMD = iters * 64 * blocks_per_cu (4 waves of a block land on the 4 SIMDs)
            const double winst = (double)iters * 64.0 * blocks_per_cu;
            printf("%-24s %d wave(s)/SIMD: %8.3f ms  %.2f cycles per wave-instruction per SIMD at 2.4 GHz\n", names[op], blocks_per_cu, best, best * 1e-3 * 2.4e9 / winst);
        }
    return 0;
}
