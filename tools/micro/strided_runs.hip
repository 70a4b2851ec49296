// Micro-benchmark: what a transform pass's access pattern reaches on this GPU as a function of its row-segment length.
// A pass with radix R over L = R * ncol complex points reads, per tile of T columns, R row segments of T consecutive 16-byte
// elements (segment stride: ncol elements) and writes the tile back as one contiguous run of T * R elements (first-pass order).
// Question behind it: a TWO-pass transform of the 10-minute capture (L = 3 583 125 = 1875 x 1911) would move a third less than
// the three passes per direction it uses now, but only with T = 4 columns per tile (1911 x 4 x 16 B = 122 KB of LDS): 64-byte
// segments.  Do they hold up when the array (57 MB) lives in the Infinity Cache?
// Build: hipcc --offload-arch=gfx950 -O3 -o strided_runs tools/micro/strided_runs.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("hip error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

__global__ void __launch_bounds__(256) pass_like(const double2 *__restrict__ in, double2 *__restrict__ out, int R, int T, long long ncol, int ntiles)
{
    extern __shared__ double2 tile[];
    const int t = threadIdx.x, items = R * T;
    for (int tix = blockIdx.x; tix < ntiles; tix += gridDim.x) {
        const long long j0 = (long long)tix * T;
        const int tn = (int)(ncol - j0 < T ? ncol - j0 : T);
        __syncthreads();
        for (int it = t; it < items; it += 256) {
            const int r = it / T, c = it - r * T;
            if (c < tn) tile[c * R + r] = in[j0 + c + (long long)r * ncol];       // transposed into output order
        }
        __syncthreads();
        const long long o0 = j0 * R;
        for (int e = t; e < R * tn; e += 256) out[o0 + e] = tile[e];
    }
}

int main()
{
    const long long L = 3583125;
    double2 *a, *b;
    CK(hipMalloc(&a, L * 16 + 256));
    CK(hipMalloc(&b, L * 16 + 256));
    CK(hipMemset(a, 0, L * 16));
    CK(hipMemset(b, 0, L * 16));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    const int cfg[][2] = {{225, 16}, {91, 32}, {175, 16}, {1911, 4}, {1875, 4}, {1911, 2}, {637, 8}, {637, 4}, {3185, 2}, {2275, 4}};
    for (auto &c : cfg) {
        const int R = c[0], T = c[1];
        if (L % R) continue;
        const long long ncol = L / R;
        const int ntiles = (int)((ncol + T - 1) / T);
        const size_t lds = (size_t)R * T * 16;
        if (lds > 160 * 1024) continue;
        CK(hipFuncSetAttribute((const void *)pass_like, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        const int per_cu = lds > 80 * 1024 ? 1 : lds > 53 * 1024 ? 2 : 3;
        const int grid = ntiles < 256 * per_cu ? ntiles : 256 * per_cu;
        for (int w = 0; w < 3; ++w) {
            hipLaunchKernelGGL(pass_like, dim3(grid), dim3(256), lds, 0, a, b, R, T, ncol, ntiles);
            hipLaunchKernelGGL(pass_like, dim3(grid), dim3(256), lds, 0, b, a, R, T, ncol, ntiles);
        }
        CK(hipEventRecord(e0));
        const int reps = 10;
        for (int w = 0; w < reps; ++w) {
            hipLaunchKernelGGL(pass_like, dim3(grid), dim3(256), lds, 0, a, b, R, T, ncol, ntiles);
            hipLaunchKernelGGL(pass_like, dim3(grid), dim3(256), lds, 0, b, a, R, T, ncol, ntiles);
        }
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms = 0;
        CK(hipEventElapsedTime(&ms, e0, e1));
        const double us = 1e3 * ms / (2 * reps);
        printf("R %5d  T %3d  segment %4d B  LDS %6zu B  grid %4d : %7.1f us per pass = %5.2f TB/s (read + write %.0f MB)\n", R, T, T * 16, lds, grid, us,
               2.0 * L * 16 / us / 1e6, 2.0 * L * 16 / 1e6);
    }
    return 0;
}
