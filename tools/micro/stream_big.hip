// Micro-benchmark: the read-only streaming ceiling of this GPU on an array far beyond the Infinity Cache (16 GiB), as the
// front end's ingest kernel sees it: 16-byte loads, UN of them in flight per lane, workgroups walking contiguous tiles.
// Build: hipcc --offload-arch=gfx950 -O3 -o stream_big tools/micro/stream_big.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("hip error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

// tile = UN * 4 KiB per workgroup and step (256 lanes x 16 B x UN), tiles dealt round-robin to the grid
template <int UN>
__global__ void __launch_bounds__(256) read_tiles(const uint4 *__restrict__ v, uint64_t ntiles, unsigned *__restrict__ out)
{
    const int t = threadIdx.x;
    unsigned acc = 0;
    for (uint64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const uint4 *p = v + tile * (uint64_t)(UN * 256) + t;
        uint4 a[UN];
#pragma unroll
        for (int u = 0; u < UN; ++u) a[u] = p[u * 256];
#pragma unroll
        for (int u = 0; u < UN; ++u) acc += a[u].x ^ a[u].y ^ a[u].z ^ a[u].w;
    }
    if (acc == 0x12345678u) out[0] = acc;
}

template <int UN>
static int run(const uint4 *v, uint64_t bytes, unsigned *out, int grid)
{
    const uint64_t ntiles = bytes / (UN * 4096ull);
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    hipLaunchKernelGGL(read_tiles<UN>, dim3(grid), dim3(256), 0, 0, v, ntiles, out);
    CK(hipEventRecord(e0));
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(read_tiles<UN>, dim3(grid), dim3(256), 0, 0, v, ntiles, out);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    printf("UN %2d (tile %3d KiB) grid %5d: %.3f ms  %.2f TB/s\n", UN, UN * 4, grid, ms / 3, bytes / (ms / 3 * 1e-3) / 1e12);
    return 0;
}

int main()
{
    const uint64_t bytes = 16ull << 30;
    uint4 *v;
    unsigned *out;
    CK(hipMalloc(&v, bytes));
    CK(hipMalloc(&out, 64));
    CK(hipMemset(v, 1, bytes));
    CK(hipDeviceSynchronize());
    for (int grid : {1024, 2048, 4096, 8192}) {
        if (run<4>(v, bytes, out, grid)) return 1;
        if (run<9>(v, bytes, out, grid)) return 1;
        if (run<16>(v, bytes, out, grid)) return 1;
    }
    return 0;
}
