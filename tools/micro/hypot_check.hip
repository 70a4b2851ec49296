// Is hypot(x, y) of the device library bit-identical to sqrt(fma(x, x, y*y)) for the value ranges of the
// envelope kernel (|x|, |y| < 1e6, down to exact zeros)?  Counts mismatches of both operand orders.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cmath>
#include <vector>
#include <random>
__global__ void check(const double *x, const double *y, int n, unsigned *bad)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const double h = hypot(x[i], y[i]);
    const double a = sqrt(__builtin_fma(x[i], x[i], y[i] * y[i]));
    const double b = sqrt(__builtin_fma(y[i], y[i], x[i] * x[i]));
    if (a != h) atomicAdd(&bad[0], 1u);
    if (b != h) atomicAdd(&bad[1], 1u);
}
int main()
{
    const int n = 1 << 24;
    std::vector<double> x(n), y(n);
    std::mt19937_64 g(1);
    std::normal_distribution<double> nd(0.0, 8000.0);
    std::uniform_real_distribution<double> ud(-1.0, 1.0);
    for (int i = 0; i < n; ++i) {
        x[i] = (i & 7) == 0 ? std::rint(nd(g)) : nd(g);
        y[i] = (i & 15) == 1 ? 0.0 : (i & 15) == 2 ? ud(g) * 1e-9 : nd(g);
        if ((i & 1023) == 5) x[i] = 0.0;
    }
    double *dx, *dy;
    unsigned *bad, hb[2];
    hipMalloc(&dx, n * 8); hipMalloc(&dy, n * 8); hipMalloc(&bad, 8);
    hipMemcpy(dx, x.data(), n * 8, hipMemcpyHostToDevice);
    hipMemcpy(dy, y.data(), n * 8, hipMemcpyHostToDevice);
    hipMemset(bad, 0, 8);
    check<<<n / 256, 256>>>(dx, dy, n, bad);
    hipMemcpy(hb, bad, 8, hipMemcpyDeviceToHost);
    // and against the host libm
    unsigned host_bad = 0;
    for (int i = 0; i < n; ++i) host_bad += std::sqrt(std::fma(x[i], x[i], y[i] * y[i])) != std::hypot(x[i], y[i]);
    printf("n=%d  device hypot != sqrt(fma(x,x,y*y)): %u   != sqrt(fma(y,y,x*x)): %u   host libm hypot != sqrt(fma(x,x,y*y)): %u\n", n, hb[0], hb[1], host_bad);
    return 0;
}
