// a6 (wefax.py:63-72): what the notch kernels of wfx_stages.hip and the fused notch + P2M kernel of wfx_fmm.hip share -- the 49-tap form's
// coefficients, filtfilt's odd extension in the capture's own dtype, the biquad recurrence of the exact edges, and the host's preparation.
#pragma once
#include <cmath>
#include <cstdint>

#define NOTCH_K 24
#define NOTCH_EDGE 64
#define NOTCH_SETTLE 63
#define NOTCH_PAD 9
#define NOTCH_SMALL (2 * (NOTCH_EDGE + NOTCH_SETTLE))

struct notch_coef {
    double g[NOTCH_K + 1];
    double b[3], a[3], zi[2];
    int has_ext;                      // the odd extension is given (evaluated by the host in the capture's own dtype)
    double extl[NOTCH_PAD], extr[NOTCH_PAD];
};

template <typename TIN>
__device__ __forceinline__ double notch_ext_left(const TIN *x, int k);   // 2*x[0] - x[k]
template <>
__device__ __forceinline__ double notch_ext_left<short>(const short *x, int k)
{
    return (double)(short)(2 * (int)x[0] - (int)x[k]);     // int16 wrap, as numpy does for an int16 array
}
template <>
__device__ __forceinline__ double notch_ext_left<double>(const double *x, int k)
{
    return 2 * x[0] - x[k];
}
template <typename TIN>
__device__ __forceinline__ double notch_ext_right(const TIN *x, uint64_t n, int k);   // 2*x[n-1] - x[n-1-k]
template <>
__device__ __forceinline__ double notch_ext_right<short>(const short *x, uint64_t n, int k)
{
    return (double)(short)(2 * (int)x[n - 1] - (int)x[n - 1 - k]);
}
template <>
__device__ __forceinline__ double notch_ext_right<double>(const double *x, uint64_t n, int k)
{
    return 2 * x[n - 1] - x[n - 1 - k];
}

// extended sample k places before x[0] / after x[n-1] (k = 1..9)
template <typename TIN>
__device__ __forceinline__ double notch_left(const notch_coef &c, const TIN *x, int k)
{
    return c.has_ext ? c.extl[NOTCH_PAD - k] : notch_ext_left<TIN>(x, k);
}
template <typename TIN>
__device__ __forceinline__ double notch_right(const notch_coef &c, const TIN *x, uint64_t n, int k)
{
    return c.has_ext ? c.extr[k - 1] : notch_ext_right<TIN>(x, n, k);
}

// transposed direct form II step, the recurrence of scipy's lfilter
__device__ __forceinline__ double biquad_step(const notch_coef &c, double xi, double &z0, double &z1)
{
    const double yi = z0 + c.b[0] * xi;
    z0 = z1 + c.b[1] * xi - c.a[1] * yi;
    z1 = c.b[2] * xi - c.a[2] * yi;
    return yi;
}


static inline void notch_prepare(notch_coef &c, const double b[3], const double a[3], const double *ext18 = nullptr)
{
    c.has_ext = ext18 != nullptr;
    for (int i = 0; i < NOTCH_PAD; ++i) {
        c.extl[i] = ext18 ? ext18[i] : 0.0;
        c.extr[i] = ext18 ? ext18[NOTCH_PAD + i] : 0.0;
    }
    for (int i = 0; i < 3; ++i) {
        c.b[i] = b[i] / a[0];
        c.a[i] = a[i] / a[0];
    }
    // lfilter_zi: solve (I - companion(a).T) zi = b[1:] - a[1:] b[0]
    {
        const double m00 = 1.0 + c.a[1], m01 = -1.0, m10 = c.a[2], m11 = 1.0;
        const double r0 = c.b[1] - c.a[1] * c.b[0], r1 = c.b[2] - c.a[2] * c.b[0];
        const double det = m00 * m11 - m01 * m10;
        c.zi[0] = (r0 * m11 - m01 * r1) / det;
        c.zi[1] = (m00 * r1 - m10 * r0) / det;
    }
    // impulse response and its autocorrelation
    {
        double imp[160];
        double z0 = 0.0, z1 = 0.0;
        for (int i = 0; i < 160; ++i) {
            const double xi = i == 0 ? 1.0 : 0.0;
            const double yi = z0 + c.b[0] * xi;
            z0 = z1 + c.b[1] * xi - c.a[1] * yi;
            z1 = c.b[2] * xi - c.a[2] * yi;
            imp[i] = yi;
        }
        for (int k = 0; k <= NOTCH_K; ++k) {
            double s = 0.0;
            for (int i = 0; i + k < 160; ++i) s += imp[i] * imp[i + k];
            c.g[k] = s;
        }
    }
}


static inline double biquad_pole_radius(const double a[3])
{
    const double a1 = a[1] / a[0], a2 = a[2] / a[0];
    const double disc = a1 * a1 - 4.0 * a2;
    if (disc < 0.0) return sqrt(a2);
    const double s = sqrt(disc);
    return fmax(fabs((-a1 + s) / 2.0), fabs((-a1 - s) / 2.0));
}

