// Stencil / elementwise / order-statistic / sync / image kernels of the WEFAX
// hot path.  Each block cites the reference lines it replaces.  All of these are
// HBM-bound byte / double streaming kernels (no MFMA): 16-byte accesses where the
// data allows, LDS tiles for the stencils, wave64 ballots and shuffles for the
// reductions, one global atomic per workgroup.
#include "wfx_internal.h"

// ===========================================================================
// a4  stereo merge (wefax.py:360-373): np.add on np.int16 scalars wraps, /2 -> f64
// ===========================================================================
__global__ void __launch_bounds__(256) merge_kernel(const short2 *__restrict__ lr, uint64_t n, double *__restrict__ out)
{
    for (uint64_t i = blockIdx.x * 256ull + threadIdx.x; i < n; i += (uint64_t)gridDim.x * 256ull) {
        const short2 v = lr[i];
        const short s = (short)((int)v.x + (int)v.y);   // wraps modulo 2^16
        out[i] = (double)s / 2.0;
    }
}

int wfx_dev_merge(wfx_ctx *ctx, const int16_t *lr, uint64_t n, double *out)
{
    WFX_LAUNCH(ctx, K_MERGE, merge_kernel, dim3(wfx_stream_grid(n, 256)), dim3(256), (const short2 *)lr, n, out);
    return 0;
}

__global__ void __launch_bounds__(256) i16_to_f64_kernel(const short *__restrict__ in, uint64_t n, double *__restrict__ out)
{
    for (uint64_t i = blockIdx.x * 256ull + threadIdx.x; i < n; i += (uint64_t)gridDim.x * 256ull) out[i] = (double)in[i];
}

int wfx_dev_i16_to_f64(wfx_ctx *ctx, const int16_t *in, uint64_t n, double *out)
{
    WFX_LAUNCH(ctx, K_MERGE, i16_to_f64_kernel, dim3(wfx_stream_grid(n, 256)), dim3(256), (const short *)in, n, out);
    return 0;
}

// ===========================================================================
// a6  notch / slope filter (wefax.py:68-72): scipy.signal.filtfilt(b, a, x)
//
// Interior: forward-backward filtering with a biquad whose poles have radius
// 0.21 equals convolution with g = autocorrelation of the impulse response; |g[k]|
// falls below 1e-16 of g[0] for |k| > 24, so a 49-tap symmetric FIR reproduces it to
// double rounding.  Each thread produces 4 consecutive outputs from a 52-sample
// register window (13 LDS reads per output instead of 49); the LDS tile is padded by
// one double per four so that the stride-4 accesses of a wave are conflict-free.  The first/last NOTCH_EDGE samples depend on filtfilt's odd extension
// (9 samples) and lfilter_zi initial state, and are computed with the actual
// recurrence by two single threads of one extra workgroup.
// ===========================================================================
#define NOTCH_K 24
#define NOTCH_EDGE 64
#define NOTCH_SETTLE 63
#define NOTCH_PAD 9
#define NOTCH_SMALL (2 * (NOTCH_EDGE + NOTCH_SETTLE))

struct notch_coef {
    double g[NOTCH_K + 1];
    double b[3], a[3], zi[2];
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

// transposed direct form II step, the recurrence of scipy's lfilter
__device__ __forceinline__ double biquad_step(const notch_coef &c, double xi, double &z0, double &z1)
{
    const double yi = z0 + c.b[0] * xi;
    z0 = z1 + c.b[1] * xi - c.a[1] * yi;
    z1 = c.b[2] * xi - c.a[2] * yi;
    return yi;
}

template <typename TIN>
__global__ void __launch_bounds__(256) notch_kernel(const TIN *__restrict__ x, uint64_t n, notch_coef c, double *__restrict__ y, unsigned interior_blocks)
{
    constexpr int TLEN = 1024 + 2 * NOTCH_K;
    __shared__ double tile[TLEN + TLEN / 4 + 4];
    __shared__ double ebuf[2][NOTCH_SMALL + 2 * NOTCH_PAD + 8];
    const int t = threadIdx.x;
    if (blockIdx.x < interior_blocks) {
        // interior outputs [lo, hi) = [EDGE, n - EDGE)
        const uint64_t lo = NOTCH_EDGE, hi = n - NOTCH_EDGE;
        for (uint64_t base = lo + (uint64_t)blockIdx.x * 1024ull; base < hi; base += (uint64_t)interior_blocks * 1024ull) {
            __syncthreads();
            for (int i = t; i < TLEN; i += 256) {
                const uint64_t src = base - NOTCH_K + i;       // >= EDGE - K >= 0
                tile[i + (i >> 2)] = src < n ? (double)x[src] : 0.0;
            }
            __syncthreads();
            double win[2 * NOTCH_K + 4];
#pragma unroll
            for (int i = 0; i < 2 * NOTCH_K + 4; ++i) win[i] = tile[(4 * t + i) + ((4 * t + i) >> 2)];
            double acc[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                acc[u] = c.g[0] * win[u + NOTCH_K];
#pragma unroll
                for (int k = 1; k <= NOTCH_K; ++k) acc[u] = fma(c.g[k], win[u + NOTCH_K - k] + win[u + NOTCH_K + k], acc[u]);
            }
            // a lane's 4 outputs are contiguous: exchange through LDS so that each store
            // instruction writes 64 x 16 contiguous bytes instead of 64 quarter lines
            __syncthreads();
#pragma unroll
            for (int u = 0; u < 4; ++u) tile[5 * t + u] = acc[u];
            __syncthreads();
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                const int j = half * 512 + 2 * t;                 // even -> same padded group of four
                const uint64_t o = base + j;                      // base is even: 16-byte aligned
                const double v0 = tile[j + (j >> 2)], v1 = tile[j + 1 + ((j + 1) >> 2)];
                if (o + 1 < hi)
                    *(double2 *)(y + o) = make_double2(v0, v1);
                else if (o < hi)
                    y[o] = v0;
            }
        }
        return;
    }
    // ---- edge workgroup -------------------------------------------------------
    if (n < NOTCH_SMALL) {
        // whole signal with the exact recurrence (one thread)
        if (t == 0) {
            double *e = ebuf[0];
            const int len = (int)n + 2 * NOTCH_PAD;
            for (int i = 0; i < NOTCH_PAD; ++i) e[i] = notch_ext_left<TIN>(x, NOTCH_PAD - i);
            for (int i = 0; i < (int)n; ++i) e[NOTCH_PAD + i] = (double)x[i];
            for (int i = 0; i < NOTCH_PAD; ++i) e[NOTCH_PAD + n + i] = notch_ext_right<TIN>(x, n, i + 1);
            double z0 = c.zi[0] * e[0], z1 = c.zi[1] * e[0];
            for (int i = 0; i < len; ++i) e[i] = biquad_step(c, e[i], z0, z1);
            z0 = c.zi[0] * e[len - 1];
            z1 = c.zi[1] * e[len - 1];
            for (int i = len - 1; i >= 0; --i) e[i] = biquad_step(c, e[i], z0, z1);
            for (int i = 0; i < (int)n; ++i) y[i] = e[NOTCH_PAD + i];
        }
        return;
    }
    // Both edges: the extended samples are staged in LDS by the whole workgroup, then one
    // thread per edge runs the two recurrences eight samples at a time (loads and stores
    // are kept off the dependent chain), then the workgroup writes the results.
    constexpr int L = NOTCH_EDGE + NOTCH_SETTLE;        // 127
    constexpr int LEN = NOTCH_PAD + L;                  // 136, a multiple of 8
    static_assert(LEN % 8 == 0, "edge length must be a multiple of 8");
    for (int i = t; i < LEN; i += 256) {
        ebuf[0][i] = i < NOTCH_PAD ? notch_ext_left<TIN>(x, NOTCH_PAD - i) : (double)x[i - NOTCH_PAD];
        ebuf[1][i] = i < L ? (double)x[n - L + i] : notch_ext_right<TIN>(x, n, i - L + 1);
    }
    __syncthreads();
    if (t == 0 || t == 64) {
        double *e = ebuf[t == 0 ? 0 : 1];
        // left: exact forward start, backward started SETTLE samples to the right with a zero state
        // right: forward started SETTLE samples early with a zero state, exact backward start
        double z0 = t == 0 ? c.zi[0] * e[0] : 0.0, z1 = t == 0 ? c.zi[1] * e[0] : 0.0;
        for (int i0 = 0; i0 < LEN; i0 += 8) {
            double v[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) v[k] = e[i0 + k];
#pragma unroll
            for (int k = 0; k < 8; ++k) v[k] = biquad_step(c, v[k], z0, z1);
#pragma unroll
            for (int k = 0; k < 8; ++k) e[i0 + k] = v[k];
        }
        z0 = t == 0 ? 0.0 : c.zi[0] * e[LEN - 1];
        z1 = t == 0 ? 0.0 : c.zi[1] * e[LEN - 1];
        for (int i0 = LEN - 8; i0 >= 0; i0 -= 8) {
            double v[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) v[k] = e[i0 + k];
#pragma unroll
            for (int k = 7; k >= 0; --k) v[k] = biquad_step(c, v[k], z0, z1);
#pragma unroll
            for (int k = 0; k < 8; ++k) e[i0 + k] = v[k];
        }
    }
    __syncthreads();
    if (t < NOTCH_EDGE)
        y[t] = ebuf[0][NOTCH_PAD + t];
    else if (t < 2 * NOTCH_EDGE)
        y[n - NOTCH_EDGE + (t - NOTCH_EDGE)] = ebuf[1][L - NOTCH_EDGE + (t - NOTCH_EDGE)];
}

int wfx_dev_notch(wfx_ctx *ctx, const void *in, int in_kind, uint64_t n, const double b[3], const double a[3], double *out)
{
    if (n <= NOTCH_PAD)
        return wfx_fail(ctx, WFX_ERR_BAD_ARG, "The length of the input vector x must be greater than padlen, which is 9.");
    notch_coef c;
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
    unsigned ib = 0;
    if (n >= NOTCH_SMALL) {
        ib = wfx_stream_grid(n - 2 * NOTCH_EDGE, 1024);
    }
    if (in_kind == WFX_IN_I16_MONO)
        WFX_LAUNCH(ctx, K_NOTCH, notch_kernel<short>, dim3(ib + 1), dim3(256), (const short *)in, n, c, out, ib);
    else
        WFX_LAUNCH(ctx, K_NOTCH, notch_kernel<double>, dim3(ib + 1), dim3(256), (const double *)in, n, c, out, ib);
    return 0;
}

// ===========================================================================
// a7  5-tap median (wefax.py:175 scipy.signal.medfilt(., 5)): zeros beyond both ends
// ===========================================================================
__device__ __forceinline__ void cswap(double &a, double &b)
{
    const double lo = fmin(a, b), hi = fmax(a, b);
    a = lo;
    b = hi;
}

__device__ __forceinline__ double median5(double a, double b, double c, double d, double e)
{
    cswap(a, b);
    cswap(d, e);
    cswap(a, d);      // a is the smallest of a, b, d, e -> not the median
    cswap(b, e);      // e is the largest of a, b, d, e  -> not the median
    cswap(b, c);      // remaining: b, c, d -> median of three
    cswap(c, d);
    cswap(b, c);
    return c;
}

__global__ void __launch_bounds__(256) median5_kernel(const double *__restrict__ r, uint64_t n, double *__restrict__ out)
{
    __shared__ double tile[1024 + 4];
    const int t = threadIdx.x;
    for (uint64_t base = (uint64_t)blockIdx.x * 1024ull; base < n; base += (uint64_t)gridDim.x * 1024ull) {
        __syncthreads();
        for (int i = t; i < 1024 + 4; i += 256) {
            const int64_t src = (int64_t)base - 2 + i;
            tile[i] = (src >= 0 && (uint64_t)src < n) ? r[src] : 0.0;
        }
        __syncthreads();
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int j = t + 256 * u;
            if (base + j < n) out[base + j] = median5(tile[j], tile[j + 1], tile[j + 2], tile[j + 3], tile[j + 4]);
        }
    }
}

int wfx_dev_median5(wfx_ctx *ctx, const double *env_raw, uint64_t n, double *env)
{
    WFX_LAUNCH(ctx, K_MEDIAN, median5_kernel, dim3(wfx_stream_grid(n, 1024)), dim3(256), env_raw, n, env);
    return 0;
}

// ===========================================================================
// a8  np.percentile(data, (0.5, 99.5)) (wefax.py:196): exact order statistics by
// most-significant-digit radix select on the 64-bit keys, four ranks at once.
// Each level reads the array once, builds one LDS histogram per distinct prefix
// (wave-uniform digits are added with one atomic per wave), flushes it with one
// global atomic per non-empty bin, and a single-workgroup scan picks the digit.
// ===========================================================================
#define SEL_BITS 11
#define SEL_BINS (1 << SEL_BITS)
#define SEL_LEVELS 6    // 11+11+11+11+11+9 = 64 bits

__device__ __forceinline__ unsigned long long f64_key(double v)
{
    unsigned long long u = (unsigned long long)__double_as_longlong(v);
    return (u >> 63) ? ~u : (u | 0x8000000000000000ull);
}
__device__ __forceinline__ double key_f64(unsigned long long k)
{
    const unsigned long long u = (k >> 63) ? (k & 0x7fffffffffffffffull) : ~k;
    return __longlong_as_double((long long)u);
}

__host__ __device__ static inline int sel_shift(int level) { return level < 5 ? 53 - 11 * level : 0; }
__host__ __device__ static inline int sel_width(int level) { return level < 5 ? 11 : 9; }

// One level: histogram of the level's digit over the elements whose higher bits equal
// a query's prefix.  Queries that share a prefix share a histogram ("owner").
__global__ void __launch_bounds__(256) select_hist(const double *__restrict__ v, uint64_t n, int level, const wfx_dev_scalars *__restrict__ s, unsigned *__restrict__ ghist)
{
    __shared__ unsigned h[4][SEL_BINS];
    __shared__ unsigned long long pfx[4];
    __shared__ int owner[4];
    const int t = threadIdx.x;
    for (int i = t; i < 4 * SEL_BINS; i += 256) (&h[0][0])[i] = 0;
    if (t == 0) {
        for (int q = 0; q < 4; ++q) {
            pfx[q] = level == 0 ? 0ull : s->sel_prefix[q];
            int o = q;
            for (int p = 0; p < q; ++p)
                if (level == 0 || s->sel_prefix[p] == s->sel_prefix[q]) {
                    o = p;
                    break;
                }
            owner[q] = o;
        }
    }
    __syncthreads();
    const int shift = sel_shift(level), width = sel_width(level);
    const unsigned dmask = (1u << width) - 1;
    const int lane = t & 63;
    bool active[4];
    unsigned long long mypfx[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        active[q] = owner[q] == q;
        mypfx[q] = pfx[q];
    }
    auto count = [&](unsigned long long key, bool valid) {
        const unsigned digit = (unsigned)(key >> shift) & dmask;
        const unsigned long long hi = level == 0 ? 0ull : (key >> (shift + width));
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            if (!active[q]) continue;                       // block-uniform
            const bool match = valid && hi == mypfx[q];
            const unsigned long long m = __ballot(match);
            if (m == 0) continue;                           // wave-uniform
            const int first = __ffsll((long long)m) - 1;
            const unsigned d0 = __shfl(digit, first);
            const bool same = !match || digit == d0;
            if (__all(same)) {
                if (lane == first) atomicAdd(&h[q][d0], (unsigned)__popcll(m));
            } else if (match) {
                atomicAdd(&h[q][digit], 1u);
            }
        }
    };
    // four values per thread and iteration (two 16-byte loads in flight)
    const uint64_t quads = (n + 3) / 4;
    const uint64_t stride = (uint64_t)gridDim.x * 256ull;
    const uint64_t nround = (quads + stride - 1) / stride;
    for (uint64_t it = 0; it < nround; ++it) {
        const uint64_t i0 = (it * stride + blockIdx.x * 256ull + t) * 4;
        double a0 = 0, a1 = 0, a2 = 0, a3 = 0;
        if (i0 + 4 <= n) {
            const double2 lo = *(const double2 *)(v + i0), hi2 = *(const double2 *)(v + i0 + 2);
            a0 = lo.x;
            a1 = lo.y;
            a2 = hi2.x;
            a3 = hi2.y;
        } else {
            if (i0 < n) a0 = v[i0];
            if (i0 + 1 < n) a1 = v[i0 + 1];
            if (i0 + 2 < n) a2 = v[i0 + 2];
        }
        count(f64_key(a0), i0 < n);
        count(f64_key(a1), i0 + 1 < n);
        count(f64_key(a2), i0 + 2 < n);
        count(f64_key(a3), i0 + 3 < n);
    }
    __syncthreads();
    for (int i = t; i < 4 * SEL_BINS; i += 256) {
        const unsigned c = (&h[0][0])[i];
        if (c) atomicAdd(&ghist[i], c);
    }
}

// numpy's _lerp (numpy/lib/_function_base_impl.py): a + (b-a)*t, or b - (b-a)*(1-t) when t >= 0.5
__device__ __forceinline__ double np_lerp(double a, double b, double t)
{
    const double diff = b - a;
    double r = a + diff * t;
    if (t >= 0.5) r = b - diff * (1 - t);
    return r;
}

// Pick each query's digit from the level's histogram: one wave per query, 32 bins per
// lane, shuffle scan.  Also clears the other histogram buffer for the next level and,
// after the last level, applies the percentile interpolation.
__global__ void __launch_bounds__(256) select_scan(int level, wfx_dev_scalars *s, const unsigned *__restrict__ ghist, unsigned *__restrict__ gnext,
                                                  uint64_t r0, uint64_t r1, uint64_t r2, uint64_t r3, int do_lerp, double gamma_lo, double gamma_hi)
{
    __shared__ double vals[4];
    const int t = threadIdx.x;
    const int q = t >> 6, lane = t & 63;
    const int width = sel_width(level);
    const int bins = 1 << width;
    int o = q;
    for (int p = 0; p < q; ++p)
        if (level == 0 || s->sel_prefix[p] == s->sel_prefix[q]) {
            o = p;
            break;
        }
    const unsigned long long rank = level == 0 ? (q == 0 ? r0 : q == 1 ? r1 : q == 2 ? r2 : r3) : s->sel_rank[q];
    const unsigned long long prefix = level == 0 ? 0ull : s->sel_prefix[q];
    const unsigned *hq = ghist + o * SEL_BINS + lane * 32;
    unsigned loc[32];
    unsigned sum = 0;                          // counts are < 2^32 (n <= 2^31)
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const uint4 u = ((const uint4 *)hq)[k];
        loc[4 * k] = u.x;
        loc[4 * k + 1] = u.y;
        loc[4 * k + 2] = u.z;
        loc[4 * k + 3] = u.w;
    }
#pragma unroll
    for (int k = 0; k < 32; ++k) {
        if (lane * 32 + k >= bins) loc[k] = 0;
        sum += loc[k];
    }
    unsigned incl = sum;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const unsigned o2 = __shfl_up(incl, off);
        if (lane >= off) incl += o2;
    }
    unsigned long long cum = incl - sum;      // exclusive prefix of this lane's first bin
    __syncthreads();                           // every wave has read s->sel_* before anyone rewrites it
    unsigned long long newp = 0, newr = 0;
    bool mine = false;
#pragma unroll
    for (int k = 0; k < 32; ++k) {
        if (!mine && rank >= cum && rank < cum + loc[k]) {
            newp = (prefix << width) | (unsigned long long)(lane * 32 + k);
            newr = rank - cum;
            mine = true;
        }
        cum += loc[k];
    }
    if (mine) {
        s->sel_prefix[q] = newp;
        s->sel_rank[q] = newr;
        if (level == SEL_LEVELS - 1) {
            s->sel_value[q] = key_f64(newp);
            vals[q] = key_f64(newp);
        }
    }
    for (int i = t; i < 4 * SEL_BINS; i += 256) gnext[i] = 0;
    if (do_lerp && level == SEL_LEVELS - 1) {
        __syncthreads();
        if (t == 0) {
            s->low = np_lerp(vals[0], vals[1], gamma_lo);
            s->high = np_lerp(vals[2], vals[3], gamma_hi);
            s->nan_count = 0;
        }
    }
}

static int select_run(wfx_ctx *ctx, const double *env, uint64_t n, const uint64_t ranks[4], wfx_dev_scalars *d_scal, int do_lerp,
                      double gamma_lo, double gamma_hi)
{
    for (int q = 0; q < 4; ++q)
        if (ranks[q] >= n) return wfx_fail(ctx, WFX_ERR_BAD_ARG, "order statistic rank %llu >= n", (unsigned long long)ranks[q]);
    const size_t hbytes = 4 * SEL_BINS * sizeof(unsigned);
    WFX_TRY(wfx_reserve(ctx, ctx->b_hist, 2 * hbytes));
    unsigned *gh[2] = {(unsigned *)ctx->b_hist.p, (unsigned *)ctx->b_hist.p + 4 * SEL_BINS};
    WFX_HIP(ctx, hipMemsetAsync(gh[0], 0, hbytes, ctx->stream));
    for (int level = 0; level < SEL_LEVELS; ++level) {
        unsigned *cur = gh[level & 1], *nxt = gh[(level + 1) & 1];
        WFX_LAUNCH(ctx, K_SELECT_HIST, select_hist, dim3(wfx_stream_grid(n, 4096)), dim3(256), env, n, level,
                   (const wfx_dev_scalars *)d_scal, cur);
        WFX_LAUNCH(ctx, K_SELECT_SCAN, select_scan, dim3(1), dim3(256), level, d_scal, (const unsigned *)cur, nxt, ranks[0], ranks[1],
                   ranks[2], ranks[3], do_lerp, gamma_lo, gamma_hi);
    }
    return 0;
}

int wfx_dev_select(wfx_ctx *ctx, const double *env, uint64_t n, const uint64_t ranks[4], wfx_dev_scalars *d_scal)
{
    return select_run(ctx, env, n, ranks, d_scal, 0, 0.0, 0.0);
}

int wfx_dev_percentiles(wfx_ctx *ctx, const double *env, uint64_t n, const uint64_t ranks[4], double gamma_lo, double gamma_hi,
                        wfx_dev_scalars *d_scal)
{
    return select_run(ctx, env, n, ranks, d_scal, 1, gamma_lo, gamma_hi);
}

__global__ void percentile_lerp_kernel(wfx_dev_scalars *s, double gamma_lo, double gamma_hi)
{
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        s->low = np_lerp(s->sel_value[0], s->sel_value[1], gamma_lo);
        s->high = np_lerp(s->sel_value[2], s->sel_value[3], gamma_hi);
        s->nan_count = 0;
    }
}

int wfx_dev_percentile_lerp(wfx_ctx *ctx, double gamma_lo, double gamma_hi, wfx_dev_scalars *d_scal)
{
    WFX_LAUNCH(ctx, K_SELECT_SCAN, percentile_lerp_kernel, dim3(1), dim3(64), d_scal, gamma_lo, gamma_hi);
    return 0;
}

// ===========================================================================
// a8  quantise (wefax.py:197-200,216): np.round(255 * (data - low) / delta), clamp
// ===========================================================================
__device__ __forceinline__ unsigned quantise_one(double v, double low, double delta, unsigned &nan)
{
    double q = 255 * (v - low) / delta;     // same operation order as numpy; IEEE division
    q = rint(q);                            // round half to even == np.round
    if (q != q) {
        nan += 1;
        return 0;
    }
    if (q < 0) q = 0;
    if (q > 255) q = 255;
    return (unsigned)q;
}

__global__ void __launch_bounds__(256) quantise_kernel(const double *__restrict__ env, uint64_t n, const wfx_dev_scalars *__restrict__ s,
                                                      uint8_t *__restrict__ out, wfx_dev_scalars *__restrict__ sout)
{
    const double low = s->low, high = s->high;
    const double delta = high - low;
    unsigned nan = 0;
    const uint64_t groups = (n + 7) / 8;
    for (uint64_t gi = blockIdx.x * 256ull + threadIdx.x; gi < groups; gi += (uint64_t)gridDim.x * 256ull) {
        const uint64_t i0 = gi * 8;
        if (i0 + 8 <= n) {
            const double2 *p = (const double2 *)(env + i0);
            const double2 a = p[0], b = p[1], c = p[2], d = p[3];
            unsigned w0 = quantise_one(a.x, low, delta, nan) | (quantise_one(a.y, low, delta, nan) << 8) |
                          (quantise_one(b.x, low, delta, nan) << 16) | (quantise_one(b.y, low, delta, nan) << 24);
            unsigned w1 = quantise_one(c.x, low, delta, nan) | (quantise_one(c.y, low, delta, nan) << 8) |
                          (quantise_one(d.x, low, delta, nan) << 16) | (quantise_one(d.y, low, delta, nan) << 24);
            *(uint2 *)(out + i0) = make_uint2(w0, w1);
        } else {
            for (uint64_t i = i0; i < n; ++i) out[i] = (uint8_t)quantise_one(env[i], low, delta, nan);
        }
    }
    // one atomic per wave that saw a NaN (rare path)
    const unsigned long long m = __ballot(nan != 0);
    if (m) {
        unsigned tot = nan;
        for (int off = 32; off > 0; off >>= 1) tot += __shfl_down(tot, off);
        if ((threadIdx.x & 63) == 0) atomicAdd(&sout->nan_count, (unsigned long long)tot);
    }
}

int wfx_dev_quantise(wfx_ctx *ctx, const double *env, uint64_t n, const wfx_dev_scalars *d_scal, uint8_t *out, wfx_dev_scalars *d_scal_out)
{
    WFX_LAUNCH(ctx, K_QUANTISE, quantise_kernel, dim3(wfx_stream_grid((n + 7) / 8, 256)), dim3(256), env, n, d_scal, out, d_scal_out);
    return 0;
}

// ===========================================================================
// a9  sync search (wefax.py:218-294)
// corr[i] = -127 * sum_{k<L}(d[i+k]-128) - sum_{k in middle run}(d[i+k]-128)
// ===========================================================================
// corr for `cnt` positions starting at global position p0; bytes d[p0 .. p0+cnt+L) are in ds
__device__ __forceinline__ void corr_from_lds(const uint8_t *ds, int cnt, int n1, int n0, int *corr_out, int t, int nthreads)
{
    const int L = 2 * n1 + n0;
    // 16 consecutive positions per thread
    for (int j0 = t * 16; j0 < cnt; j0 += nthreads * 16) {
        int sall = 0, smid = 0;
        for (int k = 0; k < L; ++k) sall += ds[j0 + k];
        for (int k = 0; k < n0; ++k) smid += ds[j0 + n1 + k];
        const int jend = min(j0 + 16, cnt);
        for (int j = j0; j < jend; ++j) {
            corr_out[j] = -127 * (sall - 128 * L) - (smid - 128 * n0);
            sall += (int)ds[j + L] - (int)ds[j];
            smid += (int)ds[j + n1 + n0] - (int)ds[j + n1];
        }
    }
}

#define CORR_CH 4096
// wave64 reductions with DPP (no LDS crossbar round trips): after the four row steps every
// lane of a 16-lane row holds the row result; row_bcast15 / row_bcast31 fold the rows
// into lane 63.
#define WFX_DPP(v, ctrl, rowmask) __builtin_amdgcn_update_dpp((v), (v), (ctrl), (rowmask), 0xf, false)
__device__ __forceinline__ int wave_max_i32(int v)
{
    v = max(v, WFX_DPP(v, 0xB1, 0xf));    // quad_perm [1,0,3,2]
    v = max(v, WFX_DPP(v, 0x4E, 0xf));    // quad_perm [2,3,0,1]
    v = max(v, WFX_DPP(v, 0x141, 0xf));   // row_half_mirror
    v = max(v, WFX_DPP(v, 0x140, 0xf));   // row_mirror
    v = max(v, WFX_DPP(v, 0x142, 0xa));   // row_bcast15 -> rows 1, 3
    v = max(v, WFX_DPP(v, 0x143, 0xc));   // row_bcast31 -> rows 2, 3
    return __builtin_amdgcn_readlane(v, 63);
}
__device__ __forceinline__ int wave_min_i32(int v)
{
    v = min(v, WFX_DPP(v, 0xB1, 0xf));
    v = min(v, WFX_DPP(v, 0x4E, 0xf));
    v = min(v, WFX_DPP(v, 0x141, 0xf));
    v = min(v, WFX_DPP(v, 0x140, 0xf));
    v = min(v, WFX_DPP(v, 0x142, 0xa));
    v = min(v, WFX_DPP(v, 0x143, 0xc));
    return __builtin_amdgcn_readlane(v, 63);
}

// first arg-max over the wave: (max correlation, smallest index holding it)
__device__ __forceinline__ void wave_first_argmax(int c, int idx, int &best_c, int &best_idx)
{
    best_c = wave_max_i32(c);
    best_idx = wave_min_i32(c == best_c ? idx : 0x7fffffff);
}

// corr[i] for every position, plus (optionally) per block of 64 positions the maximum
// and the offset of its first occurrence -- what the sequential picker consumes.
__global__ void __launch_bounds__(256) sync_corr_kernel(const uint8_t *__restrict__ d, uint64_t n, int n1, int n0, int *__restrict__ corr,
                                                       int *__restrict__ bmax, int *__restrict__ boff)
{
    __shared__ uint8_t ds[CORR_CH + 512];
    __shared__ int cs[CORR_CH];
    const int L = 2 * n1 + n0;
    const uint64_t ncorr = n > (uint64_t)L ? n - L : 0;
    const int t = threadIdx.x;
    for (uint64_t p0 = (uint64_t)blockIdx.x * CORR_CH; p0 < ncorr; p0 += (uint64_t)gridDim.x * CORR_CH) {
        const int cnt = (int)min((uint64_t)CORR_CH, ncorr - p0);
        __syncthreads();
        for (int i = t; i < cnt + L + 1; i += 256) ds[i] = (p0 + i < n) ? d[p0 + i] : 0;
        __syncthreads();
        corr_from_lds(ds, cnt, n1, n0, cs, t, 256);
        __syncthreads();
        for (int i = t; i < cnt; i += 256) corr[p0 + i] = cs[i];
        if (bmax) {
            const int lane = t & 63, wave = t >> 6;
            for (int b = wave; b * 64 < cnt; b += 4) {
                const int j = b * 64 + lane;
                int bc, bi;
                wave_first_argmax(j < cnt ? cs[j] : (int)0x80000000, lane, bc, bi);
                if (lane == 0) {
                    bmax[p0 / 64 + b] = bc;
                    boff[p0 / 64 + b] = bi;
                }
            }
        }
    }
}

int wfx_dev_sync_corr(wfx_ctx *ctx, const uint8_t *d, uint64_t n, int n1, int n0, int32_t *corr)
{
    if (2 * n1 + n0 > 500) return wfx_fail(ctx, WFX_ERR_BAD_ARG, "sync pattern too long");
    WFX_LAUNCH(ctx, K_SYNC_CORR, sync_corr_kernel, dim3(wfx_stream_grid(n, CORR_CH)), dim3(256), d, n, n1, n0, (int *)corr, (int *)nullptr,
               (int *)nullptr);
    return 0;
}

// Sequential peak picker (wefax.py:226-261) + grouping (wefax.py:263-294) in ONE
// workgroup.  The correlation and its per-64-block first-arg-max summaries are
// precomputed by sync_corr_kernel on the whole chip; here 1024 threads stream them
// through LDS in chunks (the next chunk's loads stay in flight in registers across the
// barriers, which wait for LDS only) and wave 0 replays the reference's scan: it only
// ever needs "first maximum of a range", answered from the block summaries plus the
// two partial blocks at the range ends, reduced across the wave with DPP.
#define PICK_CH 16384
#define PICK_THREADS 1024
#define PICK_PER_THREAD (PICK_CH / PICK_THREADS)

__device__ __forceinline__ bool dev_ok(double frame_samples, long long x)
{
    // wefax.py:263-267: max_deviation > x > min_deviation
    return (frame_samples + 500 > (double)x) && ((double)x > frame_samples - 500);
}

// workgroup barrier that orders LDS traffic only: global loads already issued (the
// prefetch of the next chunk) stay in flight across it
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

__global__ void __launch_bounds__(PICK_THREADS) sync_pick_kernel(const int *__restrict__ corr, const int *__restrict__ bmax, const int *__restrict__ boff,
                                                                 uint64_t n, int n1, int n0, long long mind, double frame_samples, int width,
                                                                 wfx_dev_scalars *__restrict__ s)
{
    __shared__ int cs[PICK_CH];
    __shared__ int smc[PICK_CH / 64], smi[PICK_CH / 64];      // per 64-block: max correlation, first index of it
    __shared__ long long pk_s[WFX_MAX_PEAKS + 1], first_s[WFX_MAX_PEAKS + 1];
    __shared__ int done_flag, np_s, hit_s;
    const int t = threadIdx.x;
    const int lane = t & 63;
    const int L = 2 * n1 + n0;
    const uint64_t ncorr = n > (uint64_t)L ? n - L : 0;
    const uint64_t nblk = (ncorr + 63) / 64;
    const int CMIN = (int)0x80000000;
    // picker state (meaningful in wave 0, kept uniform across its lanes)
    long long pos = 0;
    int val = 0, np = 1, hit = 0;
    if (t == 0) {
        done_flag = 0;
        first_s[0] = 0;
    }
    int pre[PICK_PER_THREAD], pre_c = CMIN, pre_o = 0;
    auto prefetch = [&](uint64_t p0) {
#pragma unroll
        for (int k = 0; k < PICK_PER_THREAD; ++k) {
            const uint64_t g = p0 + (uint64_t)k * PICK_THREADS + t;
            pre[k] = g < ncorr ? corr[g] : 0;
        }
        if (t < PICK_CH / 64) {
            const uint64_t bg = p0 / 64 + t;
            pre_c = bg < nblk ? bmax[bg] : CMIN;
            pre_o = bg < nblk ? boff[bg] : 0;
        }
    };
    if (ncorr) prefetch(0);
    for (uint64_t p0 = 0; p0 < ncorr; p0 += PICK_CH) {
        const int cnt = (int)min((uint64_t)PICK_CH, ncorr - p0);
#pragma unroll
        for (int k = 0; k < PICK_PER_THREAD; ++k) cs[k * PICK_THREADS + t] = pre[k];
        if (t < PICK_CH / 64) {
            smc[t] = pre_c;
            smi[t] = t * 64 + pre_o;
        }
        lds_barrier();
        if (p0 + PICK_CH < ncorr) prefetch(p0 + PICK_CH);      // in flight while this chunk is scanned
        if (t < 64) {
            long long i = (long long)p0;
            const long long ce = (long long)p0 + cnt;
            while (i < ce) {
                if (i - pos > mind) {
                    if (lane == 0) {
                        pk_s[np - 1] = pos;
                        first_s[np] = i;
                    }
                    pos = i;
                    val = cs[i - (long long)p0];
                    ++np;
                    ++i;
                    if (np == WFX_MAX_PEAKS) {
                        hit = 1;
                        break;
                    }
                } else {
                    long long lim = pos + mind;
                    if (lim > ce - 1) lim = ce - 1;
                    const int li = (int)(i - (long long)p0), ll = (int)(lim - (long long)p0);
                    const int b_lo = li >> 6, b_hi = ll >> 6;
                    int c = CMIN, ci = 0x7fffffff;
                    {
                        const int j = (b_lo << 6) + lane;                   // head (or the only) block
                        if (j >= li && j <= ll) {
                            c = cs[j];
                            ci = j;
                        }
                    }
                    if (b_hi > b_lo) {
                        for (int b = b_lo + 1 + lane; b < b_hi; b += 64) {  // whole blocks in between (ascending index)
                            const int c2 = smc[b];
                            if (c2 > c) {
                                c = c2;
                                ci = smi[b];
                            }
                        }
                        const int j = (b_hi << 6) + lane;                   // tail block (largest indices)
                        if (j <= ll) {
                            const int c2 = cs[j];
                            if (c2 > c) {
                                c = c2;
                                ci = j;
                            }
                        }
                    }
                    int bc, bi;
                    wave_first_argmax(c, ci, bc, bi);
                    if (bc > val && bi != 0x7fffffff) {
                        val = bc;
                        pos = (long long)p0 + bi;
                    }
                    i = lim + 1;
                }
            }
            if (hit && lane == 0) done_flag = 1;
        }
        lds_barrier();
        if (done_flag) break;
    }
    if (t == 0) {
        pk_s[np - 1] = pos;
        np_s = np;
        hit_s = hit;
    }
    __syncthreads();
    np = np_s;
    for (int i = t; i < np; i += PICK_THREADS) {
        s->peak_pos[i] = pk_s[i];
        s->first_pos[i] = first_s[i];
    }
    if (t != 0) return;
    s->npeaks = np;
    s->hit_limit = hit_s;
    const long long *pk = pk_s;
    // ---- grouping (wefax.py:269-294) -----------------------------------------
    int nclear = 0;
    for (int i = 1; i < np - 1; ++i)
        if (dev_ok(frame_samples, pk[i] - pk[i - 1])) ++nclear;
    int nclosed = 0, best_start = 0, best_len = -1, g_start = 0, g_len = 0;
    for (int i = 1; i < nclear - 1; ++i) {
        if (dev_ok(frame_samples, pk[i] - pk[i - 1])) {
            if (g_len == 0) g_start = i;
            ++g_len;
        } else {
            if (g_len > best_len) {
                best_len = g_len;
                best_start = g_start;
            }
            ++nclosed;
            g_len = 0;
        }
    }
    long long start = 0;
    if (nclosed == 0) {
        s->no_group = 1;
        s->n_phasing = 0;
    } else {
        s->no_group = 0;
        s->n_phasing = best_len;
        for (int k = 0; k < best_len; ++k) s->phasing[k] = pk[best_start + k];
        if (best_len > 0) start = pk[best_start + best_len - 1];     // wefax.py:80
    }
    s->start_frame = start;
    s->height = (nclosed == 0 || width <= 0) ? 0 : (int)(((long long)n - start) / width);
}

int wfx_dev_sync_pick(wfx_ctx *ctx, const uint8_t *d, uint64_t n, int n1, int n0, int64_t mindistance, double frame_samples,
                      int width, wfx_dev_scalars *d_scal)
{
    if (2 * n1 + n0 > 500 || n1 < 0 || n0 < 0) return wfx_fail(ctx, WFX_ERR_BAD_ARG, "sync pattern length out of range");
    const size_t nblk = (size_t)n / 64 + 2;
    WFX_TRY(wfx_reserve(ctx, ctx->b_corr, (size_t)n * 4 + 64));
    WFX_TRY(wfx_reserve(ctx, ctx->b_tmp2, nblk * 8));
    int *bmax = (int *)ctx->b_tmp2.p, *boff = bmax + nblk;
    WFX_LAUNCH(ctx, K_SYNC_CORR, sync_corr_kernel, dim3(wfx_stream_grid(n, CORR_CH)), dim3(256), d, n, n1, n0, (int *)ctx->b_corr.p, bmax, boff);
    WFX_LAUNCH(ctx, K_SYNC_PICK, sync_pick_kernel, dim3(1), dim3(PICK_THREADS), (const int *)ctx->b_corr.p, (const int *)bmax,
               (const int *)boff, n, n1, n0, (long long)mindistance, frame_samples, width, d_scal);
    return 0;
}

// ===========================================================================
// a10 image assembly (wefax.py:296-327): pixel(x, y) = 255 - d[start + y w + x], then
// PIL Image.resize((w, 4h)): width unchanged -> only Pillow's vertical pass
// (ImagingResampleVertical_8bpc) with BICUBIC coefficients in 22-bit fixed point.
// ===========================================================================
__device__ __forceinline__ double pil_bicubic(double x)
{
    const double a = -0.5;
    if (x < 0.0) x = -x;
    if (x < 1.0) return ((a + 2.0) * x - (a + 3.0)) * x * x + 1;
    if (x < 2.0) return (((x - 5) * x + 8) * x - 4) * a;
    return 0.0;
}

// coefficients of output row yy (precompute_coeffs + normalize_coeffs_8bpc, Resample.c)
__device__ __forceinline__ void pil_row_coeffs(int yy, int h_in, int h_out, int &ymin, int &cnt, int kk[5])
{
    const double scale = (double)h_in / (double)h_out;
    const double filterscale = scale < 1.0 ? 1.0 : scale;
    const double support = 2.0 * filterscale;
    const double ss = 1.0 / filterscale;
    const double center = (yy + 0.5) * scale;
    ymin = (int)(center - support + 0.5);
    if (ymin < 0) ymin = 0;
    int ymax = (int)(center + support + 0.5);
    if (ymax > h_in) ymax = h_in;
    cnt = ymax - ymin;
    double w[5], ww = 0.0;
#pragma unroll
    for (int y = 0; y < 5; ++y) {
        w[y] = y < cnt ? pil_bicubic((y + ymin - center + 0.5) * ss) : 0.0;
        if (y < cnt) ww += w[y];
    }
#pragma unroll
    for (int y = 0; y < 5; ++y) {
        double v = w[y];
        if (y < cnt && ww != 0.0) v /= ww;
        kk[y] = y < cnt ? (v < 0 ? (int)(-0.5 + v * 4194304.0) : (int)(0.5 + v * 4194304.0)) : 0;
    }
}

// One workgroup per SOURCE row y: it produces the four output rows 4y..4y+3, which
// draw on source rows y-2..y+2, so every source byte is fetched once per workgroup
// (aligned dword loads + v_alignbyte for the arbitrary byte offset start + y*w) and
// every thread turns 8 source columns into 4 x 8 output pixels (dword stores).
__device__ __forceinline__ void load8_any(const uint8_t *p, unsigned &lo, unsigned &hi)
{
    const uintptr_t a = (uintptr_t)p;
    const unsigned *q = (const unsigned *)(a & ~(uintptr_t)3);
    const unsigned sh = (unsigned)(a & 3);
    const unsigned w0 = q[0], w1 = q[1], w2 = q[2];      // the stream buffer is padded by 64 bytes
    lo = __builtin_amdgcn_alignbyte(w1, w0, sh);
    hi = __builtin_amdgcn_alignbyte(w2, w1, sh);
}

__global__ void __launch_bounds__(256) image_kernel(const uint8_t *__restrict__ d, uint64_t n, int w, const wfx_dev_scalars *__restrict__ s,
                                                   uint8_t *__restrict__ img)
{
    const int h = s->height;
    const int y = blockIdx.x;
    if (y >= h) return;
    const uint64_t start = (uint64_t)s->start_frame;
    // coefficients of the four output rows (row-uniform, computed redundantly per thread)
    int ymin[4], cnt[4], kk[4][5];
#pragma unroll
    for (int r = 0; r < 4; ++r) pil_row_coeffs(4 * y + r, h, 4 * h, ymin[r], cnt[r], kk[r]);
    const int ylo = max(y - 2, 0), yhi = min(y + 2, h - 1);      // source rows any of the four can touch
    for (int x0 = threadIdx.x * 8; x0 < w; x0 += 256 * 8) {
        const int nx = min(8, w - x0);
        int acc[4][8];
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int e = 0; e < 8; ++e) acc[r][e] = 1 << 21;
#pragma unroll
        for (int dy = 0; dy < 5; ++dy) {
            const int sy = ylo + dy;
            if (sy > yhi) break;
            unsigned lo, hi;
            load8_any(d + start + (uint64_t)sy * w + x0, lo, hi);
            int px[8];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                px[e] = 255 - (int)((lo >> (8 * e)) & 0xff);
                px[4 + e] = 255 - (int)((hi >> (8 * e)) & 0xff);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int ki = sy - ymin[r];
                if (ki >= 0 && ki < cnt[r]) {
                    const int k = kk[r][ki];
#pragma unroll
                    for (int e = 0; e < 8; ++e) acc[r][e] += px[e] * k;
                }
            }
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            uint8_t *dst = img + (uint64_t)(4 * y + r) * w + x0;
            unsigned o[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int v = acc[r][e] >> 22;
                o[e] = (unsigned)(v < 0 ? 0 : (v > 255 ? 255 : v));
            }
            if (nx == 8 && (((uintptr_t)dst) & 3) == 0) {
                ((unsigned *)dst)[0] = o[0] | (o[1] << 8) | (o[2] << 16) | (o[3] << 24);
                ((unsigned *)dst)[1] = o[4] | (o[5] << 8) | (o[6] << 16) | (o[7] << 24);
            } else {
                for (int e = 0; e < nx; ++e) dst[e] = (uint8_t)o[e];
            }
        }
    }
}

int wfx_dev_image(wfx_ctx *ctx, const uint8_t *d, uint64_t n, int w, int h_max, const wfx_dev_scalars *d_scal, uint8_t *img)
{
    if (h_max <= 0 || w <= 0) return 0;
    WFX_LAUNCH(ctx, K_IMAGE, image_kernel, dim3((unsigned)h_max), dim3(256), d, n, w, d_scal, img);
    return 0;
}
