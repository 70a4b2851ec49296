// Stencil / elementwise / order-statistic / sync / image kernels of the WEFAX
// hot path.  Each block cites the reference lines it replaces.  All of these are
// HBM-bound byte / double streaming kernels (no MFMA): 16-byte accesses where the
// data allows, LDS tiles for the stencils, wave64 ballots and shuffles for the
// reductions, one global atomic per workgroup.
#include "wfx_internal.h"

// workgroup barrier that orders LDS traffic only: global loads already issued (a
// register prefetch of the next tile) stay in flight across it, which __syncthreads()
// would drain with s_waitcnt vmcnt(0)
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

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

// the other sample formats of a two-channel wav (numpy scalar semantics of wefax.py:372): the add in the file's dtype, then / 2
template <int KIND>
__global__ void __launch_bounds__(256) merge_any_kernel(const void *__restrict__ lr, uint64_t n, double *__restrict__ out)
{
    for (uint64_t i = blockIdx.x * 256ull + threadIdx.x; i < n; i += (uint64_t)gridDim.x * 256ull) {
        double v;
        if (KIND == WFX_IN_U8_STEREO) {
            const uchar2 p = ((const uchar2 *)lr)[i];
            v = (double)(unsigned char)((unsigned)p.x + (unsigned)p.y) / 2.0;          // wraps modulo 2^8
        } else if (KIND == WFX_IN_I32_STEREO) {
            const int2 p = ((const int2 *)lr)[i];
            v = (double)(int)((unsigned)p.x + (unsigned)p.y) / 2.0;                    // wraps modulo 2^32
        } else {
            const float2 p = ((const float2 *)lr)[i];
            v = (double)(__fadd_rn(p.x, p.y) / 2.0f);                                   // float32 sum, float32 quotient
        }
        out[i] = v;
    }
}

int wfx_dev_merge_any(wfx_ctx *ctx, const void *lr, int in_kind, uint64_t n, double *out)
{
    const dim3 g(wfx_stream_grid(n, 256)), b(256);
    switch (in_kind) {
    case WFX_IN_I16_STEREO: return wfx_dev_merge(ctx, (const int16_t *)lr, n, out);
    case WFX_IN_U8_STEREO: WFX_LAUNCH(ctx, K_MERGE, merge_any_kernel<WFX_IN_U8_STEREO>, g, b, lr, n, out); return 0;
    case WFX_IN_I32_STEREO: WFX_LAUNCH(ctx, K_MERGE, merge_any_kernel<WFX_IN_I32_STEREO>, g, b, lr, n, out); return 0;
    case WFX_IN_F32_STEREO: WFX_LAUNCH(ctx, K_MERGE, merge_any_kernel<WFX_IN_F32_STEREO>, g, b, lr, n, out); return 0;
    default: return wfx_fail(ctx, WFX_ERR_BAD_ARG, "merge: input kind %d has one channel", in_kind);
    }
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
#include "wfx_notch.h"

// outputs per lane of the interior form: 8 halve the LDS reads per output and measure faster on float64 input (141 vs 155 us on the
// 60-minute sizes), 4 on int16 input (29 vs 32.5 us on the 10-minute capture: nine 2-byte loads per lane and tile otherwise)
template <typename TIN> struct notch_nu { static constexpr int value = sizeof(TIN) == 2 ? 4 : 8; };
template <typename TIN>
__global__ void __launch_bounds__(256, 4) notch_kernel(const TIN *__restrict__ x, uint64_t n, notch_coef c, double *__restrict__ y, unsigned interior_blocks,
                                                       int exact_edges, wfx_dev_scalars *__restrict__ clear)
{
    constexpr int NU = notch_nu<TIN>::value, TOUT = 256 * NU;       // outputs per lane and per tile
    constexpr int TLEN = TOUT + 2 * NOTCH_K;
    constexpr int NPRE = (TLEN + 255) / 256;
    __shared__ double tile[TLEN + TLEN / 4 + 4];
    __shared__ double ebuf[2][NOTCH_SMALL + 2 * NOTCH_PAD + 8];
    const int t = threadIdx.x;
    if (blockIdx.x < interior_blocks) {
        // interior outputs [lo, hi) = [EDGE, n - EDGE); the next tile's samples are fetched
        // into registers while the current tile is filtered
        // exact_edges: bit 0 = the segment starts at the true start of the capture, bit 1 = it ends at the
        // true end (filtfilt's odd extension + lfilter_zi there); otherwise the FIR form runs up to K
        // samples from that end and the caller's halo covers the rest
        const uint64_t lo = (exact_edges & 1) ? NOTCH_EDGE : NOTCH_K, hi = (exact_edges & 2) ? n - NOTCH_EDGE : n - NOTCH_K;
        const uint64_t step = (uint64_t)interior_blocks * (uint64_t)TOUT;
        // raw samples, unconditional loads from a clamped index: a guarded load compiles to a branch with its
        // own s_waitcnt, which serialises the five loads of a tile instead of leaving them in flight (36 -> 27 us)
        TIN pre[NPRE];
        auto prefetch = [&](uint64_t base) {
#pragma unroll
            for (int k = 0; k < NPRE; ++k) {
                const uint64_t src = base - NOTCH_K + (uint64_t)(t + 256 * k);       // >= lo - K >= 0
                pre[k] = x[src < n ? src : n - 1];
            }
        };
        uint64_t base = lo + (uint64_t)blockIdx.x * (uint64_t)TOUT;
        if (base < hi) prefetch(base);
        for (; base < hi; base += step) {
            lds_barrier();
#pragma unroll
            for (int k = 0; k < NPRE; ++k) {
                const int i = t + 256 * k;
                const uint64_t src = base - NOTCH_K + (uint64_t)i;
                if (i < TLEN) tile[i + (i >> 2)] = src < n ? (double)pre[k] : 0.0;
            }
            lds_barrier();
            if (base + step < hi) prefetch(base + step);
            // window element i feeds output u with tap |i - u - K|: one LDS read, up to NU FMAs
            double acc[NU];
#pragma unroll
            for (int u = 0; u < NU; ++u) acc[u] = 0.0;
#pragma unroll
            for (int i = 0; i < 2 * NOTCH_K + NU; ++i) {
                const double wv = tile[(NU * t + i) + ((NU * t + i) >> 2)];
#pragma unroll
                for (int u = 0; u < NU; ++u) {
                    const int k = i - u - NOTCH_K;
                    if (k >= -NOTCH_K && k <= NOTCH_K) acc[u] = fma(c.g[k < 0 ? -k : k], wv, acc[u]);
                }
            }
            // a lane's NU outputs are contiguous: exchange through LDS so that each store
            // instruction writes 64 x 16 contiguous bytes instead of 64 quarter lines
            lds_barrier();
#pragma unroll
            for (int u = 0; u < NU; ++u) tile[(NU * t + u) + ((NU * t + u) >> 2)] = acc[u];
            lds_barrier();
#pragma unroll
            for (int half = 0; half < NU / 2; ++half) {
                const int j = half * 512 + 2 * t;                 // even -> same padded group of four
                const uint64_t o = base + j;                      // base is even (64 or 24 + k*1024): 16-byte aligned
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
    if (clear)       // the decode's device scalars start from zero (this is the first kernel of the path that sees them)
        for (int i = t; i < (int)(sizeof(wfx_dev_scalars) / 8); i += 256) ((unsigned long long *)clear)[i] = 0ull;
    if (n < NOTCH_SMALL) {      // (only reached through wfx_dev_notch: both ends are true edges)
        // whole signal with the exact recurrence (one thread)
        if (t == 0) {
            double *e = ebuf[0];
            const int len = (int)n + 2 * NOTCH_PAD;
            for (int i = 0; i < NOTCH_PAD; ++i) e[i] = notch_left<TIN>(c, x, NOTCH_PAD - i);
            for (int i = 0; i < (int)n; ++i) e[NOTCH_PAD + i] = (double)x[i];
            for (int i = 0; i < NOTCH_PAD; ++i) e[NOTCH_PAD + n + i] = notch_right<TIN>(c, x, n, i + 1);
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
        ebuf[0][i] = i < NOTCH_PAD ? notch_left<TIN>(c, x, NOTCH_PAD - i) : (double)x[i - NOTCH_PAD];
        ebuf[1][i] = i < L ? (double)x[n - L + i] : notch_right<TIN>(c, x, n, i - L + 1);
    }
    __syncthreads();
    if ((t == 0 && (exact_edges & 1)) || (t == 64 && (exact_edges & 2))) {
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
    if (t < NOTCH_EDGE) {
        if (exact_edges & 1) y[t] = ebuf[0][NOTCH_PAD + t];
    } else if (t < 2 * NOTCH_EDGE) {
        if (exact_edges & 2) y[n - NOTCH_EDGE + (t - NOTCH_EDGE)] = ebuf[1][L - NOTCH_EDGE + (t - NOTCH_EDGE)];
    }
}

// interior workgroups: 4 per CU, each walks its tiles of 256 * NU samples with the next tile's loads in flight
static unsigned notch_grid(uint64_t n_interior, int nu)
{
    return std::min(wfx_blocks(n_interior, 256 * nu), 1024u);
}

// ---- filtfilt for any stable biquad (the live path designs the notch at the sound card's rate:
// data_packet.py:430-432; at 48 kHz the pole radius is 0.84 and the 49-tap form above does not apply) --------
// The recurrence forgets its state at the rate of the pole radius r: a lane that starts `warm` samples early
// from a zero state holds, after the warm-up, the state of the sequential run to within r^warm (< 1e-20
// relative, far below one ulp), and from there on performs the same operations on the same values.  Each lane
// owns GEN_CHUNK outputs; the lane whose warm-up reaches the end of the extended signal starts there with
// lfilter_zi, exactly as scipy does.  Pass 1 runs forward over the odd-extended input into `fwd`
// (n + 2*PAD values), pass 2 backward over `fwd` into y.
#define GEN_CHUNK 64
template <typename TIN>
__device__ __forceinline__ double notch_ext_at(const notch_coef &c, const TIN *x, uint64_t n, int64_t i)   // i in [0, n + 2*PAD)
{
    if (i < NOTCH_PAD) return notch_left<TIN>(c, x, NOTCH_PAD - (int)i);
    if (i >= (int64_t)n + NOTCH_PAD) return notch_right<TIN>(c, x, n, (int)(i - (int64_t)n - NOTCH_PAD) + 1);
    return (double)x[i - NOTCH_PAD];
}

template <typename TIN>
__global__ void __launch_bounds__(64) biquad_forward_kernel(const TIN *__restrict__ x, uint64_t n, notch_coef c, int warm, double *__restrict__ fwd)
{
    const int64_t len = (int64_t)n + 2 * NOTCH_PAD;
    const int64_t first = ((int64_t)blockIdx.x * 64 + threadIdx.x) * GEN_CHUNK;
    if (first >= len) return;
    int64_t i = first - warm;
    double z0 = 0.0, z1 = 0.0;
    if (i <= 0) {
        i = 0;
        const double e0 = notch_ext_at<TIN>(c, x, n, 0);
        z0 = c.zi[0] * e0;
        z1 = c.zi[1] * e0;
    }
    const int64_t last = first + GEN_CHUNK < len ? first + GEN_CHUNK : len;
    for (; i < last; i += 8) {
        double v[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = i + k < last ? notch_ext_at<TIN>(c, x, n, i + k) : 0.0;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            if (i + k < last) {
                v[k] = biquad_step(c, v[k], z0, z1);
                if (i + k >= first) fwd[i + k] = v[k];
            }
        }
    }
}

__global__ void __launch_bounds__(64) biquad_backward_kernel(const double *__restrict__ fwd, uint64_t n, notch_coef c, int warm, double *__restrict__ y)
{
    const int64_t len = (int64_t)n + 2 * NOTCH_PAD;
    const int64_t first = ((int64_t)blockIdx.x * 64 + threadIdx.x) * GEN_CHUNK;     // this lane's outputs: [first, last)
    if (first >= len) return;
    const int64_t last = first + GEN_CHUNK < len ? first + GEN_CHUNK : len;
    int64_t i = last - 1 + warm;
    double z0 = 0.0, z1 = 0.0;
    if (i >= len - 1) {
        i = len - 1;
        const double e0 = fwd[len - 1];
        z0 = c.zi[0] * e0;
        z1 = c.zi[1] * e0;
    }
    for (; i >= first; i -= 8) {
        double v[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = i - k >= first ? fwd[i - k] : 0.0;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int64_t j = i - k;
            if (j >= first) {
                v[k] = biquad_step(c, v[k], z0, z1);
                if (j < last && j >= NOTCH_PAD && j < (int64_t)n + NOTCH_PAD) y[j - NOTCH_PAD] = v[k];
            }
        }
    }
}

// largest pole modulus of 1 + a1 z^-1 + a2 z^-2
static int notch_general(wfx_ctx *ctx, const void *in, int in_kind, uint64_t n, const notch_coef &c, double radius, double *out)
{
    if (!(radius < 0.9995)) return wfx_fail(ctx, WFX_ERR_BAD_ARG, "notch: pole radius %.6f is too close to the unit circle", radius);
    int warm = radius > 0.0 ? (int)ceil(log(1e-20) / log(radius)) + 8 : 8;
    warm = (warm + 7) & ~7;
    const uint64_t len = n + 2 * NOTCH_PAD;
    WFX_TRY(wfx_reserve(ctx, ctx->b_work2, len * sizeof(double)));
    double *fwd = (double *)ctx->b_work2.p;
    const unsigned lanes = (unsigned)((len + GEN_CHUNK - 1) / GEN_CHUNK), grid = (lanes + 63) / 64;
    if (in_kind == WFX_IN_I16_MONO)
        WFX_LAUNCH(ctx, K_NOTCH, biquad_forward_kernel<short>, dim3(grid), dim3(64), (const short *)in, n, c, warm, fwd);
    else
        WFX_LAUNCH(ctx, K_NOTCH, biquad_forward_kernel<double>, dim3(grid), dim3(64), (const double *)in, n, c, warm, fwd);
    WFX_LAUNCH(ctx, K_NOTCH, biquad_backward_kernel, dim3(grid), dim3(64), (const double *)fwd, n, c, warm, out);
    return 0;
}

int wfx_dev_notch(wfx_ctx *ctx, const void *in, int in_kind, uint64_t n, const double b[3], const double a[3], double *out, wfx_dev_scalars *clear,
                  bool *cleared, const double *ext18)
{
    if (cleared) *cleared = false;
    if (n <= NOTCH_PAD)
        return wfx_fail(ctx, WFX_ERR_BAD_ARG, "The length of the input vector x must be greater than padlen, which is 9.");
    notch_coef c;
    notch_prepare(c, b, a, ext18);
    // the 49-tap form holds while the impulse response has died within NOTCH_K samples (the reference's own
    // design at 11 025 Hz: radius 0.21); any other stable biquad takes the chunked recurrence
    const double radius = biquad_pole_radius(a);
    if (n >= NOTCH_SMALL && pow(radius, NOTCH_K) > 1e-16) return notch_general(ctx, in, in_kind, n, c, radius, out);
    unsigned ib = 0;
    if (n >= NOTCH_SMALL) {
        ib = notch_grid(n - 2 * NOTCH_EDGE, in_kind == WFX_IN_I16_MONO ? notch_nu<short>::value : notch_nu<double>::value);
    }
    if (in_kind == WFX_IN_I16_MONO)
        WFX_LAUNCH(ctx, K_NOTCH, notch_kernel<short>, dim3(ib + 1), dim3(256), (const short *)in, n, c, out, ib, 3, clear);
    else
        WFX_LAUNCH(ctx, K_NOTCH, notch_kernel<double>, dim3(ib + 1), dim3(256), (const double *)in, n, c, out, ib, 3, clear);
    if (cleared) *cleared = clear != nullptr;
    return 0;
}

// segment of a longer capture: exact filtfilt edges only where the segment touches the capture's
// true start (edge_flags bit 0) / end (bit 1); elsewhere the FIR form, valid K samples from the end
int wfx_dev_notch_fir_only(wfx_ctx *ctx, const void *in, int in_kind, uint64_t n, const double b[3], const double a[3], double *out,
                           int edge_flags, const double *ext18)
{
    if (n < NOTCH_SMALL) return wfx_fail(ctx, WFX_ERR_BAD_ARG, "segment of %llu samples is too short for the notch", (unsigned long long)n);
    // the 49-tap form is filtfilt only while the impulse response has died within NOTCH_K samples (wfx_dev_notch falls back to the
    // chunked recurrence otherwise; a segment has no such fallback: its halo is sized for the 49 taps)
    const double radius = biquad_pole_radius(a);
    if (pow(radius, NOTCH_K) > 1e-16)
        return wfx_fail(ctx, WFX_ERR_BAD_ARG, "notch: pole radius %.4f is too large for the 49-tap segment form (design at 11 025 Hz with Q <= ~1)", radius);
    notch_coef c;
    notch_prepare(c, b, a, ext18);
    const unsigned ib = notch_grid(n - 2 * NOTCH_K, in_kind == WFX_IN_I16_MONO ? notch_nu<short>::value : notch_nu<double>::value);
    if (in_kind == WFX_IN_I16_MONO)
        WFX_LAUNCH(ctx, K_NOTCH, notch_kernel<short>, dim3(ib + (edge_flags ? 1 : 0)), dim3(256), (const short *)in, n, c, out, ib, edge_flags & 3, (wfx_dev_scalars *)nullptr);
    else if (in_kind == WFX_IN_F64_MONO)
        WFX_LAUNCH(ctx, K_NOTCH, notch_kernel<double>, dim3(ib + (edge_flags ? 1 : 0)), dim3(256), (const double *)in, n, c, out, ib, edge_flags & 3, (wfx_dev_scalars *)nullptr);
    else
        return wfx_fail(ctx, WFX_ERR_BAD_ARG, "notch: input kind %d", in_kind);
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

__global__ void __launch_bounds__(256) median5_kernel(const double *__restrict__ r, uint64_t n, double *__restrict__ out, unsigned *__restrict__ l0hist)
{
    __shared__ double tile[1024 + 4];
    __shared__ unsigned h0[WFX_SEL_BINS];
    const int t = threadIdx.x;
    if (l0hist)
        for (int i = t; i < WFX_SEL_BINS; i += 256) h0[i] = 0;
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
            const bool valid = base + j < n;
            double m = 0.0;
            if (valid) {
                m = median5(tile[j], tile[j + 1], tile[j + 2], tile[j + 3], tile[j + 4]);
                out[base + j] = m;
            }
            if (l0hist) wfx_sel_count(h0, (unsigned)(wfx_f64_key(m) >> 53), valid, t & 63);
        }
    }
    if (l0hist) {
        __syncthreads();
        for (int i = t; i < WFX_SEL_BINS; i += 256)
            if (h0[i]) atomicAdd(&l0hist[i], h0[i]);
    }
}

int wfx_dev_median5(wfx_ctx *ctx, const double *env_raw, uint64_t n, double *env, unsigned *l0hist)
{
    WFX_LAUNCH(ctx, K_MEDIAN, median5_kernel, dim3(wfx_stream_grid(n, 1024)), dim3(256), env_raw, n, env, l0hist);
    return 0;
}

// numpy's _lerp (numpy/lib/_function_base_impl.py): a + (b-a)*t, or b - (b-a)*(1-t) when t >= 0.5
__device__ __forceinline__ double np_lerp(double a, double b, double t)
{
    const double diff = b - a;
    double r = a + diff * t;
    if (t >= 0.5) r = b - diff * (1 - t);
    return r;
}

// ===========================================================================
// a8  np.percentile(data, (0.5, 99.5)) (wefax.py:196): exact order statistics by
// most-significant-digit radix select on the 64-bit keys, four ranks at once.
//
//   level 0  bits 63..53   histogram fused into the kernel that writes the envelope
//   level 1  bits 52..42   one pass over the envelope (select_l1_kernel)
//   compact               one pass: keys whose top 22 bits match a query are appended
//                         to that query's candidate list (select_compact_kernel)
//   levels 2..5           one workgroup finishes on the candidate lists, applies
//                         numpy's percentile interpolation and clears the histograms
//
// Every kernel re-derives the digit choices it needs from the previous level's global
// histogram in its prologue (block 0 also records them), so there are no separate scan
// launches.  Histograms live in LDS (wave-uniform digits cost one atomic per wave) and
// are flushed with one global atomic per non-empty bin.
// ===========================================================================
#define SEL_BITS 11
#define SEL_BINS (1 << SEL_BITS)
#define SEL_LEVELS 6    // 11+11+11+11+11+9 = 64 bits

#define f64_key wfx_f64_key
__device__ __forceinline__ double key_f64(unsigned long long k)
{
    const unsigned long long u = (k >> 63) ? (k & 0x7fffffffffffffffull) : ~k;
    return __longlong_as_double((long long)u);
}

__host__ __device__ static inline int sel_shift(int level) { return level < 5 ? 53 - 11 * level : 0; }
__host__ __device__ static inline int sel_width(int level) { return level < 5 ? 11 : 9; }

// layout of the select workspace (unsigned words)
#define SEL_H0 0                          // level-0 histogram [2048]
#define SEL_H1 SEL_BINS                   // level-1 histograms [4][2048]
#define SEL_CNT (5 * SEL_BINS)            // candidate counts [4]
#define SEL_TICKET (5 * SEL_BINS + 4)     // finish kernel: workgroups done
#define SEL_WORDS (5 * SEL_BINS + 8)

#define sel_count wfx_sel_count

// exclusive prefix of v over the threads of the block (NT <= 1024, multiple of 64)
template <int NT>
__device__ __forceinline__ unsigned block_excl_scan(unsigned v, unsigned *wave_tot /* LDS [NT/64] */, unsigned *total)
{
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    unsigned incl = v;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const unsigned o = __shfl_up(incl, off);
        if (lane >= off) incl += o;
    }
    if (lane == 63) wave_tot[wave] = incl;
    __syncthreads();
    unsigned base = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < NT / 64; ++w) {
        const unsigned x = wave_tot[w];
        if (w < wave) base += x;
        tot += x;
    }
    __syncthreads();
    if (total) *total = tot;
    return base + incl - v;
}

// For the 4 (owner, rank) queries: find, in the histogram of the query's owner, the bin that
// holds the rank.  One wave per query (blocks of >= 256 threads), 32 bins per lane, no block
// barrier inside; results go to LDS arrays (the caller synchronises).
__device__ __forceinline__ void sel_pick_digits(const unsigned *hist /* [4][SEL_BINS] or shared [SEL_BINS] */, bool shared_hist, int bins,
                                                const int *owner, const unsigned long long *rank_in, unsigned *digit_out,
                                                unsigned long long *rank_out)
{
    const int q = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (q >= 4) return;
    const unsigned *hq = (shared_hist ? hist : hist + owner[q] * SEL_BINS) + lane * 32;
    unsigned loc[32], sum = 0;
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
    unsigned long long cum = incl - sum;
    const unsigned long long r = rank_in[q];
#pragma unroll
    for (int k = 0; k < 32; ++k) {
        if (r >= cum && r < cum + loc[k]) {
            digit_out[q] = (unsigned)(lane * 32 + k);
            rank_out[q] = r - cum;
        }
        cum += loc[k];
    }
}

__device__ __forceinline__ void sel_owners(const unsigned long long *pfx, int *owner)
{
    for (int q = 0; q < 4; ++q) {
        int o = q;
        for (int p = 0; p < q; ++p)
            if (pfx[p] == pfx[q]) {
                o = p;
                break;
            }
        owner[q] = o;
    }
}

// one generic level with the prefixes given by the host (sharded decode: the histograms of
// all ranks are summed between the levels, the host picks the digits)
__global__ void __launch_bounds__(256) select_level_kernel(const double *__restrict__ v, uint64_t n, int level, unsigned long long p0, unsigned long long p1,
                                                          unsigned long long p2, unsigned long long p3, unsigned *__restrict__ ghist)
{
    __shared__ unsigned h[4][SEL_BINS];
    const int t = threadIdx.x, lane = t & 63;
    for (int i = t; i < 4 * SEL_BINS; i += 256) (&h[0][0])[i] = 0;
    __syncthreads();
    const unsigned long long pfx[4] = {p0, p1, p2, p3};
    int owner[4];
    for (int q = 0; q < 4; ++q) {
        owner[q] = q;
        for (int p = 0; p < q; ++p)
            if (level == 0 || pfx[p] == pfx[q]) {
                owner[q] = p;
                break;
            }
    }
    const int shift = sel_shift(level), width = sel_width(level);
    const unsigned dmask = (1u << width) - 1;
    const uint64_t stride = (uint64_t)gridDim.x * 256ull;
    const uint64_t nround = (n + stride - 1) / stride;
    for (uint64_t it = 0; it < nround; ++it) {
        const uint64_t i = it * stride + blockIdx.x * 256ull + t;
        const bool valid = i < n;
        const unsigned long long key = valid ? f64_key(v[i]) : 0ull;
        const unsigned digit = (unsigned)(key >> shift) & dmask;
        const unsigned long long hi = level == 0 ? 0ull : (key >> (shift + width));
#pragma unroll
        for (int q = 0; q < 4; ++q)
            if (owner[q] == q) sel_count(h[q], digit, valid && (level == 0 || hi == pfx[q]), lane);
    }
    __syncthreads();
    for (int q = 0; q < 4; ++q)        // aliases get a copy so that the host can treat the 4 queries independently
        for (int i = t; i < SEL_BINS; i += 256) {
            const unsigned c = h[owner[q]][i];
            if (c) atomicAdd(&ghist[q * SEL_BINS + i], c);
        }
}

int wfx_dev_select_level(wfx_ctx *ctx, const double *env, uint64_t n, int level, const uint64_t prefix[4], unsigned *hist)
{
    if (level < 0 || level >= SEL_LEVELS) return wfx_fail(ctx, WFX_ERR_BAD_ARG, "select level %d out of range", level);
    WFX_LAUNCH(ctx, K_SELECT_HIST, select_level_kernel, dim3(wfx_stream_grid(n, 4096)), dim3(256), env, n, level, (unsigned long long)prefix[0],
               (unsigned long long)prefix[1], (unsigned long long)prefix[2], (unsigned long long)prefix[3], hist);
    return 0;
}

// stand-alone level 0 (used when the envelope was not produced by a kernel that fuses it)
__global__ void __launch_bounds__(256) select_l0_kernel(const double *__restrict__ v, uint64_t n, unsigned *__restrict__ ws)
{
    __shared__ unsigned h[SEL_BINS];
    const int t = threadIdx.x, lane = t & 63;
    for (int i = t; i < SEL_BINS; i += 256) h[i] = 0;
    __syncthreads();
    const uint64_t stride = (uint64_t)gridDim.x * 256ull;
    const uint64_t nround = (n + stride - 1) / stride;
    for (uint64_t it = 0; it < nround; ++it) {
        const uint64_t i = it * stride + blockIdx.x * 256ull + t;
        const bool valid = i < n;
        const unsigned long long key = valid ? f64_key(v[i]) : 0ull;
        sel_count(h, (unsigned)(key >> 53), valid, lane);
    }
    __syncthreads();
    for (int i = t; i < SEL_BINS; i += 256)
        if (h[i]) atomicAdd(&ws[SEL_H0 + i], h[i]);
}

// Streams v[0..n) through fn(value, valid) for a grid of 256-thread blocks, 4 values (32 bytes) per lane
// and step.  SEL_UN steps are loaded back to back before any of them is consumed: bytes in flight,
// not occupancy, hide the HBM latency of these one-pass kernels.
#define SEL_UN 4
template <int NT = 256, typename FN>
__device__ __forceinline__ void sel_stream(const double *__restrict__ v, uint64_t n, FN fn, int nt)
{
    const int t = threadIdx.x;
    const uint64_t quads = (n + 3) / 4;
    const uint64_t stride = (uint64_t)gridDim.x * (uint64_t)NT;
    const uint64_t nround = (quads + stride - 1) / stride;
    uint64_t it = 0;
    for (; it + SEL_UN <= nround && (it + SEL_UN) * stride * 4 <= n; it += SEL_UN) {      // whole batch inside the array: no guards
        double2 lo[SEL_UN], hi[SEL_UN];
#pragma unroll
        for (int u = 0; u < SEL_UN; ++u) {
            const uint64_t i0 = ((it + u) * stride + blockIdx.x * (uint64_t)NT + t) * 4;
            lo[u] = wfx_ld((const double2 *)(v + i0), nt);
            hi[u] = wfx_ld((const double2 *)(v + i0 + 2), nt);
        }
#pragma unroll
        for (int u = 0; u < SEL_UN; ++u) {
            fn(lo[u].x, true);
            fn(lo[u].y, true);
            fn(hi[u].x, true);
            fn(hi[u].y, true);
        }
    }
    for (; it < nround; ++it) {
        const uint64_t i0 = (it * stride + blockIdx.x * (uint64_t)NT + t) * 4;
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
        fn(a0, i0 < n);
        fn(a1, i0 + 1 < n);
        fn(a2, i0 + 2 < n);
        fn(a3, i0 + 3 < n);
    }
}

// the same walk in two phases per batch: every value of the batch goes through `first`, then `between` runs once, then every
// value goes through `second` (the compaction counts a lane's matches, reserves list space for the whole wave with ONE atomic,
// then writes -- an atomic per element slot and wave made the pass latency-bound on captures whose bins are full)
template <int NT, typename F1, typename FB, typename F2>
__device__ __forceinline__ void sel_stream2(const double *__restrict__ v, uint64_t n, F1 first, FB between, F2 second, int nt)
{
    const int t = threadIdx.x;
    const uint64_t quads = (n + 3) / 4;
    const uint64_t stride = (uint64_t)gridDim.x * (uint64_t)NT;
    const uint64_t nround = (quads + stride - 1) / stride;
    uint64_t it = 0;
    for (; it + SEL_UN <= nround && (it + SEL_UN) * stride * 4 <= n; it += SEL_UN) {
        double2 lo[SEL_UN], hi[SEL_UN];
#pragma unroll
        for (int u = 0; u < SEL_UN; ++u) {
            const uint64_t i0 = ((it + u) * stride + blockIdx.x * (uint64_t)NT + t) * 4;
            lo[u] = wfx_ld((const double2 *)(v + i0), nt);
            hi[u] = wfx_ld((const double2 *)(v + i0 + 2), nt);
        }
#pragma unroll
        for (int u = 0; u < SEL_UN; ++u) {
            first(lo[u].x, true);
            first(lo[u].y, true);
            first(hi[u].x, true);
            first(hi[u].y, true);
        }
        between();
#pragma unroll
        for (int u = 0; u < SEL_UN; ++u) {
            second(lo[u].x, true);
            second(lo[u].y, true);
            second(hi[u].x, true);
            second(hi[u].y, true);
        }
    }
    for (; it < nround; ++it) {
        const uint64_t i0 = (it * stride + blockIdx.x * (uint64_t)NT + t) * 4;
        double a[4] = {0, 0, 0, 0};
        if (i0 + 4 <= n) {
            const double2 lo = *(const double2 *)(v + i0), hi2 = *(const double2 *)(v + i0 + 2);
            a[0] = lo.x;
            a[1] = lo.y;
            a[2] = hi2.x;
            a[3] = hi2.y;
        } else {
            if (i0 < n) a[0] = v[i0];
            if (i0 + 1 < n) a[1] = v[i0 + 1];
            if (i0 + 2 < n) a[2] = v[i0 + 2];
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) first(a[k], i0 + k < n);
        between();
#pragma unroll
        for (int k = 0; k < 4; ++k) second(a[k], i0 + k < n);
    }
}

// level 1: choose the level-0 digit of every query, then histogram bits 52..42 of the
// values whose top 11 bits match
template <int NT>
__global__ void __launch_bounds__(NT) select_l1_kernel(const double *__restrict__ v, uint64_t n, uint64_t r0, uint64_t r1, uint64_t r2, uint64_t r3,
                                                       unsigned *__restrict__ ws, wfx_dev_scalars *__restrict__ s, int nt)
{
    __shared__ unsigned h[4][SEL_BINS];
    __shared__ unsigned long long pfx[4], rnk[4], rin[4];
    __shared__ unsigned dig[4];
    __shared__ int owner[4];
    const int t = threadIdx.x;
    for (int i = t; i < 4 * SEL_BINS; i += NT) (&h[0][0])[i] = 0;
    if (t == 0) {
        rin[0] = r0;
        rin[1] = r1;
        rin[2] = r2;
        rin[3] = r3;
        owner[0] = owner[1] = owner[2] = owner[3] = 0;
    }
    __syncthreads();
    sel_pick_digits(ws + SEL_H0, true, SEL_BINS, owner, rin, dig, rnk);
    __syncthreads();
    if (t == 0) {
        for (int q = 0; q < 4; ++q) pfx[q] = dig[q];
        sel_owners(pfx, owner);
        if (blockIdx.x == 0)
            for (int q = 0; q < 4; ++q) {
                s->sel_prefix[q] = pfx[q];
                s->sel_rank[q] = rnk[q];
            }
    }
    __syncthreads();
    bool active[4];
    unsigned long long mypfx[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        active[q] = owner[q] == q;
        mypfx[q] = pfx[q];
    }
    auto count = [&](double x, bool valid) {
        const unsigned long long key = f64_key(x);
        const unsigned digit = (unsigned)(key >> 42) & (SEL_BINS - 1);
        const unsigned long long hi = key >> 53;
#pragma unroll
        for (int q = 0; q < 4; ++q)
            if (active[q] && valid && hi == mypfx[q]) atomicAdd(&h[q][digit], 1u);   // level-1 digits are diverse: no wave aggregation
    };
    sel_stream<NT>(v, n, count, nt);
    __syncthreads();
    for (int i = t; i < 4 * SEL_BINS; i += NT) {
        const unsigned c = (&h[0][0])[i];
        if (c) atomicAdd(&ws[SEL_H1 + i], c);
    }
}

// compaction: choose the level-1 digit of every query, then append the keys whose top
// 22 bits match to the query's candidate list (one atomic per wave and list)
template <int NT>
__global__ void __launch_bounds__(NT) select_compact_kernel(const double *__restrict__ v, uint64_t n, unsigned *__restrict__ ws,
                                                            wfx_dev_scalars *__restrict__ s, unsigned long long *__restrict__ cand, uint64_t cap, int nt)
{
    __shared__ unsigned long long pfx[4], rnk[4], rin[4];
    __shared__ unsigned dig[4];
    __shared__ int owner[4];
    const int t = threadIdx.x, lane = t & 63;
    if (t == 0) {
        for (int q = 0; q < 4; ++q) {
            pfx[q] = s->sel_prefix[q];
            rin[q] = s->sel_rank[q];
        }
        sel_owners(pfx, owner);
    }
    __syncthreads();
    sel_pick_digits(ws + SEL_H1, false, SEL_BINS, owner, rin, dig, rnk);
    __syncthreads();
    if (t == 0) {
        for (int q = 0; q < 4; ++q) pfx[q] = (pfx[q] << SEL_BITS) | dig[q];
        sel_owners(pfx, owner);
    }
    __syncthreads();
    // every block has read s->sel_* above before block 0 can overwrite it?  No: other blocks may
    // start later, so the 22-bit state goes to separate fields (sel_value is reused as scratch
    // for the ranks until the finish kernel overwrites it with the results).
    if (blockIdx.x == 0 && t == 0)
        for (int q = 0; q < 4; ++q) {
            s->sel_prefix2[q] = pfx[q];
            s->sel_rank2[q] = rnk[q];
        }
    bool active[4];
    unsigned long long mypfx[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        active[q] = owner[q] == q;
        mypfx[q] = pfx[q];
    }
    unsigned cnt[4] = {0, 0, 0, 0}, off[4] = {0, 0, 0, 0};
    auto count = [&](double x, bool valid) {
        const unsigned long long hi = f64_key(x) >> 42;
#pragma unroll
        for (int q = 0; q < 4; ++q)
            if (active[q] && valid && hi == mypfx[q]) ++cnt[q];
    };
    auto reserve = [&]() {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            if (!active[q]) continue;
            if (__ballot(cnt[q] != 0) == 0) continue;                  // nothing in this wave (the common case on noisy captures)
            unsigned incl = cnt[q];
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {
                const unsigned o = __shfl_up(incl, d);
                if (lane >= d) incl += o;
            }
            unsigned base = 0;
            if (lane == 63) base = atomicAdd(&ws[SEL_CNT + q], incl);   // one atomic per wave, list and batch
            base = (unsigned)__builtin_amdgcn_readlane((int)base, 63);
            off[q] = base + incl - cnt[q];
            cnt[q] = 0;
        }
    };
    auto append = [&](double x, bool valid) {
        const unsigned long long key = f64_key(x);
        const unsigned long long hi = key >> 42;
#pragma unroll
        for (int q = 0; q < 4; ++q)
            if (active[q] && valid && hi == mypfx[q]) {
                const uint64_t slot = off[q]++;
                if (slot < cap) cand[(uint64_t)q * cap + slot] = key;
            }
    };
    sel_stream2<NT>(v, n, count, reserve, append, nt);
}

// levels 2..5 on the candidate lists, numpy's lerp, and clearing of the workspace
__global__ void __launch_bounds__(1024) select_finish_kernel(unsigned *__restrict__ ws, wfx_dev_scalars *__restrict__ s,
                                                            const unsigned long long *__restrict__ cand, uint64_t cap, int do_lerp, double gamma_lo,
                                                            double gamma_hi)
{
    __shared__ unsigned h[SEL_BINS];
    __shared__ unsigned wave_tot[16];
    __shared__ unsigned long long pfx[4], rnk[4];
    __shared__ int owner[4];
    __shared__ unsigned long long cur_p, cur_r, cur_full;
    __shared__ unsigned cur_c, few_n;
    __shared__ unsigned long long few[64];
    bool done_direct = false;
    const int t = threadIdx.x, lane = t & 63;
    if (t == 0) {
        for (int q = 0; q < 4; ++q) {
            pfx[q] = s->sel_prefix2[q];
            rnk[q] = s->sel_rank2[q];
        }
        sel_owners(pfx, owner);
    }
    __syncthreads();
    {
        const int q = blockIdx.x;               // one workgroup per query
        const int o = owner[q];
        const uint64_t cnt = min((uint64_t)ws[SEL_CNT + o], cap);
        const unsigned long long *list = cand + (uint64_t)o * cap;
        if (t == 0) {
            cur_p = pfx[q];
            cur_r = rnk[q];
        }
        __syncthreads();
        for (int level = 2; level < SEL_LEVELS; ++level) {
            const int shift = sel_shift(level), width = sel_width(level);
            const unsigned dmask = (1u << width) - 1;
            const unsigned long long want = cur_p;
            for (int i = t; i < SEL_BINS; i += 1024) h[i] = 0;
            __syncthreads();
            const uint64_t nround = (cnt + 1023) / 1024;
            for (uint64_t it = 0; it < nround; ++it) {
                const uint64_t i = it * 1024 + t;
                const bool valid = i < cnt;
                const unsigned long long key = valid ? list[i] : 0ull;
                sel_count(h, (unsigned)(key >> shift) & dmask, valid && (key >> (shift + width)) == want, lane);
            }
            __syncthreads();
            // thread t owns bins 2t, 2t+1
            const unsigned c0 = h[2 * t], c1 = h[2 * t + 1];
            unsigned long long cum = block_excl_scan<1024>(c0 + c1, wave_tot, nullptr);
            const unsigned long long r = cur_r;
            __syncthreads();
            if (r >= cum && r < cum + c0) {
                cur_p = (want << width) | (unsigned long long)(2 * t);
                cur_r = r - cum;
                cur_c = c0;
            } else if (r >= cum + c0 && r < cum + c0 + c1) {
                cur_p = (want << width) | (unsigned long long)(2 * t + 1);
                cur_r = r - cum - c0;
                cur_c = c1;
            }
            __syncthreads();
            // A level leaves a handful of keys in the chosen bin (a few thousand candidates over 2048 bins): finish
            // those directly -- gather them, rank each against the others -- instead of three more histogram levels
            // (six workgroup barriers each).
            if (cur_c <= 64 && level + 1 < SEL_LEVELS) {
                const unsigned long long wantk = cur_p;
                const int sh = shift;
                if (t == 0) few_n = 0;
                __syncthreads();
                const uint64_t nround = (cnt + 1023) / 1024;
                for (uint64_t it = 0; it < nround; ++it) {
                    const uint64_t i = it * 1024 + t;
                    if (i < cnt) {
                        const unsigned long long key = list[i];
                        if ((key >> sh) == wantk) few[atomicAdd(&few_n, 1u)] = key;
                    }
                }
                __syncthreads();
                const unsigned m = few_n;                               // == cur_c
                if (t < (int)m) {
                    const unsigned long long mine = few[t];
                    unsigned below = 0, equal_before = 0;
                    for (unsigned j = 0; j < m; ++j) {
                        const unsigned long long o2 = few[j];
                        below += o2 < mine;
                        equal_before += (o2 == mine) & (j < (unsigned)t);
                    }
                    if (below + equal_before == (unsigned)cur_r) cur_full = mine;      // exactly one lane
                }
                __syncthreads();
                done_direct = true;
                break;
            }
        }
        if (t == 0) s->sel_value[q] = key_f64(done_direct ? cur_full : cur_p);
    }
    // The workgroup that finishes last applies numpy's interpolation between the two order statistics of
    // each percentile and clears the workspace for the next run (a separate 1-workgroup launch cost 4.5 us).
    // Hand-off: result store -> release fence -> ticket; the last one: acquire fence -> sc1 loads.
    __shared__ int is_last;
    if (t == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const unsigned ticket = __hip_atomic_fetch_add(&ws[SEL_TICKET], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        is_last = ticket == gridDim.x - 1;
        if (is_last) {
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (do_lerp) {
                double v[4];
                for (int q = 0; q < 4; ++q) v[q] = __hip_atomic_load(&s->sel_value[q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                s->low = np_lerp(v[0], v[1], gamma_lo);
                s->high = np_lerp(v[2], v[3], gamma_hi);
                s->nan_count = 0;
            }
        }
    }
    __syncthreads();
    if (is_last)
        for (int i = t; i < SEL_WORDS; i += 1024) ws[i] = 0;
}

// workspace + candidate lists; the workspace is zeroed when it is (re)allocated and by
// the finish kernel after every use
static int select_prepare(wfx_ctx *ctx, uint64_t n, unsigned **ws, unsigned long long **cand)
{
    const bool fresh = ctx->b_hist.cap < SEL_WORDS * sizeof(unsigned) || !ctx->b_hist.p;
    WFX_TRY(wfx_reserve(ctx, ctx->b_hist, SEL_WORDS * sizeof(unsigned)));
    if (fresh) WFX_HIP(ctx, hipMemsetAsync(ctx->b_hist.p, 0, SEL_WORDS * sizeof(unsigned), ctx->stream));
    WFX_TRY(wfx_reserve(ctx, ctx->b_cand, 4 * (size_t)n * sizeof(unsigned long long)));
    *ws = (unsigned *)ctx->b_hist.p;
    *cand = (unsigned long long *)ctx->b_cand.p;
    return 0;
}

// levels 1.. given that the level-0 histogram is already in the workspace
static int select_after_l0(wfx_ctx *ctx, const double *env, uint64_t n, const uint64_t ranks[4], wfx_dev_scalars *d_scal, int do_lerp,
                           double gamma_lo, double gamma_hi)
{
    unsigned *ws = (unsigned *)ctx->b_hist.p;
    unsigned long long *cand = (unsigned long long *)ctx->b_cand.p;
    // level 1: one 1024-lane workgroup per CU -- the flush of a workgroup's 2 x 2048 live bins is 64 atomic
    // wave-instructions that execute at the memory side (~1.3 TB/s chip-wide), so fewer, larger workgroups
    // (18.5 us) beat 512 x 256 lanes (28 us)
    const unsigned g1 = std::min(wfx_stream_grid(n, 16384), 256u);
    WFX_LAUNCH(ctx, K_SELECT_HIST, select_l1_kernel<1024>, dim3(g1), dim3(1024), env, n, ranks[0], ranks[1], ranks[2], ranks[3], ws, d_scal, wfx_nt_for(8.0 * (double)n));
    WFX_LAUNCH(ctx, K_SELECT_HIST, select_compact_kernel<1024>, dim3(g1), dim3(1024), env, n, ws, d_scal, cand, n, wfx_nt_for(8.0 * (double)n));
    WFX_LAUNCH(ctx, K_SELECT_SCAN, select_finish_kernel, dim3(4), dim3(1024), ws, d_scal, (const unsigned long long *)cand, n, do_lerp, gamma_lo,
               gamma_hi);
    return 0;
}

static int check_ranks(wfx_ctx *ctx, uint64_t n, const uint64_t ranks[4])
{
    for (int q = 0; q < 4; ++q)
        if (ranks[q] >= n) return wfx_fail(ctx, WFX_ERR_BAD_ARG, "order statistic rank %llu >= n", (unsigned long long)ranks[q]);
    return 0;
}

int wfx_dev_select(wfx_ctx *ctx, const double *env, uint64_t n, const uint64_t ranks[4], wfx_dev_scalars *d_scal)
{
    WFX_TRY(check_ranks(ctx, n, ranks));
    unsigned *ws;
    unsigned long long *cand;
    WFX_TRY(select_prepare(ctx, n, &ws, &cand));
    WFX_LAUNCH(ctx, K_SELECT_HIST, select_l0_kernel, dim3(wfx_stream_grid(n, 2048)), dim3(256), env, n, ws);
    return select_after_l0(ctx, env, n, ranks, d_scal, 0, 0.0, 0.0);
}

// percentiles of an envelope whose level-0 histogram was NOT fused into its producer
int wfx_dev_percentiles(wfx_ctx *ctx, const double *env, uint64_t n, const uint64_t ranks[4], double gamma_lo, double gamma_hi,
                        wfx_dev_scalars *d_scal)
{
    WFX_TRY(check_ranks(ctx, n, ranks));
    unsigned *ws;
    unsigned long long *cand;
    WFX_TRY(select_prepare(ctx, n, &ws, &cand));
    WFX_LAUNCH(ctx, K_SELECT_HIST, select_l0_kernel, dim3(wfx_stream_grid(n, 2048)), dim3(256), env, n, ws);
    return select_after_l0(ctx, env, n, ranks, d_scal, 1, gamma_lo, gamma_hi);
}

// call BEFORE the kernel that fuses the level-0 histogram: returns the workspace pointer
int wfx_dev_select_workspace(wfx_ctx *ctx, uint64_t n, unsigned **ws)
{
    unsigned long long *cand;
    return select_prepare(ctx, n, ws, &cand);
}

// percentiles when the level-0 histogram has already been accumulated into the workspace
int wfx_dev_percentiles_fused(wfx_ctx *ctx, const double *env, uint64_t n, const uint64_t ranks[4], double gamma_lo, double gamma_hi,
                              wfx_dev_scalars *d_scal)
{
    WFX_TRY(check_ranks(ctx, n, ranks));
    return select_after_l0(ctx, env, n, ranks, d_scal, 1, gamma_lo, gamma_hi);
}

// ---- the same select in the steps a sharded decode separates with collectives (wfx_shard.hip) ----------------
// level-0 histogram (fused into the envelope kernel) -> all-reduce -> level 1 -> all-reduce -> compaction into a fixed-size
// block per rank -> all-gather of the blocks -> merge + finish on every rank (identical inputs, identical results).
int wfx_dev_select_sharded_ws(wfx_ctx *ctx, unsigned **ws)
{
    const bool fresh = ctx->b_hist.cap < SEL_WORDS * sizeof(unsigned) || !ctx->b_hist.p;
    WFX_TRY(wfx_reserve(ctx, ctx->b_hist, SEL_WORDS * sizeof(unsigned)));
    if (fresh) WFX_HIP(ctx, hipMemsetAsync(ctx->b_hist.p, 0, SEL_WORDS * sizeof(unsigned), ctx->stream));
    *ws = (unsigned *)ctx->b_hist.p;
    return 0;
}

int wfx_dev_select_l1(wfx_ctx *ctx, const double *env, uint64_t n, const uint64_t ranks[4], unsigned *ws, wfx_dev_scalars *d_scal)
{
    const unsigned g1 = std::min(wfx_stream_grid(n, 16384), 256u);
    WFX_LAUNCH(ctx, K_SELECT_HIST, select_l1_kernel<1024>, dim3(g1), dim3(1024), env, n, ranks[0], ranks[1], ranks[2], ranks[3], ws, d_scal, wfx_nt_for(8.0 * (double)n));
    return 0;
}

__global__ void select_block_counts_kernel(const unsigned *__restrict__ ws, unsigned *__restrict__ counts)
{
    if (threadIdx.x < 4) counts[threadIdx.x] = ws[SEL_CNT + threadIdx.x];
}

int wfx_dev_select_compact_block(wfx_ctx *ctx, const double *env, uint64_t n, unsigned *ws, wfx_dev_scalars *d_scal, void *block, uint64_t cap)
{
    const unsigned g1 = std::min(wfx_stream_grid(n, 16384), 256u);
    WFX_LAUNCH(ctx, K_SELECT_HIST, select_compact_kernel<1024>, dim3(g1), dim3(1024), env, n, ws, d_scal, (unsigned long long *)block, cap, wfx_nt_for(8.0 * (double)n));
    WFX_LAUNCH(ctx, K_SELECT_SCAN, select_block_counts_kernel, dim3(1), dim3(64), (const unsigned *)ws, (unsigned *)((char *)block + cap * 32));
    return 0;
}

// blocks[r] = {u64 keys[4][cap]; u32 count[4]; u32 pad[4]} -> merged[q][capm], ws[SEL_CNT + q] = total
__global__ void __launch_bounds__(256) select_merge_blocks_kernel(const char *__restrict__ blocks, int nblocks, uint64_t cap, unsigned long long *__restrict__ merged,
                                                                 uint64_t capm, unsigned *__restrict__ ws, unsigned *__restrict__ overflow)
{
    const int r = blockIdx.y, q = blockIdx.z;
    const size_t bb = (size_t)cap * 32 + 32;
    uint64_t off = 0, total = 0;
    bool over = false;
    for (int k = 0; k < nblocks; ++k) {
        const unsigned c = ((const unsigned *)(blocks + (size_t)k * bb + cap * 32))[q];
        const uint64_t cc = c < cap ? c : cap;
        over = over || c > cap;
        if (k < r) off += cc;
        total += cc;
    }
    const unsigned mine = ((const unsigned *)(blocks + (size_t)r * bb + cap * 32))[q];
    const uint64_t cnt = mine < cap ? mine : cap;
    const unsigned long long *src = (const unsigned long long *)(blocks + (size_t)r * bb) + (size_t)q * cap;
    for (uint64_t i = blockIdx.x * 256ull + threadIdx.x; i < cnt; i += (uint64_t)gridDim.x * 256ull) merged[(size_t)q * capm + off + i] = src[i];
    if (r == 0 && blockIdx.x == 0 && threadIdx.x == 0) {
        ws[SEL_CNT + q] = (unsigned)total;
        if (over && q == 0) atomicAdd(overflow, 1u);
        else if (over) atomicAdd(overflow, 1u);
    }
}

int wfx_dev_select_finish_blocks(wfx_ctx *ctx, unsigned *ws, wfx_dev_scalars *d_scal, const void *blocks, int nblocks, uint64_t cap, double gamma_lo,
                                 double gamma_hi, unsigned *overflow)
{
    const uint64_t capm = cap * (uint64_t)nblocks;
    WFX_TRY(wfx_reserve(ctx, ctx->b_cand, 4 * (size_t)capm * sizeof(unsigned long long)));
    unsigned long long *merged = (unsigned long long *)ctx->b_cand.p;
    unsigned gx = (unsigned)((cap + 2047) / 2048);
    if (gx > 64) gx = 64;
    WFX_LAUNCH(ctx, K_SELECT_SCAN, select_merge_blocks_kernel, dim3(gx, nblocks, 4), dim3(256), (const char *)blocks, nblocks, cap, merged, capm, ws, overflow);
    WFX_LAUNCH(ctx, K_SELECT_SCAN, select_finish_kernel, dim3(4), dim3(1024), ws, d_scal, (const unsigned long long *)merged, capm, 1, gamma_lo, gamma_hi);
    return 0;
}

__global__ void add_u64_kernel(unsigned long long *dst, const unsigned long long *src, int n)
{
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        unsigned long long s = 0;
        for (int i = 0; i < n; ++i) s += src[i];
        *dst += s;
    }
}

int wfx_dev_add_u64(wfx_ctx *ctx, unsigned long long *dst, const unsigned long long *src, int n)
{
    WFX_LAUNCH(ctx, K_SELECT_SCAN, add_u64_kernel, dim3(1), dim3(64), dst, src, n);
    return 0;
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
// The IEEE division was ~35 of the ~70 VALU instructions a sample cost in the fused kernel.  q' = a * (1/delta)
// is within 2 ulp of the true quotient, so rint(q') == rint(a / delta) unless q' lies within a few ulp of a
// half-integer; only then (|frac - 0.5| < 1e-9: practically never) is the division performed.  Out-of-range
// quotients clamp to the same end either way.  The reciprocal is used only when it is a normal finite number.
struct quant_par {
    double low, delta, rcp;
    bool fast;
};
__device__ __forceinline__ quant_par quant_make(double low, double delta)
{
    quant_par p;
    p.low = low;
    p.delta = delta;
    p.rcp = 1.0 / delta;
    const double ar = fabs(p.rcp);
    p.fast = ar > 1e-290 && ar < 1e290;
    return p;
}
__device__ __forceinline__ unsigned quantise_one(double v, const quant_par &p, unsigned &nan)
{
    const double a = 255 * (v - p.low);     // same operation order as numpy
    double q = a * p.rcp;
    if (!p.fast || fabs((q - floor(q)) - 0.5) < 1e-9) q = a / p.delta;      // IEEE division, as numpy
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
                                                      uint8_t *__restrict__ out, wfx_dev_scalars *__restrict__ sout, double eps)
{
    const double low = s->low, high = s->high;
    const double delta = (high - low) + eps;        // eps = 0: wefax.py:197; 1e-6: the live path's guard (data_packet.py:461)
    const quant_par qp = quant_make(low, delta);
    unsigned nan = 0;
    const uint64_t groups = (n + 7) / 8;
    for (uint64_t gi = blockIdx.x * 256ull + threadIdx.x; gi < groups; gi += (uint64_t)gridDim.x * 256ull) {
        const uint64_t i0 = gi * 8;
        if (i0 + 8 <= n) {
            const double2 *p = (const double2 *)(env + i0);
            const double2 a = p[0], b = p[1], c = p[2], d = p[3];
            unsigned w0 = quantise_one(a.x, qp, nan) | (quantise_one(a.y, qp, nan) << 8) |
                          (quantise_one(b.x, qp, nan) << 16) | (quantise_one(b.y, qp, nan) << 24);
            unsigned w1 = quantise_one(c.x, qp, nan) | (quantise_one(c.y, qp, nan) << 8) |
                          (quantise_one(d.x, qp, nan) << 16) | (quantise_one(d.y, qp, nan) << 24);
            *(uint2 *)(out + i0) = make_uint2(w0, w1);
        } else {
            for (uint64_t i = i0; i < n; ++i) out[i] = (uint8_t)quantise_one(env[i], qp, nan);
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

int wfx_dev_quantise(wfx_ctx *ctx, const double *env, uint64_t n, const wfx_dev_scalars *d_scal, uint8_t *out, wfx_dev_scalars *d_scal_out,
                     double eps)
{
    WFX_LAUNCH(ctx, K_QUANTISE, quantise_kernel, dim3(wfx_stream_grid((n + 7) / 8, 256)), dim3(256), env, n, d_scal, out, d_scal_out, eps);
    return 0;
}

// 3-tap median with zeros beyond both ends: scipy.signal.medfilt(x, 3) of the live path (data_packet.py:446)
__global__ void __launch_bounds__(256) median3_kernel(const double *__restrict__ r, uint64_t n, double *__restrict__ out)
{
    for (uint64_t i = blockIdx.x * 256ull + threadIdx.x; i < n; i += (uint64_t)gridDim.x * 256ull) {
        const double a = i > 0 ? r[i - 1] : 0.0, b = r[i], c = i + 1 < n ? r[i + 1] : 0.0;
        out[i] = fmax(fmin(a, b), fmin(fmax(a, b), c));
    }
}

int wfx_dev_median3(wfx_ctx *ctx, const double *env_raw, uint64_t n, double *env)
{
    WFX_LAUNCH(ctx, K_MEDIAN, median3_kernel, dim3(wfx_stream_grid(n, 256)), dim3(256), env_raw, n, env);
    return 0;
}

// ===========================================================================
// a9  sync search (wefax.py:218-294)
// corr[i] = -127 * sum_{k<L}(d[i+k]-128) - sum_{k in middle run}(d[i+k]-128)
// ===========================================================================
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

// the same reductions over each 32-lane half of the wave (lanes 0..31 -> lo, 32..63 -> hi): five DPP steps,
// the last one (row_bcast15 into rows 1 and 3) leaves the results in lanes 31 and 63
// (written as v_max/v_min with the DPP modifier on the instruction itself: the builtin form compiles to mov + mov_dpp + op)
#define WFX_DPP_RED(OP, v)                                                                                    \
    asm volatile("s_nop 1\n\t" OP " %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"           \
                 "s_nop 1\n\t" OP " %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"           \
                 "s_nop 1\n\t" OP " %0, %0, %0 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"               \
                 "s_nop 1\n\t" OP " %0, %0, %0 row_mirror row_mask:0xf bank_mask:0xf\n\t"                    \
                 "s_nop 1\n\t" OP " %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"                  \
                 "s_nop 1"                                                                                    \
                 : "+v"(v))
__device__ __forceinline__ void half_max_i32(int v, int &lo, int &hi)
{
    WFX_DPP_RED("v_max_i32_dpp", v);
    lo = __builtin_amdgcn_readlane(v, 31);
    hi = __builtin_amdgcn_readlane(v, 63);
}
__device__ __forceinline__ void half_min_i32(int v, int &lo, int &hi)
{
    WFX_DPP_RED("v_min_i32_dpp", v);
    lo = __builtin_amdgcn_readlane(v, 31);
    hi = __builtin_amdgcn_readlane(v, 63);
}

// first arg-max over the wave: (max correlation, smallest index holding it)
__device__ __forceinline__ void wave_first_argmax(int c, int idx, int &best_c, int &best_idx)
{
    best_c = wave_max_i32(c);
    best_idx = wave_min_i32(c == best_c ? idx : 0x7fffffff);
}

// corr[i] for every position, plus (optionally) per block of 64 positions the maximum
// and the offset of its first occurrence -- what the sequential picker consumes.
//
// Both window sums are differences of a running byte sum P: the workgroup forms P over its chunk + halo in LDS
// (18 consecutive bytes per lane, wave scan by DPP, one exchange across the four waves), then every lane owns
// one position per row of 256: four conflict-free LDS reads, a coalesced 4-byte store, and per wave (= per block
// of 64 positions) a DPP max + ballot for the summaries.  No per-lane serial loop over the pattern length: the
// sliding-sum form (16 consecutive positions per lane, ~130 dependent LDS byte reads) took 27 us, this takes ~9.
#define CORR_BPT 18                                   // bytes per lane: 256 * 18 = CORR_CH + 512
__device__ __forceinline__ int wave_incl_scan_i32(int v)
{
    // row_shr:1,2,4,8 inside the rows of 16, then the row totals across rows
    int x = v;
    int y;
    y = __builtin_amdgcn_update_dpp(0, x, 0x111, 0xf, 0xf, false); x += y;     // row_shr:1
    y = __builtin_amdgcn_update_dpp(0, x, 0x112, 0xf, 0xf, false); x += y;     // row_shr:2
    y = __builtin_amdgcn_update_dpp(0, x, 0x114, 0xf, 0xf, false); x += y;     // row_shr:4
    y = __builtin_amdgcn_update_dpp(0, x, 0x118, 0xf, 0xf, false); x += y;     // row_shr:8
    y = __builtin_amdgcn_update_dpp(0, x, 0x142, 0xa, 0xf, false); x += y;     // row_bcast15 -> rows 1, 3
    y = __builtin_amdgcn_update_dpp(0, x, 0x143, 0xc, 0xf, false); x += y;     // row_bcast31 -> rows 2, 3
    return x;
}

__global__ void __launch_bounds__(256) sync_corr_kernel(const uint8_t *__restrict__ d, uint64_t n, int n1, int n0, int *__restrict__ corr,
                                                       int *__restrict__ bmax, int *__restrict__ boff)
{
    __shared__ int P[256 * CORR_BPT + 1];
    __shared__ int wave_tot[4];
    const int L = 2 * n1 + n0;
    const uint64_t ncorr = n > (uint64_t)L ? n - L : 0;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    for (uint64_t p0 = (uint64_t)blockIdx.x * CORR_CH; p0 < ncorr; p0 += (uint64_t)gridDim.x * CORR_CH) {
        // bytes [p0 + 18 t, + 18): nine 2-byte loads (p0 is a multiple of 4096, the stream is padded by 64 bytes;
        // anything at or beyond n counts as 0)
        const uint64_t b0 = p0 + (uint64_t)(CORR_BPT * t);
        unsigned short raw[CORR_BPT / 2];
#pragma unroll
        for (int k = 0; k < CORR_BPT / 2; ++k) {
            const uint64_t i = b0 + 2 * k;
            raw[k] = *(const unsigned short *)(d + (i + 1 < n + 48 ? i : 0));
        }
        int loc[CORR_BPT];
        int sum = 0;
#pragma unroll
        for (int k = 0; k < CORR_BPT; ++k) {
            const int byte = (b0 + k < n) ? (int)((raw[k >> 1] >> (8 * (k & 1))) & 0xff) : 0;
            loc[k] = sum;
            sum += byte;
        }
        const int incl = wave_incl_scan_i32(sum);
        __syncthreads();                                   // previous chunk's rows are done with P
        if (lane == 63) wave_tot[wave] = incl;
        __syncthreads();
        int offset = incl - sum;
#pragma unroll
        for (int w = 0; w < 3; ++w)
            if (w < wave) offset += wave_tot[w];
#pragma unroll
        for (int k = 0; k < CORR_BPT; ++k) P[CORR_BPT * t + k] = offset + loc[k];
        if (t == 255) P[256 * CORR_BPT] = offset + sum;
        __syncthreads();
        const int cnt = (int)min((uint64_t)CORR_CH, ncorr - p0);
        const int kall = 128 * L, kmid = 128 * n0;
#pragma unroll 4
        for (int m = 0; m < CORR_CH / 256; ++m) {
            const int j = t + 256 * m;
            if (256 * m >= cnt) break;
            const int sall = P[j + L] - P[j], smid = P[j + n1 + n0] - P[j + n1];
            const int c = -127 * (sall - kall) - (smid - kmid);
            const bool valid = j < cnt;
            if (valid) corr[p0 + j] = c;
            if (bmax) {
                const int cm = valid ? c : (int)0x80000000;
                const int best = wave_max_i32(cm);
                const unsigned long long hit = __ballot(cm == best);
                if (lane == 0 && (j & ~63) < cnt) {
                    bmax[(p0 + j) / 64] = best;
                    boff[(p0 + j) / 64] = __ffsll((long long)hit) - 1;
                }
            }
        }
    }
}

// a8 + the correlation of a9 for the decode path.  (A fused kernel -- quantise a chunk and its halo into LDS,
// correlate from there -- was measured at 34 us: every workgroup strung its loads, the quantiser and the
// correlation into one dependent chain.  Two streaming kernels take 12 + 14 us.)
int wfx_dev_quantise_corr(wfx_ctx *ctx, const double *env, uint64_t n, wfx_dev_scalars *d_scal, uint8_t *out, int n1, int n0)
{
    if (2 * n1 + n0 > 500 || n1 < 0 || n0 < 0) return wfx_fail(ctx, WFX_ERR_BAD_ARG, "sync pattern length out of range");
    const size_t nblk = (size_t)n / 64 + 2;
    WFX_TRY(wfx_reserve(ctx, ctx->b_corr, (size_t)n * 4 + 256));
    WFX_TRY(wfx_reserve(ctx, ctx->b_tmp2, nblk * 8));
    int *bmax = (int *)ctx->b_tmp2.p, *boff = bmax + nblk;
    WFX_LAUNCH(ctx, K_QUANTISE, quantise_kernel, dim3(wfx_stream_grid((n + 7) / 8, 256)), dim3(256), env, n, (const wfx_dev_scalars *)d_scal, out, d_scal,
               0.0);
    WFX_LAUNCH(ctx, K_SYNC_CORR, sync_corr_kernel, dim3(wfx_stream_grid(n, CORR_CH)), dim3(256), (const uint8_t *)out, n, n1, n0, (int *)ctx->b_corr.p, bmax,
               boff);
    return 0;
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
#define PICK_CH 32768
#define PICK_THREADS 1024
#define PICK_PER_THREAD (PICK_CH / PICK_THREADS)

__device__ __forceinline__ bool dev_ok(double frame_samples, long long x)
{
    // wefax.py:263-267: max_deviation > x > min_deviation
    return (frame_samples + 500 > (double)x) && ((double)x > frame_samples - 500);
}

// ---- the scan in segments ------------------------------------------------------------------------------------
// The scan is a state machine whose state right after a peak is appended at position i is (i, corr[i]) -- it has
// forgotten everything before.  So a scan started anywhere, from the made-up state "a peak was just appended
// here", IS the reference's scan from the first position at which both append a peak (the sync pulses pull any
// start onto the same train within a peak or two).  MODE 1 runs one such scan per workgroup over a window of
// PICK_CH positions starting every PICK_SEG (all CUs at once, ~7 us); the MODE 0 kernel then joins the lists:
// segment k is cut at its last appended position that segment k + 1 also appended at, and continues there.
// Peaks, first positions, count and the 100-peak stop are the reference's by construction; when a join is
// missing (or the segments do not reach the 100th peak or the end of the data) the MODE 0 kernel simply runs
// the sequential scan as before.
#define PICK_SEG 16384
#define PICK_SEG_MAX 128                 // segments joined by one wave, two per lane
#define PICK_SEG_ENT 24                  // list entries per segment (a window of 32768 holds 14 peaks at 240 LPM, 8 at 120)
struct pick_seg {
    int n;                               // entries: completed peaks, then (tail != 0) one appended but not completed
    int tail, end, overflow;             // end: the window reached the end of the data (every entry is final)
    long long first[PICK_SEG_ENT];       // position at which the peak was appended
    long long fin[PICK_SEG_ENT];         // its final position (undefined for the tail entry)
};

template <int MODE>
__global__ void __launch_bounds__(PICK_THREADS) sync_pick_kernel(const int *__restrict__ corr, const int *__restrict__ bmax, const int *__restrict__ boff,
                                                                 uint64_t n, int n1, int n0, long long mind, double frame_samples, int width,
                                                                 wfx_dev_scalars *__restrict__ s, pick_seg *__restrict__ seg, int nseg)
{
    __shared__ __attribute__((aligned(16))) int cs[PICK_CH + 64];
    __shared__ int2 sm2[PICK_CH / 64 + 192];                  // per 64-block: (max correlation, first index of it)
    __shared__ long long pk_s[WFX_MAX_PEAKS + 1], first_s[WFX_MAX_PEAKS + 1];
    __shared__ int done_flag, np_s, hit_s;
    __shared__ int ok_s[WFX_MAX_PEAKS + 1];
    const int t = threadIdx.x;
    const int lane = t & 63;
    const int L = 2 * n1 + n0;
    const uint64_t ncorr = n > (uint64_t)L ? n - L : 0;
    const uint64_t nblk = (ncorr + 63) / 64;
    const int CMIN = (int)0x80000000;
    // picker state (meaningful in wave 0, kept uniform across its lanes)
    long long pos = 0;
    int val = 0, np = 1, hit = 0;
#ifdef WFX_PICK_STATS      // diagnostic build only (python -m wefax_amd.build --pick-stats): cycle stamps per phase
    long long n_ops = 0, n_chunks = 0, t_pick = 0, t_all = 0, t_rd = 0, t_red = 0, n_spec = 0;
    const long long t_begin = (long long)__builtin_amdgcn_s_memtime();
#define PICK_STAT(x) x
#else
#define PICK_STAT(x)
#endif
    if (t == 0) {
        done_flag = 0;
        first_s[0] = 0;
    }
    // ---- MODE 0: join the segment scans (see above); `joined` != 0 replaces the sequential scan below ----
    __shared__ int join_a[PICK_SEG_MAX], join_b[PICK_SEG_MAX], join_off[PICK_SEG_MAX], join_res[PICK_SEG_MAX];
    __shared__ int join_ok, join_u, join_np, join_hit, join_known;
    int joined = 0;
    if (MODE == 0 && seg != nullptr && nseg > 0 && nseg <= PICK_SEG_MAX) {
        // the lists (272 bytes per segment) are read many times: bring them into LDS first (the chunk buffer is idle)
        const pick_seg *gseg = seg;
        pick_seg *lseg = (pick_seg *)cs;
        static_assert(sizeof(pick_seg) % 8 == 0 && PICK_SEG_MAX * sizeof(pick_seg) <= sizeof(cs), "segment lists fit the chunk buffer");
        for (int w = t; w < nseg * (int)(sizeof(pick_seg) / 8); w += PICK_THREADS) ((long long *)lseg)[w] = ((const long long *)gseg)[w];
        __syncthreads();
        seg = lseg;
        if (t < 64) {
            // phase A, a lane per pair (k, k + 1): the last position segment k appended a peak at that k + 1 also appended at
            for (int k = lane; k < nseg; k += 64) {
                int a = -1, b = -1;
                if (k + 1 < nseg) {
                    const int ek = seg[k].n + seg[k].tail, en = seg[k + 1].n + seg[k + 1].tail;
                    for (int ia = ek - 1; ia >= 0 && a < 0; --ia) {
                        const long long f = seg[k].first[ia];
                        for (int ib = 0; ib < en; ++ib)
                            if (seg[k + 1].first[ib] == f) {
                                a = ia;
                                b = ib;
                                break;
                            }
                    }
                }
                join_a[k] = a;
                join_b[k] = b;
            }
        }
        __syncthreads();
        if (t < 64) {
            // phase B: where each segment's list is entered (the index its predecessor's join points at) and left (its own
            // join), how many peaks it contributes, and the first segment whose list is used to its end (no join, end of the
            // data, or the 100th peak inside) -- two segments per lane, one wave-wide prefix sum
            int resv[2], cntv[2], leavev[2], badv[2];
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int k = lane + 64 * h;
                resv[h] = cntv[h] = leavev[h] = badv[h] = 0;
                if (k < nseg) {
                    const int res = k == 0 ? 0 : join_b[k - 1], a = join_a[k];
                    const int ent = seg[k].n + seg[k].tail;
                    const bool bad = seg[k].overflow || res < 0 || res > ent;
                    const bool leave = k + 1 < nseg && a >= 0 && a >= res && !seg[k].end && !bad;
                    resv[h] = res;
                    badv[h] = bad;
                    leavev[h] = leave;
                    cntv[h] = leave ? a - res : 0;
                }
            }
            const int inc0 = wave_incl_scan_i32(cntv[0]);
            const int tot0 = __builtin_amdgcn_readlane(inc0, 63);
            const int inc1 = wave_incl_scan_i32(cntv[1]) + tot0;
            const int exc[2] = {inc0 - cntv[0], inc1 - cntv[1]};
            // stop at k: its list is not left, or the 100th peak falls inside its stretch
            const unsigned long long st0 = __ballot(lane >= nseg || !leavev[0] || exc[0] + cntv[0] >= WFX_MAX_PEAKS);
            const unsigned long long st1 = __ballot(lane + 64 >= nseg || !leavev[1] || exc[1] + cntv[1] >= WFX_MAX_PEAKS);
            const int u = st0 ? __ffsll((long long)st0) - 1 : 64 + __ffsll((long long)st1) - 1;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int k = lane + 64 * h;
                if (k < nseg) {
                    join_res[k] = resv[h];
                    join_off[k] = exc[h];
                }
            }
            const int hu = u >> 6, lu = u & 63;
            const int res_u = __builtin_amdgcn_readlane(hu ? resv[1] : resv[0], lu), off_u = __builtin_amdgcn_readlane(hu ? exc[1] : exc[0], lu);
            const int bad_u = __builtin_amdgcn_readlane(hu ? badv[1] : badv[0], lu);
            if (lane == 0) {
                int ok = u < nseg && !bad_u;
                const int known = ok ? off_u + max(seg[u].n - res_u, 0) : 0;     // peaks with a final position
                int npj = 0, hitj = 0;
                if (ok) {
                    if (known >= WFX_MAX_PEAKS) {
                        npj = WFX_MAX_PEAKS;
                        hitj = 1;
                    } else if (known == WFX_MAX_PEAKS - 1 && seg[u].tail && res_u <= seg[u].n) {
                        npj = WFX_MAX_PEAKS;                     // the 100th peak is the one appended last: the scan stops there
                        hitj = 1;
                    } else if (seg[u].end) {
                        npj = known;                             // the data ended first
                    } else
                        ok = 0;
                }
                if (npj < 1) ok = 0;
                join_ok = ok;
                join_u = u;
                join_np = npj;
                join_hit = hitj;
                join_known = known;
            }
        }
        __syncthreads();
        joined = join_ok;
        if (joined) {
            if (t < 64) {
                // phase C, a lane per segment: its stretch of the list goes to the peak arrays
                const int u = join_u;
                for (int k = lane; k <= u; k += 64) {
                    const int res = join_res[k], off = join_off[k];
                    const int stop = k < u ? join_a[k] : seg[k].n;
                    for (int e = res; e < stop; ++e) {
                        const int g = off + (e - res);
                        if (g < WFX_MAX_PEAKS) {
                            first_s[g] = seg[k].first[e];
                            pk_s[g] = g == WFX_MAX_PEAKS - 1 ? seg[k].first[e] : seg[k].fin[e];    // the scan stops on appending the 100th
                        }
                    }
                    if (k == u && seg[k].tail && join_known == WFX_MAX_PEAKS - 1 && res <= seg[k].n) {
                        first_s[WFX_MAX_PEAKS - 1] = seg[k].first[seg[k].n];
                        pk_s[WFX_MAX_PEAKS - 1] = seg[k].first[seg[k].n];
                    }
                }
            }
            np = join_np;
            hit = join_hit;
            __syncthreads();
        }
    }
    // the correlation buffer is padded to a multiple of 4 ints beyond ncorr, so whole int4s are loaded
    int4 pre[PICK_PER_THREAD / 4];
    int pre_c = CMIN, pre_o = 0;
    auto prefetch = [&](uint64_t p0) {
#pragma unroll
        for (int k = 0; k < PICK_PER_THREAD / 4; ++k) {
            const uint64_t g = p0 + ((uint64_t)k * PICK_THREADS + t) * 4;
            pre[k] = g < ncorr ? *(const int4 *)(corr + g) : make_int4(0, 0, 0, 0);
        }
        if (t < PICK_CH / 64) {
            const uint64_t bg = p0 / 64 + t;
            pre_c = bg < nblk ? bmax[bg] : CMIN;
            pre_o = bg < nblk ? boff[bg] : 0;
        }
    };
    // MODE 1: exactly one window, starting at this workgroup's segment
    const uint64_t p_begin = MODE == 1 ? (uint64_t)blockIdx.x * PICK_SEG : 0;
    const uint64_t p_stop = joined ? 0 : (MODE == 1 ? min(ncorr, p_begin + 1) : ncorr);
    int cnt_last = 0;
    if (p_begin < p_stop) prefetch(p_begin);
    for (uint64_t p0 = p_begin; p0 < p_stop; p0 += PICK_CH) {
        const int cnt = (int)min((uint64_t)PICK_CH, ncorr - p0);
        cnt_last = cnt;
#pragma unroll
        for (int k = 0; k < PICK_PER_THREAD / 4; ++k) ((int4 *)cs)[k * PICK_THREADS + t] = pre[k];
        if (t < PICK_CH / 64) sm2[t] = make_int2(pre_c, t * 64 + pre_o);
        lds_barrier();
        if (MODE == 0 && p0 + PICK_CH < ncorr) prefetch(p0 + PICK_CH);      // in flight while this chunk is scanned
        if (t < 64) {
            PICK_STAT(const long long t0 = (long long)__builtin_amdgcn_s_memtime(); ++n_chunks;)
            // chunk-local 32-bit coordinates; the scan state is forced into scalar registers
            // (readfirstlane) so that the control flow of this single wave runs on the SALU
            const int mind32 = (int)mind;
            int rpos = __builtin_amdgcn_readfirstlane((int)(pos - (long long)p0));   // may be negative
            int i = 0;
            if (MODE == 1 && blockIdx.x > 0) {                 // made-up state: a peak was just appended at the segment's first position
                rpos = 0;
                val = __builtin_amdgcn_readfirstlane(cs[0]);
                i = 1;
                if (lane == 0) first_s[0] = (long long)p0;
            }
            // "first maximum of [li, ll]" for the whole wave: (value, index), index = 0x7fffffff where no lane holds it.
            // Every LDS read of the range is issued first (head block, the whole blocks in between through their
            // summaries, tail block), then combined in index order, branch-free.
            auto range_max = [&](int li, int ll, int &c, int &ci) {
                const int b_lo = li >> 6, b_hi = ll >> 6;
                const int jh = (b_lo << 6) + lane, jt = (b_hi << 6) + lane;
                const int b1 = b_lo + 1 + lane, b2 = b1 + 64, b3 = b2 + 64;   // up to 192 whole blocks: mind <= 12000
                const int ch = cs[jh], ct = cs[jt];
                const int2 s1 = sm2[b1], s2 = sm2[b2], s3 = sm2[b3];
                const bool vh = (jh >= li) & (jh <= ll);
                c = vh ? ch : CMIN;
                ci = vh ? jh : 0x7fffffff;
                const bool t1 = (b1 < b_hi) & (s1.x > c);
                c = t1 ? s1.x : c;
                ci = t1 ? s1.y : ci;
                const bool t2 = (b2 < b_hi) & (s2.x > c);
                c = t2 ? s2.x : c;
                ci = t2 ? s2.y : ci;
                const bool t3 = (b3 < b_hi) & (s3.x > c);
                c = t3 ? s3.x : c;
                ci = t3 ? s3.y : ci;
                const bool tt = (b_hi > b_lo) & (jt <= ll) & (ct > c);
                c = tt ? ct : c;
                ci = tt ? jt : ci;
            };
            // smallest index among the lanes that hold the wave maximum bc
            auto first_of = [&](int c, int ci, int bc) {
                const unsigned long long tie = __ballot(c == bc);
                if (__popcll(tie) == 1) return __builtin_amdgcn_readlane(ci, __ffsll((long long)tie) - 1);
                return wave_min_i32(c == bc ? ci : 0x7fffffff);
            };
            while (i < cnt) {
                PICK_STAT(++n_ops;)
                // the scan state is wave-uniform: pin it to scalar registers so that the branches below are s_cbranch on SCC,
                // not exec-mask manipulation
                i = __builtin_amdgcn_readfirstlane(i);
                rpos = __builtin_amdgcn_readfirstlane(rpos);
                val = __builtin_amdgcn_readfirstlane(val);
                np = __builtin_amdgcn_readfirstlane(np);
                // Steady state of the reference's scan is three dependent steps per peak: "nothing greater in the rest of
                // this peak's window" -> "a new peak starts at rpos + mind + 1" -> "first maximum of ITS window".  The second
                // and third do not depend on the outcome of the first, only on it being negative (which it almost always
                // is), so all three are evaluated in one step and the speculation is dropped when the first finds something.
                const int sp = rpos + mind32 + 1;
                if (i <= rpos + mind32 && sp + mind32 <= cnt - 1 && np + 1 < WFX_MAX_PEAKS && mind32 <= 6000) {
                    // lanes 0..31 search the rest of the current window, lanes 32..63 the window of the speculated peak:
                    // 32 lanes x 3 summaries cover 96 whole blocks (mind <= 6000), head and tail blocks take two reads each
                    const bool hiq = lane >= 32;
                    const int hl = lane & 31;
                    const int li = hiq ? sp + 1 : i, ll = hiq ? sp + mind32 : rpos + mind32;
                    const int b_lo = li >> 6, b_hi = ll >> 6;
                    const int jh0 = (b_lo << 6) + hl, jh1 = jh0 + 32, jt0 = (b_hi << 6) + hl, jt1 = jt0 + 32;
                    const int b1 = b_lo + 1 + hl, b2 = b1 + 32, b3 = b2 + 32;
                    PICK_STAT(const long long ts0 = (long long)__builtin_amdgcn_s_memtime(); ++n_spec;)
                    const int vb_l = cs[sp];                  // (issued with the other reads, used after the reductions)
                    const int ch0 = cs[jh0], ch1 = cs[jh1], ct0 = cs[jt0], ct1 = cs[jt1];
                    const int2 s1 = sm2[b1], s2 = sm2[b2], s3 = sm2[b3];
                    // candidates stay apart by source (head block halves, three summary groups, tail block halves): the
                    // sources follow each other in index order and inside a source lane order is index order, so the
                    // first maximum is "first source that holds the maximum, lowest lane" -- one DPP max chain, then
                    // ballots and scalar code instead of a second (min-index) DPP chain
                    const int v_h0 = ((jh0 >= li) & (jh0 <= ll)) ? ch0 : CMIN;
                    const int v_h1 = ((jh1 >= li) & (jh1 <= ll)) ? ch1 : CMIN;
                    const int v_s1 = b1 < b_hi ? s1.x : CMIN, v_s2 = b2 < b_hi ? s2.x : CMIN, v_s3 = b3 < b_hi ? s3.x : CMIN;
                    const int v_t0 = ((b_hi > b_lo) & (jt0 <= ll)) ? ct0 : CMIN;
                    const int v_t1 = ((b_hi > b_lo) & (jt1 <= ll)) ? ct1 : CMIN;
                    const int c = max(max(max(v_h0, v_h1), max(v_s1, v_s2)), max(max(v_s3, v_t0), v_t1));
                    const int vb = __builtin_amdgcn_readfirstlane(vb_l);
                    PICK_STAT(const long long ts1 = (long long)__builtin_amdgcn_s_memtime(); t_rd += ts1 - ts0;)
                    int bca, bcb;
                    half_max_i32(c, bca, bcb);
                    const int M = hiq ? bcb : bca;
                    const unsigned long long e_h0 = __ballot(v_h0 == M), e_h1 = __ballot(v_h1 == M), e_s1 = __ballot(v_s1 == M),
                                             e_s2 = __ballot(v_s2 == M), e_s3 = __ballot(v_s3 == M), e_t0 = __ballot(v_t0 == M),
                                             e_t1 = __ballot(v_t1 == M);
                    // index of the first maximum of half `sh` (0: lanes 0..31, 32: lanes 32..63); blo / bhi = its range's end blocks
                    auto first_in = [&](int sh, int blo, int bhi) {
                        const unsigned h0 = (unsigned)(e_h0 >> sh), h1 = (unsigned)(e_h1 >> sh), q1 = (unsigned)(e_s1 >> sh), q2 = (unsigned)(e_s2 >> sh),
                                       q3 = (unsigned)(e_s3 >> sh), t0 = (unsigned)(e_t0 >> sh), t1 = (unsigned)(e_t1 >> sh);
                        if (h0) return (blo << 6) + __builtin_ctz(h0);
                        if (h1) return (blo << 6) + 32 + __builtin_ctz(h1);
                        if (q1) return __builtin_amdgcn_readlane(s1.y, sh + __builtin_ctz(q1));
                        if (q2) return __builtin_amdgcn_readlane(s2.y, sh + __builtin_ctz(q2));
                        if (q3) return __builtin_amdgcn_readlane(s3.y, sh + __builtin_ctz(q3));
                        if (t0) return (bhi << 6) + __builtin_ctz(t0);
                        return (bhi << 6) + 32 + __builtin_ctz(t1);
                    };
                    if (bca > val) {                      // the window still held something greater: plain step, speculation unused
                        rpos = first_in(0, i >> 6, (rpos + mind32) >> 6);
                        val = bca;
                        i = sp;                           // == old rpos + mind + 1, the end of the range just searched + 1
                        continue;
                    }
                    if (lane == 0) {
                        pk_s[np - 1] = (long long)p0 + rpos;
                        first_s[np] = (long long)p0 + sp;
                    }
                    ++np;
                    rpos = sp;
                    val = vb;
                    if (bcb > val) {
                        rpos = first_in(32, (sp + 1) >> 6, (sp + mind32) >> 6);
                        val = bcb;
                    }
                    i = sp + mind32 + 1;
                    continue;
                }
                PICK_STAT(const long long tq0 = (long long)__builtin_amdgcn_s_memtime();)
                if (i - rpos > mind32) {
                    if (lane == 0) {
                        pk_s[np - 1] = (long long)p0 + rpos;
                        first_s[np] = (long long)p0 + i;
                    }
                    rpos = i;
                    val = __builtin_amdgcn_readfirstlane(cs[i]);
                    ++np;
                    ++i;
                    if (np == WFX_MAX_PEAKS) {
                        hit = 1;
                        break;
                    }
                } else {
                    const int ll = min(rpos + mind32, cnt - 1);
                    int c, ci;
                    range_max(i, ll, c, ci);
                    const int bc = wave_max_i32(c);
                    if (bc > val) {
                        rpos = first_of(c, ci, bc);
                        val = bc;
                    }
                    i = ll + 1;
                }
                PICK_STAT(t_red += (long long)__builtin_amdgcn_s_memtime() - tq0;)
            }
            pos = (long long)p0 + rpos;
            if (hit && lane == 0) done_flag = 1;
            PICK_STAT(t_pick += (long long)__builtin_amdgcn_s_memtime() - t0;)
        }
        lds_barrier();
        if (done_flag) break;
    }
    PICK_STAT(t_all = (long long)__builtin_amdgcn_s_memtime() - t_begin; if (t == 0) {
        s->dbg[0] = n_ops;
        s->dbg[1] = n_chunks;
        s->dbg[2] = t_pick;
        s->dbg[3] = t_all;
        s->dbg[5] = t_rd;
        s->dbg[6] = n_spec;
    })
    if (MODE == 1) {
        // this segment's list: every peak but the last is complete (the next one has been appended); the last one
        // is complete only if the window ended with the data
        __syncthreads();
        if (t < 64) {
            pick_seg *sg = seg + blockIdx.x;
            const bool at_end = p_begin + (uint64_t)cnt_last == ncorr;
            const int n_out = at_end ? np : np - 1, tail = at_end ? 0 : 1;
            const bool over = n_out + tail > PICK_SEG_ENT || hit;
            for (int j = lane; j < n_out + tail && j < PICK_SEG_ENT; j += 64) {
                sg->first[j] = first_s[j];
                sg->fin[j] = j < np - 1 ? pk_s[j] : pos;
            }
            if (lane == 0) {
                sg->n = over ? 0 : n_out;
                sg->tail = over ? 0 : tail;
                sg->end = at_end ? 1 : 0;
                sg->overflow = over ? 1 : 0;
            }
        }
        return;
    }
    if (t == 0) {
        if (!joined) pk_s[np - 1] = pos;
        np_s = np;
        hit_s = hit;
        s->dbg[7] = joined ? 1 : (seg != nullptr ? -1 : 0);     // which form produced the peaks: joined segments / sequential after a failed join / sequential
    }
    __syncthreads();
    np = np_s;
    for (int i = t; i < np; i += PICK_THREADS) {
        s->peak_pos[i] = pk_s[i];
        s->first_pos[i] = first_s[i];
        ok_s[i] = (i >= 1 && dev_ok(frame_samples, pk_s[i] - pk_s[i - 1])) ? 1 : 0;    // wefax.py:263-267, all gaps at once
    }
    __syncthreads();
    if (t >= 64) return;
    // ---- grouping (wefax.py:269-294): the regularity flags become a bit mask in scalar registers (two ballots),
    // so the sequential scan below runs without a single LDS round trip; the results are stored by the whole wave
    static_assert(WFX_MAX_PEAKS <= 127, "the flag mask holds 128 peaks");
    const unsigned long long okm0 = __ballot(t < np && ok_s[t] != 0);
    const unsigned long long okm1 = __ballot(64 + t < np && ok_s[64 + t] != 0);
    auto okbit = [&](int i) { return (int)(((i < 64 ? okm0 >> i : okm1 >> (i - 64))) & 1ull); };
    // nclear = number of set flags among peaks 1 .. np - 2
    auto below = [](unsigned long long m, int k) { return k <= 0 ? 0ull : (k >= 64 ? m : (m & ((1ull << k) - 1))); };
    const int nclear = __popcll(below(okm0, np - 1) & ~1ull) + __popcll(below(okm1, np - 1 - 64));
    // The reference walks i = 1 .. nclear - 2: a set flag extends the current group, a clear one closes it (empty groups
    // included) and the longest closed group wins, the first one on ties.  Per clear flag z that is "the run of set flags
    // right below z"; every lane takes two positions z and measures its run with one 128-bit shift + count-leading-ones.
    const int r_end = nclear - 1;                                      // positions 1 .. r_end - 1
    const unsigned __int128 M = ((unsigned __int128)okm1 << 64) | (unsigned __int128)okm0;      // flag 0 is never set
    int my_len = -1, my_z = 0x7fffffff;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int z = t + 64 * h;
        if (z >= 1 && z < r_end && !okbit(z)) {
            const unsigned __int128 inv = ~(M << (128 - z));               // bit 127 = flag z - 1, ...; zeros shifted in below
            const unsigned long long hi = (unsigned long long)(inv >> 64), lo = (unsigned long long)inv;
            const int len = hi ? __clzll((long long)hi) : 64 + (lo ? __clzll((long long)lo) : 64);
            if (len > my_len) {                                            // (z grows with h: the first one wins ties)
                my_len = len;
                my_z = z;
            }
        }
    }
    const int best_len = wave_max_i32(my_len);
    const int best_z = wave_min_i32(my_len == best_len ? my_z : 0x7fffffff);
    const int best_start = best_len > 0 ? best_z - best_len : 0;
    const int in_range = r_end > 1 ? r_end - 1 : 0;
    const int ones_in_range = __popcll(below(okm0, r_end) & ~1ull) + __popcll(below(okm1, r_end - 64));
    const int nclosed = in_range - ones_in_range;
    long long start = 0;
    if (nclosed != 0) {
        for (int k = t; k < best_len; k += 64) s->phasing[k] = pk_s[best_start + k];
        if (best_len > 0) start = pk_s[best_start + best_len - 1];     // wefax.py:80
    }
    if (t == 0) {
        s->npeaks = np;
        s->hit_limit = hit_s;
        s->no_group = nclosed == 0 ? 1 : 0;
        s->n_phasing = nclosed == 0 ? 0 : best_len;
        s->start_frame = start;
        s->height = (nclosed == 0 || width <= 0) ? 0 : (int)(((long long)n - start) / width);
    }
    PICK_STAT(s->dbg[4] = (long long)__builtin_amdgcn_s_memtime() - t_begin;)
}

// Launches the picker: the segment scans on all CUs, then the joining / sequential kernel on one.  The
// segment form needs the window of PICK_CH positions to hold a segment plus a few peak distances of overlap.
static int launch_pick(wfx_ctx *ctx, const int *bmax, const int *boff, uint64_t n, int n1, int n0, int64_t mindistance, double frame_samples, int width,
                       wfx_dev_scalars *d_scal)
{
    const uint64_t L = (uint64_t)(2 * n1 + n0), ncorr = n > L ? n - L : 0;
    pick_seg *seg = nullptr;
    int nseg = 0;
    const char *e = getenv("WFX_PICK_SEG");
    const bool want = !(e && atoi(e) == 0);
    if (want && ncorr >= 4ull * PICK_SEG && mindistance >= 64 && mindistance <= 6000) {
        nseg = (int)std::min<uint64_t>((ncorr + PICK_SEG - 1) / PICK_SEG, PICK_SEG_MAX);
        WFX_TRY(wfx_reserve(ctx, ctx->b_seg, (size_t)PICK_SEG_MAX * sizeof(pick_seg)));
        seg = (pick_seg *)ctx->b_seg.p;
        WFX_LAUNCH(ctx, K_SYNC_PICK, sync_pick_kernel<1>, dim3(nseg), dim3(PICK_THREADS), (const int *)ctx->b_corr.p, bmax, boff, n, n1, n0,
                   (long long)mindistance, frame_samples, width, d_scal, seg, nseg);
    }
    WFX_LAUNCH(ctx, K_SYNC_PICK, sync_pick_kernel<0>, dim3(1), dim3(PICK_THREADS), (const int *)ctx->b_corr.p, bmax, boff, n, n1, n0,
               (long long)mindistance, frame_samples, width, d_scal, seg, nseg);
    return 0;
}

// the correlation and its summaries are already in b_corr / b_tmp2 (wfx_dev_quantise_corr)
int wfx_dev_sync_pick_precomputed(wfx_ctx *ctx, uint64_t n, int n1, int n0, int64_t mindistance, double frame_samples, int width,
                                  wfx_dev_scalars *d_scal)
{
    if (mindistance < 0 || mindistance > 12000) return wfx_fail(ctx, WFX_ERR_BAD_ARG, "peak distance %lld out of range [0, 12000]", (long long)mindistance);
    const size_t nblk = (size_t)n / 64 + 2;
    const int *bmax = (const int *)ctx->b_tmp2.p, *boff = bmax + nblk;
    return launch_pick(ctx, bmax, boff, n, n1, n0, mindistance, frame_samples, width, d_scal);
}

int wfx_dev_sync_pick(wfx_ctx *ctx, const uint8_t *d, uint64_t n, int n1, int n0, int64_t mindistance, double frame_samples,
                      int width, wfx_dev_scalars *d_scal)
{
    if (2 * n1 + n0 > 500 || n1 < 0 || n0 < 0) return wfx_fail(ctx, WFX_ERR_BAD_ARG, "sync pattern length out of range");
    if (mindistance < 0 || mindistance > 12000) return wfx_fail(ctx, WFX_ERR_BAD_ARG, "peak distance %lld out of range [0, 12000]", (long long)mindistance);
    const size_t nblk = (size_t)n / 64 + 2;
    WFX_TRY(wfx_reserve(ctx, ctx->b_corr, (size_t)n * 4 + 256));
    WFX_TRY(wfx_reserve(ctx, ctx->b_tmp2, nblk * 8));
    int *bmax = (int *)ctx->b_tmp2.p, *boff = bmax + nblk;
    WFX_LAUNCH(ctx, K_SYNC_CORR, sync_corr_kernel, dim3(wfx_stream_grid(n, CORR_CH)), dim3(256), d, n, n1, n0, (int *)ctx->b_corr.p, bmax, boff);
    return launch_pick(ctx, bmax, boff, n, n1, n0, mindistance, frame_samples, width, d_scal);
}

// ===========================================================================
// a10 image assembly (wefax.py:296-327): pixel(x, y) = 255 - d[start + y w + x], then
// PIL Image.resize((w, 4h)): width unchanged -> only Pillow's vertical pass
// (ImagingResampleVertical_8bpc) with BICUBIC coefficients in 22-bit fixed point.
// ===========================================================================
__host__ __device__ __forceinline__ double pil_bicubic(double x)
{
    const double a = -0.5;
    if (x < 0.0) x = -x;
    if (x < 1.0) return ((a + 2.0) * x - (a + 3.0)) * x * x + 1;
    if (x < 2.0) return (((x - 5) * x + 8) * x - 4) * a;
    return 0.0;
}

// coefficients of output row yy (precompute_coeffs + normalize_coeffs_8bpc, Resample.c)
__host__ __device__ __forceinline__ void pil_row_coeffs(int yy, int h_in, int h_out, int &ymin, int &cnt, int kk[5])
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

// (h, start) come from the device scalars of a fused decode (s != nullptr) or from the arguments
// (sharded decode: this GPU renders lines [y0, y0 + gridDim.x) of an image of h_arg lines; d points
// at global sample 0 of the stream, possibly virtually: only the lines' own bytes are touched)
// K[r][dy] = fixed-point tap of source row ylo + dy in output row 4y + r (0 outside the row's window)
struct image_taps {
    int K[4][5];
};
__host__ __device__ __forceinline__ void image_row_taps(int y, int r, int h, int K5[5])
{
    const int ylo = y - 2 > 0 ? y - 2 : 0, yhi = y + 2 < h - 1 ? y + 2 : h - 1;
    int ym, c, k5[5];
    pil_row_coeffs(4 * y + r, h, 4 * h, ym, c, k5);
    for (int dy = 0; dy < 5; ++dy) {
        const int ki = ylo + dy - ym;
        int k = 0;
        for (int i = 0; i < 5; ++i) k = (ki == i && i < c && ylo + dy <= yhi) ? k5[i] : k;
        K5[dy] = k;
    }
}

// `interior` = the taps of any source row 2 <= y <= h - 3, evaluated on the host: for the 4x enlargement the
// filter centre is y + (r + 0.5) / 4 exactly, so Pillow's float64 weights do not depend on y there; only the
// four edge rows (clamped windows) evaluate the filter on the device
__global__ void __launch_bounds__(256) image_kernel(const uint8_t *__restrict__ d, uint64_t n, int w, const wfx_dev_scalars *__restrict__ s,
                                                   uint8_t *__restrict__ img_base, int h_arg, long long start_arg, int y0, int xchunks,
                                                   image_taps interior, wfx_dev_scalars *__restrict__ mirror, long long *__restrict__ hdr, long long room)
{
    // header of an image that is written straight into a collective's send slot: {bytes, width}
    if (hdr && blockIdx.x == 0 && threadIdx.x == 0) {
        long long nb = 4ll * (long long)(s ? s->height : h_arg) * (long long)w;
        hdr[0] = nb > room ? room : nb;
        hdr[1] = w;
    }
    // the scalars of the decode (peaks, start frame, height, levels) go to the caller's pinned host copy from here
    if (mirror && blockIdx.x == 0)
        for (int i = threadIdx.x; i < (int)(sizeof(wfx_dev_scalars) / 8); i += 256)
            ((unsigned long long *)mirror)[i] = ((const unsigned long long *)s)[i];
    // blockIdx.x = source row * xchunks + chunk of 2048 columns: every thread makes one 8-column strip
    const int h = s ? s->height : h_arg;
    const int y = y0 + (int)(blockIdx.x / (unsigned)xchunks);
    const int xc = (int)(blockIdx.x % (unsigned)xchunks);
    if (y >= h) return;
    const uint64_t start = s ? (uint64_t)s->start_frame : (uint64_t)start_arg;
    uint8_t *img = img_base - (uint64_t)4 * y0 * w;          // row 4*y0 is the first row of the local buffer
    // coefficients of the four output rows: row-uniform, so four lanes evaluate Pillow's float64 filter once
    // and hand the fixed-point taps to the workgroup through LDS
    __shared__ int sh_K[4][5];
    const int ylo = max(y - 2, 0), yhi = min(y + 2, h - 1);      // source rows any of the four can touch
    int K[4][5];
    if (y >= 2 && y <= h - 3) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int dy = 0; dy < 5; ++dy) K[r][dy] = interior.K[r][dy];
    } else {
        if (threadIdx.x < 4) {
            int K5[5];
            image_row_taps(y, (int)threadIdx.x, h, K5);
#pragma unroll
            for (int dy = 0; dy < 5; ++dy) sh_K[threadIdx.x][dy] = K5[dy];
        }
        __syncthreads();
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int dy = 0; dy < 5; ++dy) K[r][dy] = __builtin_amdgcn_readfirstlane(sh_K[r][dy]);
    }
    const int x0 = xc * 2048 + (int)threadIdx.x * 8;
    if (x0 < w) {
        const int nx = min(8, w - x0);
        // all five source rows are fetched before any is used (a row beyond yhi is clamped to yhi: it is loaded
        // but matches no tap): with the loads inside the tap loop every row was its own memory round trip
        unsigned lo[5], hi[5];
#pragma unroll
        for (int dy = 0; dy < 5; ++dy) load8_any(d + start + (uint64_t)min(ylo + dy, yhi) * w + x0, lo[dy], hi[dy]);
        int acc[4][8];
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int e = 0; e < 8; ++e) acc[r][e] = 1 << 21;
#pragma unroll
        for (int dy = 0; dy < 5; ++dy) {
            int px[8];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                px[e] = 255 - (int)((lo[dy] >> (8 * e)) & 0xff);
                px[4 + e] = 255 - (int)((hi[dy] >> (8 * e)) & 0xff);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int k = K[r][dy];
                // 8-bit pixel x 22-bit fixed-point tap (|tap| < 2^23): the 24-bit multiply-add is full rate,
                // the 32-bit v_mul_lo_u32 quarter rate -- it made this kernel VALU-bound
#pragma unroll
                for (int e = 0; e < 8; ++e) acc[r][e] += __mul24(px[e], k);
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
            const unsigned p0 = o[0] | (o[1] << 8) | (o[2] << 16) | (o[3] << 24), p1 = o[4] | (o[5] << 8) | (o[6] << 16) | (o[7] << 24);
            if (nx == 8 && (((uintptr_t)dst) & 7) == 0) {
                *(uint2 *)dst = make_uint2(p0, p1);               // a wave writes 512 contiguous bytes per instruction
            } else if (nx == 8 && (((uintptr_t)dst) & 3) == 0) {
                ((unsigned *)dst)[0] = p0;
                ((unsigned *)dst)[1] = p1;
            } else {
                for (int e = 0; e < nx; ++e) dst[e] = (uint8_t)o[e];
            }
        }
    }
}

static image_taps image_interior_taps()
{
    image_taps T;
    for (int r = 0; r < 4; ++r) image_row_taps(8, r, 64, T.K[r]);
    return T;
}

int wfx_dev_image(wfx_ctx *ctx, const uint8_t *d, uint64_t n, int w, int h_max, const wfx_dev_scalars *d_scal, uint8_t *img, wfx_dev_scalars *mirror,
                  long long *hdr, long long room)
{
    if (h_max <= 0 || w <= 0) {
        if (mirror) WFX_HIP(ctx, hipMemcpyAsync(mirror, d_scal, sizeof(wfx_dev_scalars), hipMemcpyDeviceToHost, ctx->stream));
        if (hdr) WFX_TRY(wfx_dev_export_header(ctx, d_scal, 0, w, room, hdr));
        return 0;
    }
    const int xchunks = (w + 2047) / 2048;
    WFX_LAUNCH(ctx, K_IMAGE, image_kernel, dim3((unsigned)h_max * xchunks), dim3(256), d, n, w, d_scal, img, 0, 0ll, 0, xchunks, image_interior_taps(), mirror, hdr, room);
    return 0;
}

int wfx_dev_image_rows(wfx_ctx *ctx, const uint8_t *d, uint64_t g0, uint64_t start, int w, int h_total, int y0, int rows, uint8_t *img)
{
    if (rows <= 0) return 0;
    if (w <= 0 || y0 < 0 || y0 + rows > h_total) return wfx_fail(ctx, WFX_ERR_BAD_ARG, "image rows out of range");
    const int xchunks = (w + 2047) / 2048;
    WFX_LAUNCH(ctx, K_IMAGE, image_kernel, dim3((unsigned)rows * xchunks), dim3(256), d - g0, (uint64_t)0, w, (const wfx_dev_scalars *)nullptr, img,
               h_total, (long long)start, y0, xchunks, image_interior_taps(), (wfx_dev_scalars *)nullptr, (long long *)nullptr, 0ll);
    return 0;
}

// ===========================================================================
// header of an exported stage buffer: {bytes, width}, written on the device so that no
// host round trip separates a decode from the collective that ships its image
// ===========================================================================
__global__ void export_header_kernel(const wfx_dev_scalars *__restrict__ sc, long long fixed, int width, long long room, long long *__restrict__ hdr)
{
    long long nb = fixed >= 0 ? fixed : 4ll * (long long)sc->height * (long long)width;
    if (nb > room) nb = room;
    hdr[0] = nb;
    hdr[1] = width;
}

int wfx_dev_export_header(wfx_ctx *ctx, const wfx_dev_scalars *d_scal, long long fixed, int width, long long room, long long *hdr)
{
    WFX_LAUNCH(ctx, K_IMAGE, export_header_kernel, dim3(1), dim3(1), d_scal, fixed, width, room, hdr);
    return 0;
}
