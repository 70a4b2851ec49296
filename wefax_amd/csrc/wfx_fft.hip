// Exact DFT engine for the two global-FFT operators of the reference:
//   scipy.signal.hilbert  (wefax.py:174)  -> wfx_dev_hilbert_env_fft
//   scipy.signal.resample (wefax.py:384)  -> wfx_dev_resample_fft
//
// Arbitrary length N is handled with Bluestein's chirp-z identity
//   DFT_N(a)[k] = conj(w_k) * sum_n (a[n] conj(w_n)) w_{k-n},   w_n = exp(i pi n^2 / N)
// i.e. one circular convolution of power-of-two length M >= 2N-1, computed with a
// double-precision radix-2^r pass engine:
//
//   forward  = decimation in frequency, natural order in  -> digit-reversed out
//   inverse  = decimation in time,      digit-reversed in -> natural order out
//
// so the convolution needs no reordering pass.  Each pass moves the whole array
// through HBM exactly once (16 B complex loads/stores, >= 256-byte runs) and does
// a 2^r-point sub-transform per column in registers + one LDS exchange:
// a tile is 4096 complex values (64 KiB of LDS), 256 threads x 16 values.
// The passes are HBM-bound (about 50 flop per 32 bytes moved), so there is no
// MFMA formulation here -- see DESIGN.md.
#include "wfx_internal.h"

#define WFX_TILE 4096

// ---------------------------------------------------------------------------
// complex helpers (explicit FMA: the library is built with -ffp-contract=off so
// that the parity-critical scalar code rounds like NumPy; butterflies want FMA)
// ---------------------------------------------------------------------------
__device__ __forceinline__ cplx cadd(cplx a, cplx b) { return make_double2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ cplx csub(cplx a, cplx b) { return make_double2(a.x - b.x, a.y - b.y); }
__device__ __forceinline__ cplx cmul(cplx a, cplx b)
{
    return make_double2(fma(a.x, b.x, -(a.y * b.y)), fma(a.x, b.y, a.y * b.x));
}
__device__ __forceinline__ cplx cconj(cplx a) { return make_double2(a.x, -a.y); }
// tables hold forward (e^{-i..}) values; the inverse uses the conjugate
template <int DIR>
__device__ __forceinline__ cplx dirw(cplx w)
{
    return DIR > 0 ? w : cconj(w);
}
// multiply by -i (forward) or +i (inverse)
template <int DIR>
__device__ __forceinline__ cplx mul_mi(cplx a)
{
    return DIR > 0 ? make_double2(a.y, -a.x) : make_double2(-a.y, a.x);
}

template <int DIR>
__device__ __forceinline__ void dft2(cplx &a, cplx &b)
{
    cplx t = csub(a, b);
    a = cadd(a, b);
    b = t;
}

template <int DIR>
__device__ __forceinline__ void dft4(cplx &a0, cplx &a1, cplx &a2, cplx &a3)
{
    cplx t0 = cadd(a0, a2), t1 = csub(a0, a2), t2 = cadd(a1, a3), t3 = mul_mi<DIR>(csub(a1, a3));
    a0 = cadd(t0, t2);
    a2 = csub(t0, t2);
    a1 = cadd(t1, t3);
    a3 = csub(t1, t3);
}

#define WFX_C1 0.92387953251128673848   // cos(pi/8)
#define WFX_S1 0.38268343236508978178   // sin(pi/8)
#define WFX_C2 0.70710678118654752440   // cos(pi/4)

// In-register DFTs, natural order in and out.
template <int N, int DIR>
struct dft_n;

template <int DIR>
struct dft_n<1, DIR> {
    static __device__ __forceinline__ void run(cplx *) {}
};
template <int DIR>
struct dft_n<2, DIR> {
    static __device__ __forceinline__ void run(cplx *v) { dft2<DIR>(v[0], v[1]); }
};
template <int DIR>
struct dft_n<4, DIR> {
    static __device__ __forceinline__ void run(cplx *v) { dft4<DIR>(v[0], v[1], v[2], v[3]); }
};
template <int DIR>
struct dft_n<8, DIR> {
    static __device__ __forceinline__ void run(cplx *v)
    {
        // n = g + 2 j: DFT4 over j for g = 0, 1; twiddle W8^{q} on g = 1; DFT2 over g
        dft4<DIR>(v[0], v[2], v[4], v[6]);
        dft4<DIR>(v[1], v[3], v[5], v[7]);
        const cplx w1 = dirw<DIR>(make_double2(WFX_C2, -WFX_C2));
        const cplx w3 = dirw<DIR>(make_double2(-WFX_C2, -WFX_C2));
        v[3] = cmul(v[3], w1);
        v[5] = mul_mi<DIR>(v[5]);
        v[7] = cmul(v[7], w3);
        cplx o[8];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            o[q] = cadd(v[2 * q], v[2 * q + 1]);
            o[q + 4] = csub(v[2 * q], v[2 * q + 1]);
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = o[i];
    }
};
template <int DIR>
struct dft_n<16, DIR> {
    static __device__ __forceinline__ void run(cplx *v)
    {
        // n = g + 4 j, k = q + 4 p
#pragma unroll
        for (int g = 0; g < 4; ++g) dft4<DIR>(v[g], v[g + 4], v[g + 8], v[g + 12]);
        // v[g + 4 q] = Y[g][q]; multiply by W16^{g q}
        const cplx w1 = dirw<DIR>(make_double2(WFX_C1, -WFX_S1));
        const cplx w2 = dirw<DIR>(make_double2(WFX_C2, -WFX_C2));
        const cplx w3 = dirw<DIR>(make_double2(WFX_S1, -WFX_C1));
        const cplx w6 = dirw<DIR>(make_double2(-WFX_C2, -WFX_C2));
        const cplx w9 = dirw<DIR>(make_double2(-WFX_C1, WFX_S1));
        v[1 + 4] = cmul(v[1 + 4], w1);   // g=1 q=1
        v[1 + 8] = cmul(v[1 + 8], w2);   // g=1 q=2
        v[1 + 12] = cmul(v[1 + 12], w3); // g=1 q=3
        v[2 + 4] = cmul(v[2 + 4], w2);   // g=2 q=1
        v[2 + 8] = mul_mi<DIR>(v[2 + 8]);// g=2 q=2 : W16^4 = -i
        v[2 + 12] = cmul(v[2 + 12], w6); // g=2 q=3
        v[3 + 4] = cmul(v[3 + 4], w3);   // g=3 q=1
        v[3 + 8] = cmul(v[3 + 8], w6);   // g=3 q=2
        v[3 + 12] = cmul(v[3 + 12], w9); // g=3 q=3
#pragma unroll
        for (int q = 0; q < 4; ++q) dft4<DIR>(v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]);
        // v[p + 4 q] = X[q + 4 p] -> natural order
        cplx o[16];
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int p = 0; p < 4; ++p) o[q + 4 * p] = v[p + 4 * q];
#pragma unroll
        for (int i = 0; i < 16; ++i) v[i] = o[i];
    }
};

// exp(-+ 2 pi i a / 2^log2mb), a already reduced modulo 2^log2mb
template <int DIR>
__device__ __forceinline__ cplx unit_root(uint64_t a, int log2mb)
{
    double s, c;
    // 2a / Mb is exact in double (a < 2^53, Mb a power of two)
    sincospi(ldexp((double)a, 1 - log2mb), &s, &c);
    return DIR > 0 ? make_double2(c, -s) : make_double2(c, s);
}

// ---------------------------------------------------------------------------
// One pass over the whole array.  R = RA * 16 points per column, T = 4096 / R
// columns per workgroup.
//   strided pass : column = (block of R*S, offset n2 < S), element n1 at n1*S + n2,
//                  followed (forward) / preceded (inverse) by the twiddle
//                  W_{R*S}^{n2 k1}
//   CONTIG pass  : S = 1, a column is R consecutive values (always R = 256)
// Forward: n1 = g + 16 j -> DFT_RA over j, * W_R^{g q}, LDS exchange, DFT_16 over g,
// output k1 = q + RA p.  The inverse runs the same steps backwards, conjugated.
// ---------------------------------------------------------------------------
// IN_MODE (forward passes only): 0 = read the complex work array; 1 = read a REAL signal
// packed as z[q] = x[2q+1] + i x[2q] for q < in_len, zero beyond (the Hilbert packing,
// so no separate pack kernel and no loads of the zero padding); 2 = z[q] = x[q] + 0i.
// Stores at or beyond out_limit are dropped (the inverse's last pass only has to produce
// the part of the padded result that is read afterwards).
template <int RA_BITS, int DIR, bool CONTIG, bool MULB, int IN_MODE = 0>
__global__ void __launch_bounds__(256)
fft_pass(cplx *__restrict__ A, int log2S, const cplx *__restrict__ w256, const cplx *__restrict__ bhat,
         const double *__restrict__ xin, uint64_t in_len, uint64_t out_limit)
{
    constexpr int RA = 1 << RA_BITS;
    constexpr int RB = 16;
    constexpr int R = RA * RB;
    constexpr int T = WFX_TILE / R;
    constexpr int UA = 16 / RA;
    constexpr int LDS_N = CONTIG ? T * RA * (RB + 1) : WFX_TILE;
    __shared__ cplx lds[RA > 1 ? LDS_N : 1];
    __shared__ cplx wt[RA > 1 ? 256 : 1];

    const int t = threadIdx.x;
    const uint64_t wg = blockIdx.x;
    const uint64_t colg = wg * T;
    const uint64_t n2_0 = CONTIG ? 0 : (colg & ((1ull << log2S) - 1));
    const uint64_t base = CONTIG ? wg * (uint64_t)WFX_TILE
                                 : (((colg >> log2S) << (log2S + RA_BITS + 4)) + n2_0);
    auto addr = [&](int n1, int c) -> uint64_t {
        return CONTIG ? base + (uint64_t)c * R + n1 : base + ((uint64_t)n1 << log2S) + c;
    };
    auto lidx = [&](int q, int g, int c) -> int {
        return CONTIG ? c * (RA * (RB + 1)) + q * (RB + 1) + g : (q * RB + g) * T + c;
    };
    auto ld = [&](uint64_t a) -> cplx {
        if (IN_MODE == 0) return A[a];
        if (a >= in_len) return make_double2(0.0, 0.0);
        if (IN_MODE == 1) {
            const cplx v = ((const cplx *)xin)[a];
            return make_double2(v.y, v.x);
        }
        return make_double2(xin[a], 0.0);
    };
    auto st = [&](uint64_t a, cplx v) {
        if (a < out_limit) A[a] = v;
    };

    if (RA > 1) {
        wt[t] = w256[t];
        __syncthreads();
    }

    // stage-B coordinates of this thread
    const int bq = CONTIG ? (t % RA) : (t / T);
    const int bc = CONTIG ? (t / RA) : (t % T);
    cplx v[16];

    if (DIR > 0) {
        if (RA > 1) {
#pragma unroll
            for (int u = 0; u < UA; ++u) {
                const int sa = t + 256 * u;
                const int g = CONTIG ? (sa % RB) : (sa / T);
                const int c = CONTIG ? (sa / RB) : (sa % T);
#pragma unroll
                for (int j = 0; j < RA; ++j) v[j] = ld(addr(g + RB * j, c));
                dft_n<RA, DIR>::run(v);
#pragma unroll
                for (int q = 1; q < RA; ++q) v[q] = cmul(v[q], wt[((g * q) << (4 - RA_BITS)) & 255]);
#pragma unroll
                for (int q = 0; q < RA; ++q) lds[lidx(q, g, c)] = v[q];
            }
            __syncthreads();
#pragma unroll
            for (int g = 0; g < RB; ++g) v[g] = lds[lidx(bq, g, bc)];
        } else {
#pragma unroll
            for (int g = 0; g < RB; ++g) v[g] = ld(addr(g, bc));
        }
        dft_n<16, DIR>::run(v);
        if (!CONTIG) {
            const int log2mb = log2S + RA_BITS + 4;
            const uint64_t mask = (1ull << log2mb) - 1;
            const uint64_t n2 = n2_0 + bc;
            cplx cur = unit_root<DIR>((n2 * (uint64_t)bq) & mask, log2mb);
            const cplx es = unit_root<DIR>((n2 * (uint64_t)RA) & mask, log2mb);
#pragma unroll
            for (int p = 0; p < RB; ++p) {
                v[p] = cmul(v[p], cur);
                cur = cmul(cur, es);
            }
        }
#pragma unroll
        for (int p = 0; p < RB; ++p) {
            const uint64_t a = addr(bq + RA * p, bc);
            cplx val = v[p];
            if (MULB) val = cmul(val, bhat[a]);
            st(a, val);
        }
    } else {
#pragma unroll
        for (int p = 0; p < RB; ++p) v[p] = A[addr(bq + RA * p, bc)];
        if (!CONTIG) {
            const int log2mb = log2S + RA_BITS + 4;
            const uint64_t mask = (1ull << log2mb) - 1;
            const uint64_t n2 = n2_0 + bc;
            cplx cur = unit_root<DIR>((n2 * (uint64_t)bq) & mask, log2mb);
            const cplx es = unit_root<DIR>((n2 * (uint64_t)RA) & mask, log2mb);
#pragma unroll
            for (int p = 0; p < RB; ++p) {
                v[p] = cmul(v[p], cur);
                cur = cmul(cur, es);
            }
        }
        dft_n<16, DIR>::run(v);
        if (RA > 1) {
#pragma unroll
            for (int g = 0; g < RB; ++g) {
                cplx val = v[g];
                if (g > 0) val = cmul(val, cconj(wt[((g * bq) << (4 - RA_BITS)) & 255]));
                lds[lidx(bq, g, bc)] = val;
            }
            __syncthreads();
#pragma unroll
            for (int u = 0; u < UA; ++u) {
                const int sa = t + 256 * u;
                const int g = CONTIG ? (sa % RB) : (sa / T);
                const int c = CONTIG ? (sa / RB) : (sa % T);
#pragma unroll
                for (int q = 0; q < RA; ++q) v[q] = lds[lidx(q, g, c)];
                dft_n<RA, DIR>::run(v);
#pragma unroll
                for (int j = 0; j < RA; ++j) st(addr(g + RB * j, c), v[j]);
            }
        } else {
#pragma unroll
            for (int g = 0; g < RB; ++g) st(addr(g, bc), v[g]);
        }
    }
}

// ---------------------------------------------------------------------------
// plan: log2(M) = 8 (last, contiguous) + strided passes of 4..8 bits each
// ---------------------------------------------------------------------------
int wfx_dev_fft_plan_radices(int log2m, int *ra_bits, int max_passes)
{
    if (log2m < 12) return -1;
    int rem = log2m - 8;
    int k = (rem + 7) / 8;
    if (k > max_passes) return -1;
    int basebits = rem / k, extra = rem % k;
    for (int i = 0; i < k; ++i) ra_bits[i] = (basebits + (i < extra ? 1 : 0)) - 4;
    return k;
}

static int ensure_w256(wfx_ctx *ctx)
{
    if (ctx->w256_ready) return 0;
    WFX_TRY(wfx_reserve(ctx, ctx->b_w256, 256 * sizeof(cplx)));
    cplx h[256];
    for (int i = 0; i < 256; ++i) {
        // exact octant symmetry is not needed; libm cos/sin are accurate to < 1 ulp
        double ang = -2.0 * M_PI * (double)i / 256.0;
        h[i].x = cos(ang);
        h[i].y = sin(ang);
    }
    h[0] = make_double2(1.0, 0.0);
    h[64] = make_double2(0.0, -1.0);
    h[128] = make_double2(-1.0, 0.0);
    h[192] = make_double2(0.0, 1.0);
    WFX_HIP(ctx, hipMemcpyAsync(ctx->b_w256.p, h, sizeof h, hipMemcpyHostToDevice, ctx->stream));
    WFX_HIP(ctx, hipStreamSynchronize(ctx->stream));
    ctx->w256_ready = true;
    return 0;
}

struct fft_io {
    int in_mode = 0;              // first forward pass: 0 work array, 1 packed real, 2 plain real
    const double *xin = nullptr;
    uint64_t in_len = 0;
    uint64_t out_limit = ~0ull;   // last inverse pass: drop stores at or beyond this index
};

template <int DIR, int IN_MODE>
static int launch_strided(wfx_ctx *ctx, cplx *A, int log2m, int ra_bits, int log2S, const fft_io &io, uint64_t out_limit)
{
    const unsigned grid = 1u << (log2m - 12);
    const cplx *w = (const cplx *)ctx->b_w256.p;
    const int kid = DIR > 0 ? K_FFT_FWD : K_FFT_INV;
    const cplx *nb = nullptr;
#define WFX_FFT_CASE(RAB)                                                                                                  \
    case RAB:                                                                                                              \
        WFX_LAUNCH(ctx, kid, (fft_pass<RAB, DIR, false, false, IN_MODE>), dim3(grid), dim3(256), A, log2S, w, nb, io.xin, \
                   io.in_len, out_limit);                                                                                  \
        break;
    switch (ra_bits) {
        WFX_FFT_CASE(0)
        WFX_FFT_CASE(1)
        WFX_FFT_CASE(2)
        WFX_FFT_CASE(3)
        WFX_FFT_CASE(4)
    default: return wfx_fail(ctx, WFX_ERR_STATE, "bad FFT radix %d", ra_bits);
    }
#undef WFX_FFT_CASE
    return 0;
}

// forward: natural -> digit-reversed; optionally multiply by bhat in the last pass
static int fft_forward(wfx_ctx *ctx, cplx *A, int log2m, const cplx *bhat, const fft_io &io = fft_io())
{
    WFX_TRY(ensure_w256(ctx));
    int ra[8];
    int k = wfx_dev_fft_plan_radices(log2m, ra, 8);
    if (k < 0) return wfx_fail(ctx, WFX_ERR_BAD_ARG, "unsupported FFT size 2^%d", log2m);
    int log2S = log2m;
    for (int i = 0; i < k; ++i) {
        log2S -= ra[i] + 4;
        if (i == 0 && io.in_mode == 1)
            WFX_TRY((launch_strided<1, 1>(ctx, A, log2m, ra[i], log2S, io, ~0ull)));
        else if (i == 0 && io.in_mode == 2)
            WFX_TRY((launch_strided<1, 2>(ctx, A, log2m, ra[i], log2S, io, ~0ull)));
        else
            WFX_TRY((launch_strided<1, 0>(ctx, A, log2m, ra[i], log2S, io, ~0ull)));
    }
    const unsigned grid = 1u << (log2m - 12);
    const cplx *w = (const cplx *)ctx->b_w256.p;
    const double *nx = nullptr;
    if (bhat)
        WFX_LAUNCH(ctx, K_FFT_FWD, (fft_pass<4, 1, true, true>), dim3(grid), dim3(256), A, 0, w, bhat, nx, (uint64_t)0, ~0ull);
    else
        WFX_LAUNCH(ctx, K_FFT_FWD, (fft_pass<4, 1, true, false>), dim3(grid), dim3(256), A, 0, w, bhat, nx, (uint64_t)0, ~0ull);
    return 0;
}

// inverse (unnormalised): digit-reversed -> natural
static int fft_inverse(wfx_ctx *ctx, cplx *A, int log2m, const fft_io &io = fft_io())
{
    WFX_TRY(ensure_w256(ctx));
    int ra[8];
    int k = wfx_dev_fft_plan_radices(log2m, ra, 8);
    if (k < 0) return wfx_fail(ctx, WFX_ERR_BAD_ARG, "unsupported FFT size 2^%d", log2m);
    const unsigned grid = 1u << (log2m - 12);
    const cplx *w = (const cplx *)ctx->b_w256.p;
    const double *nx = nullptr;
    WFX_LAUNCH(ctx, K_FFT_INV, (fft_pass<4, -1, true, false>), dim3(grid), dim3(256), A, 0, w, (const cplx *)nullptr, nx, (uint64_t)0, ~0ull);
    int log2S[8];
    int s = log2m;
    for (int i = 0; i < k; ++i) {
        s -= ra[i] + 4;
        log2S[i] = s;
    }
    for (int i = k - 1; i >= 0; --i)
        WFX_TRY((launch_strided<-1, 0>(ctx, A, log2m, ra[i], log2S[i], io, i == 0 ? io.out_limit : ~0ull)));
    return 0;
}

// ---------------------------------------------------------------------------
// Bluestein pointwise kernels
// ---------------------------------------------------------------------------
// w_n = exp(+i pi n^2 / N); n^2 is reduced modulo 2N in integers first
__device__ __forceinline__ cplx chirp(uint64_t n, uint64_t N)
{
    const uint64_t r = (n * n) % (2 * N);
    double s, c;
    sincospi((double)r / (double)N, &s, &c);
    return make_double2(c, s);
}

// chirp filter b[i] = w_i (i < N), w_{M-i} (i > M-N), 0 otherwise; pre-scaled by 1/M
__global__ void __launch_bounds__(256) bs_fill_b(cplx *__restrict__ B, uint64_t N, uint64_t M, double inv_m)
{
    for (uint64_t i = blockIdx.x * 256ull + threadIdx.x; i < M; i += (uint64_t)gridDim.x * 256ull) {
        cplx v = make_double2(0.0, 0.0);
        if (i < N)
            v = chirp(i, N);
        else if (M - i < N)
            v = chirp(M - i, N);
        v.x *= inv_m;
        v.y *= inv_m;
        B[i] = v;
    }
}

// U[i] = x[i] * conj(w_i) for i < N, 0 for N <= i < M
__global__ void __launch_bounds__(256) bs_prologue_real(const double *__restrict__ x, cplx *__restrict__ U, uint64_t N, uint64_t M)
{
    for (uint64_t i = blockIdx.x * 256ull + threadIdx.x; i < M; i += (uint64_t)gridDim.x * 256ull) {
        cplx v = make_double2(0.0, 0.0);
        if (i < N) {
            const cplx w = chirp(i, N);
            const double xv = x[i];
            v = make_double2(xv * w.x, -(xv * w.y));
        }
        U[i] = v;
    }
}

// Between the two transforms of the analytic signal.  With X[k] = V[k] conj(w_k):
// spectrum S = h[k] X[k] (scipy.signal.hilbert: h = 1 at DC and Nyquist, 2 on the
// positive bins, 0 on the negative ones), and the inverse DFT of S is
// conj(DFT(conj(S))) / N whose Bluestein input is conj(S[k]) conj(w_k) = h[k] conj(V[k])
// (|w_k| = 1, the two chirps cancel).
__global__ void __launch_bounds__(256) hilbert_mid(cplx *__restrict__ V, uint64_t N, uint64_t M)
{
    const uint64_t half = N / 2;
    for (uint64_t k = blockIdx.x * 256ull + threadIdx.x; k < M; k += (uint64_t)gridDim.x * 256ull) {
        cplx v = make_double2(0.0, 0.0);
        if (k < N) {
            double h;
            if (k == 0)
                h = 1.0;
            else if ((N & 1) == 0)
                h = k < half ? 2.0 : (k == half ? 1.0 : 0.0);
            else
                h = k < (N + 1) / 2 ? 2.0 : 0.0;
            if (h != 0.0) {
                const cplx a = V[k];
                v = make_double2(h * a.x, -(h * a.y));
            }
        }
        V[k] = v;
    }
}

// analytic signal z[n] = conj(V[n]) w_n / N, so |z[n]| = |V[n] / N| (np.abs(hilbert), wefax.py:175)
__global__ void __launch_bounds__(256) hilbert_abs(const cplx *__restrict__ V, uint64_t N, double inv_n, double *__restrict__ env_raw)
{
    for (uint64_t i = blockIdx.x * 256ull + threadIdx.x; i < N; i += (uint64_t)gridDim.x * 256ull) {
        const cplx a = V[i];
        env_raw[i] = hypot(a.x * inv_n, a.y * inv_n);
    }
}

static int ceil_log2_u64(uint64_t v)
{
    int l = 0;
    while ((1ull << l) < v) ++l;
    return l;
}

static int get_plan(wfx_ctx *ctx, uint64_t n, wfx_bs_plan **out)
{
    auto it = ctx->plans.find(n);
    if (it != ctx->plans.end()) {
        *out = &it->second;
        return 0;
    }
    if (n < 1 || n > (1ull << 31)) return wfx_fail(ctx, WFX_ERR_BAD_ARG, "transform length %llu out of range", (unsigned long long)n);
    if (ctx->plans.size() >= 4) {   // keep the cache small: drop everything
        hipStreamSynchronize(ctx->stream);
        for (auto &kv : ctx->plans)
            if (kv.second.bhat.p) hipFree(kv.second.bhat.p);
        ctx->plans.clear();
    }
    wfx_bs_plan pl;
    pl.n = n;
    int l = ceil_log2_u64(2 * n - 1);
    pl.log2m = l < 12 ? 12 : l;
    const uint64_t M = 1ull << pl.log2m;
    WFX_TRY(wfx_reserve(ctx, pl.bhat, M * sizeof(cplx)));
    WFX_LAUNCH(ctx, K_BS_CHIRP, bs_fill_b, dim3(wfx_stream_grid(M, 256)), dim3(256), (cplx *)pl.bhat.p, n, M, 1.0 / (double)M);
    int rc = fft_forward(ctx, (cplx *)pl.bhat.p, pl.log2m, nullptr);
    if (rc != 0) {
        hipFree(pl.bhat.p);
        return rc;
    }
    auto ins = ctx->plans.emplace(n, pl);
    *out = &ins.first->second;
    return 0;
}

// circular convolution with the chirp filter, in place: A <- IFFT(FFT(A) .* bhat)
static int bs_convolve(wfx_ctx *ctx, cplx *A, const wfx_bs_plan *pl, const fft_io &io = fft_io())
{
    WFX_TRY(fft_forward(ctx, A, pl->log2m, (const cplx *)pl->bhat.p, io));
    WFX_TRY(fft_inverse(ctx, A, pl->log2m, io));
    return 0;
}

// ---------------------------------------------------------------------------
// Analytic signal as ONE circular convolution (default exact mode).
//
// scipy.signal.hilbert is ifft(fft(x) * h): a circular convolution of x with the
// kernel ifft(h) = delta + i*kh, whose imaginary part has a closed form,
//   N even: kh[m] = (2/N) cot(pi m / N) for odd m, 0 for even m
//   N odd : kh[m] = cot(pi m / 2N) / N for odd m, -tan(pi m / 2N) / N for even m != 0
// (m reduced to (-N/2, N/2]).  The real part of the analytic signal is x itself, so
// only H = x (*) kh is computed, by zero-padded power-of-two FFTs (no Bluestein
// chirps, two transforms instead of four).
//
// N even: kh vanishes on even lags, so even outputs depend only on odd inputs and
// vice versa.  Packing z[q] = x[2q+1] + i x[2q] (length N/2) and convolving with the
// REAL kernel g[r] = kh[2r-1] gives Re = H[2p], Im = H[2p-1]: one complex
// convolution of HALF the length (M >= N-1 instead of >= 2N-1).
// ---------------------------------------------------------------------------
__device__ __forceinline__ double hilbert_tap(long long m, long long N)
{
    long long r = m % N;                    // kernel is N-periodic
    if (r > N / 2) r -= N;
    if (r < -(N / 2)) r += N;
    double s, c;
    if ((N & 1) == 0) {
        if ((r & 1) == 0) return 0.0;
        sincospi((double)r / (double)N, &s, &c);
        return (2.0 / (double)N) * (c / s);
    }
    if (r == 0) return 0.0;
    sincospi((double)r / (2.0 * (double)N), &s, &c);
    return ((r & 1) ? (c / s) : -(s / c)) / (double)N;
}

// G[i] = g_ext[j] / M with j = i (i < L) or i - M (i > M - L); g_ext[j] = kh[2j-1] (packed) or kh[j]
__global__ void __launch_bounds__(256) hconv_fill(cplx *__restrict__ G, long long N, int packed, long long L, long long M, double inv_m)
{
    for (long long i = blockIdx.x * 256ll + threadIdx.x; i < M; i += (long long)gridDim.x * 256ll) {
        double v = 0.0;
        if (i < L || M - i < L) {
            const long long j = i < L ? i : i - M;
            v = hilbert_tap(packed ? 2 * j - 1 : j, N) * inv_m;
        }
        G[i] = make_double2(v, 0.0);
    }
}

__global__ void __launch_bounds__(256) hconv_pack(const double *__restrict__ x, cplx *__restrict__ U, long long L, long long M, int packed)
{
    for (long long i = blockIdx.x * 256ll + threadIdx.x; i < M; i += (long long)gridDim.x * 256ll) {
        cplx v = make_double2(0.0, 0.0);
        if (i < L) v = packed ? make_double2(x[2 * i + 1], x[2 * i]) : make_double2(x[i], 0.0);
        U[i] = v;
    }
}

// |x + i H| (np.abs(hilbert(x)), wefax.py:175)
__global__ void __launch_bounds__(256) hconv_env(const cplx *__restrict__ V, const double *__restrict__ x, long long N, long long L, int packed,
                                                double *__restrict__ env_raw)
{
    for (long long n = blockIdx.x * 256ll + threadIdx.x; n < N; n += (long long)gridDim.x * 256ll) {
        double H;
        if (packed == 2)
            H = ((const double *)V)[n];
        else if (!packed)
            H = V[n].x;
        else if ((n & 1) == 0)
            H = V[n >> 1].x;
        else {
            const long long p = (n + 1) >> 1;
            H = V[p == L ? 0 : p].y;
        }
        env_raw[n] = hypot(x[n], H);
    }
}

static int get_hplan(wfx_ctx *ctx, uint64_t n, wfx_bs_plan **out, int *packed_out, uint64_t *L_out)
{
    const int packed = (n & 1) == 0 ? 1 : 0;
    const uint64_t L = packed ? n / 2 : n;
    *packed_out = packed;
    *L_out = L;
    auto it = ctx->hplans.find(n);
    if (it != ctx->hplans.end()) {
        *out = &it->second;
        return 0;
    }
    if (n < 1 || n > (1ull << 31)) return wfx_fail(ctx, WFX_ERR_BAD_ARG, "transform length %llu out of range", (unsigned long long)n);
    if (ctx->hplans.size() >= 2) {
        hipStreamSynchronize(ctx->stream);
        for (auto &kv : ctx->hplans)
            if (kv.second.bhat.p) hipFree(kv.second.bhat.p);
        ctx->hplans.clear();
    }
    wfx_bs_plan pl;
    pl.n = n;
    int l = ceil_log2_u64(L > 1 ? 2 * L - 1 : 1);
    pl.log2m = l < 12 ? 12 : l;
    const uint64_t M = 1ull << pl.log2m;
    WFX_TRY(wfx_reserve(ctx, pl.bhat, M * sizeof(cplx)));
    WFX_LAUNCH(ctx, K_BS_CHIRP, hconv_fill, dim3(wfx_stream_grid(M, 256)), dim3(256), (cplx *)pl.bhat.p, (long long)n, packed,
               (long long)L, (long long)M, 1.0 / (double)M);
    int rc = fft_forward(ctx, (cplx *)pl.bhat.p, pl.log2m, nullptr);
    if (rc != 0) {
        hipFree(pl.bhat.p);
        return rc;
    }
    auto ins = ctx->hplans.emplace(n, pl);
    *out = &ins.first->second;
    return 0;
}

__device__ __forceinline__ void cswap_d(double &a, double &b)
{
    const double lo = fmin(a, b), hi = fmax(a, b);
    a = lo;
    b = hi;
}

// |x + i H| of one sample.  The device library's hypot() is an exponent-scaled sqrt(fma(x, x, H * H)); for
// operands whose squares neither overflow nor underflow the scaling is exact, so the unscaled form gives the
// same bits (checked on 16.7 M pairs: tools/micro/hypot_check.hip) at a third of the instructions.
__device__ __forceinline__ double env_abs(double x, double H)
{
    const double ax = fabs(x), ah = fabs(H);
    const double big = fmax(ax, ah), small = fmin(ax, ah);
    if (big < 1e140 && (small > 1e-140 || small == 0.0)) return sqrt(__builtin_fma(x, x, H * H));
    return hypot(x, H);
}

// |x + i H| followed by the 5-tap median of wefax.py:175 (zeros beyond both ends), fused:
// a tile of 1024 envelope values + 2 halo values per side is formed in LDS.
// Packed layout: V[m] = (H[2m], H[2m-1]), so one 16-byte load serves the pair of samples (2m-1, 2m); a tile
// needs the 515 pairs m = base/2 - 1 .. base/2 + 513.  All loads of a tile are issued unguarded (clamped
// indices) and one tile ahead: a guarded load compiles to load + s_waitcnt, one HBM round trip per iteration.
__global__ void __launch_bounds__(256) hconv_env_median(const cplx *__restrict__ V, const double *__restrict__ x, long long N, long long L, int packed,
                                                       double *__restrict__ env, unsigned *__restrict__ l0hist)
{
    __shared__ double tile[1024 + 8];
    __shared__ unsigned h0[WFX_SEL_BINS];      // level 0 of the percentile radix select (bits 63..53)
    const int t = threadIdx.x;
    // level-0 digits (sign + 10 exponent bits) take a handful of values: each thread keeps a
    // run (digit, count) in registers and touches the LDS histogram only when the digit changes
    unsigned run_digit = 0, run_count = 0;
    if (l0hist)
        for (int i = t; i < WFX_SEL_BINS; i += 256) h0[i] = 0;
    const long long step = (long long)gridDim.x * 1024;
    if (packed == 1 || packed == 2) {
        // packed == 2: H is a flat array of doubles (odd lengths on the packed real transforms): the same tile structure, the pair's
        // two values of H by two 8-byte loads instead of one 16-byte one
        const double *Hf = (const double *)V;
        cplx pv[3];
        double px0[3], px1[3];
        auto prefetch = [&](long long base) {
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const long long m = base / 2 - 1 + t + 256 * k;            // pair (2m - 1, 2m)
                const long long mc = m < 0 ? 0 : (m >= L ? 0 : m);       // H[N - 1] = V[0].y
                const long long i1 = 2 * m, i0 = 2 * m - 1;
                const long long c0 = i0 < 0 ? 0 : (i0 >= N ? N - 1 : i0), c1 = i1 < 0 ? 0 : (i1 >= N ? N - 1 : i1);
                if (packed == 2)
                    pv[k] = make_double2(Hf[c1], Hf[c0]);
                else
                    pv[k] = V[mc];
                px0[k] = x[c0];
                px1[k] = x[c1];
            }
        };
        long long base = (long long)blockIdx.x * 1024;
        if (base < N) prefetch(base);
        for (; base < N; base += step) {
            __syncthreads();
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const int mi = t + 256 * k;                                // pair index inside the tile: 0 .. 514
                const long long m = base / 2 - 1 + mi;
                const long long i1 = 2 * m, i0 = 2 * m - 1;
                // V[m].y = H[2m - 1] for m < L; at m == L the clamped load read V[0], whose .y is H[N - 1]
                const double e0 = (i0 >= 0 && i0 < N) ? env_abs(px0[k], pv[k].y) : 0.0;
                const double e1 = (i1 >= 0 && i1 < N) ? env_abs(px1[k], pv[k].x) : 0.0;
                // tile[j] holds sample base - 2 + j: the pair lands at j = 2 mi - 1, 2 mi
                if (mi < 515) {
                    if (mi > 0) tile[2 * mi - 1] = e0;
                    if (mi < 514) tile[2 * mi] = e1;
                }
            }
            __syncthreads();
            if (base + step < N) prefetch(base + step);
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int j = 2 * t + 512 * u;                             // two consecutive outputs per lane
                const double w0 = tile[j], w1 = tile[j + 1], w2 = tile[j + 2], w3 = tile[j + 3], w4 = tile[j + 4], w5 = tile[j + 5];
                double r[2];
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    double a = h ? w1 : w0, b = h ? w2 : w1, c = h ? w3 : w2, d = h ? w4 : w3, e = h ? w5 : w4;
                    cswap_d(a, b);
                    cswap_d(d, e);
                    cswap_d(a, d);
                    cswap_d(b, e);
                    cswap_d(b, c);
                    cswap_d(c, d);
                    cswap_d(b, c);
                    r[h] = c;
                }
                if (base + j + 1 < N)
                    *(double2 *)(env + base + j) = make_double2(r[0], r[1]);
                else if (base + j < N)
                    env[base + j] = r[0];
                if (l0hist) {
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        if (base + j + h >= N) continue;
                        const unsigned dg = (unsigned)(wfx_f64_key(r[h]) >> 53);
                        if (dg == run_digit)
                            ++run_count;
                        else {
                            if (run_count) atomicAdd(&h0[run_digit], run_count);
                            run_digit = dg;
                            run_count = 1;
                        }
                    }
                }
            }
        }
    } else {
        for (long long base = (long long)blockIdx.x * 1024; base < N; base += step) {
            __syncthreads();
            for (int i = t; i < 1024 + 4; i += 256) {
                const long long n = base - 2 + i;
                // (packed == 2: H as a flat array of doubles -- odd lengths on the packed real transforms; 0: one point per sample)
                tile[i] = (n >= 0 && n < N) ? env_abs(x[n], packed == 2 ? ((const double *)V)[n] : V[n].x) : 0.0;
            }
            __syncthreads();
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int j = t + 256 * u;
                const bool valid = base + j < N;
                double c = 0.0;
                if (valid) {
                    double a = tile[j], b = tile[j + 1], d = tile[j + 3], e = tile[j + 4];
                    c = tile[j + 2];
                    cswap_d(a, b);
                    cswap_d(d, e);
                    cswap_d(a, d);
                    cswap_d(b, e);
                    cswap_d(b, c);
                    cswap_d(c, d);
                    cswap_d(b, c);
                    env[base + j] = c;
                }
                if (l0hist && valid) {
                    const unsigned dg = (unsigned)(wfx_f64_key(c) >> 53);
                    if (dg == run_digit)
                        ++run_count;
                    else {
                        if (run_count) atomicAdd(&h0[run_digit], run_count);
                        run_digit = dg;
                        run_count = 1;
                    }
                }
            }
        }
    }
    if (l0hist) {
        if (run_count) atomicAdd(&h0[run_digit], run_count);
        __syncthreads();
        for (int i = t; i < WFX_SEL_BINS; i += 256)
            if (h0[i]) atomicAdd(&l0hist[i], h0[i]);
    }
}

// The same for one rank's block [s0, s1) of a sharded capture (s0, s1 even; packed layout only): V and x are indexed by
// GLOBAL pair / sample index through pre-offset pointers and are valid two samples beyond the block on either side (the
// halo the distributed inverse transform delivers); zeros beyond the capture's true ends as in the reference.
__global__ void __launch_bounds__(256) hconv_env_median_block(const cplx *__restrict__ V, const double *__restrict__ x, long long N, long long s0, long long s1,
                                                             double *__restrict__ env, unsigned *__restrict__ l0hist)
{
    __shared__ double tile[1024 + 8];
    __shared__ unsigned h0[WFX_SEL_BINS];
    const int t = threadIdx.x;
    unsigned run_digit = 0, run_count = 0;
    if (l0hist)
        for (int i = t; i < WFX_SEL_BINS; i += 256) h0[i] = 0;
    const long long step = (long long)gridDim.x * 1024;
    const long long mlo = s0 / 2 - 2, mhi = s1 / 2 + 1;                     // valid pair indices
    const long long xlo = s0 - 2 < 0 ? 0 : s0 - 2, xhi = (s1 + 2 > N ? N : s1 + 2) - 1;   // valid sample indices
    cplx pv[3];
    double px0[3], px1[3];
    auto prefetch = [&](long long base) {
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const long long m = base / 2 - 1 + t + 256 * k;                 // pair (2m - 1, 2m)
            const long long mc = m < mlo ? mlo : (m > mhi ? mhi : m);
            const long long i1 = 2 * m, i0 = 2 * m - 1;
            pv[k] = V[mc];
            px0[k] = x[i0 < xlo ? xlo : (i0 > xhi ? xhi : i0)];
            px1[k] = x[i1 < xlo ? xlo : (i1 > xhi ? xhi : i1)];
        }
    };
    long long base = s0 + (long long)blockIdx.x * 1024;
    if (base < s1) prefetch(base);
    for (; base < s1; base += step) {
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const int mi = t + 256 * k;
            const long long m = base / 2 - 1 + mi;
            const long long i1 = 2 * m, i0 = 2 * m - 1;
            const double e0 = (i0 >= 0 && i0 < N) ? env_abs(px0[k], pv[k].y) : 0.0;
            const double e1 = (i1 >= 0 && i1 < N) ? env_abs(px1[k], pv[k].x) : 0.0;
            if (mi < 515) {
                if (mi > 0) tile[2 * mi - 1] = e0;
                if (mi < 514) tile[2 * mi] = e1;
            }
        }
        __syncthreads();
        if (base + step < s1) prefetch(base + step);
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int j = 2 * t + 512 * u;
            const double w0 = tile[j], w1 = tile[j + 1], w2 = tile[j + 2], w3 = tile[j + 3], w4 = tile[j + 4], w5 = tile[j + 5];
            double r[2];
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                double a = h ? w1 : w0, b = h ? w2 : w1, c = h ? w3 : w2, d = h ? w4 : w3, e = h ? w5 : w4;
                cswap_d(a, b);
                cswap_d(d, e);
                cswap_d(a, d);
                cswap_d(b, e);
                cswap_d(b, c);
                cswap_d(c, d);
                cswap_d(b, c);
                r[h] = c;
            }
            if (base + j + 1 < s1)
                *(double2 *)(env + (base + j - s0)) = make_double2(r[0], r[1]);
            else if (base + j < s1)
                env[base + j - s0] = r[0];
            if (l0hist) {
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    if (base + j + h >= s1) continue;
                    const unsigned dg = (unsigned)(wfx_f64_key(r[h]) >> 53);
                    if (dg == run_digit)
                        ++run_count;
                    else {
                        if (run_count) atomicAdd(&h0[run_digit], run_count);
                        run_digit = dg;
                        run_count = 1;
                    }
                }
            }
        }
    }
    if (l0hist) {
        if (run_count) atomicAdd(&h0[run_digit], run_count);
        __syncthreads();
        for (int i = t; i < WFX_SEL_BINS; i += 256)
            if (h0[i]) atomicAdd(&l0hist[i], h0[i]);
    }
}

int wfx_dev_env_median_block(wfx_ctx *ctx, const cplx *V_global, const double *x_global, uint64_t n_total, uint64_t s0, uint64_t s1, double *env_block,
                             unsigned *l0hist)
{
    if ((s0 & 1) || (s1 & 1) || s1 <= s0) return wfx_fail(ctx, WFX_ERR_BAD_ARG, "envelope block [%llu, %llu) must be even-aligned", (unsigned long long)s0, (unsigned long long)s1);
    const unsigned grid = std::min(wfx_stream_grid(s1 - s0, 1024), 1024u);
    WFX_LAUNCH(ctx, K_ENV_MEDIAN, hconv_env_median_block, dim3(grid), dim3(256), V_global, x_global, (long long)n_total, (long long)s0, (long long)s1, env_block,
               l0hist);
    return 0;
}

// The same for one rank's COLUMNS of a sharded capture (round 4): nseg segments of seg_len samples (even), segment s holding the
// global samples [g0 + s g_stride, + seg_len).  V rows (pairs; row s at V + s v_rs, the segment's first pair at index 0, valid
// from -1 to seg_len / 2 + 1) and x rows (samples; row s at x + s x_rs, valid from -2 to seg_len + 1) carry the halos the
// neighbouring ranks' columns delivered; zeros beyond the capture's true ends as in the reference.  env: dense [nseg][seg_len].
// flat: the rows of V hold H itself, point q = (H[2q], H[2q + 1]) (odd lengths on packed real transforms), instead of the packed
// (H[2q], H[2q - 1]).  Slots whose global index lies beyond the capture (a padded form's segments reach past its end) get the
// median of zeros and stay out of the histogram.
__global__ void __launch_bounds__(256) hconv_env_median_segs(const cplx *__restrict__ V, long long v_rs, const double *__restrict__ x, long long x_rs, int nseg,
                                                            int seg_len, long long g0, long long g_stride, long long N, double *__restrict__ env,
                                                            unsigned *__restrict__ l0hist, int flat)
{
    __shared__ double tile[1024 + 8];
    __shared__ unsigned h0[WFX_SEL_BINS];
    const int t = threadIdx.x;
    unsigned run_digit = 0, run_count = 0;
    if (l0hist)
        for (int i = t; i < WFX_SEL_BINS; i += 256) h0[i] = 0;
    const int tps = (seg_len + 1023) >> 10;                                  // tiles per segment
    const long long ntiles = (long long)nseg * tps;
    const int mhi = seg_len / 2 + 1, xhi = seg_len + 1;                      // last valid pair / sample index of a row
    cplx pv[3];
    double px0[3], px1[3];
    auto prefetch = [&](long long tix) {
        const int s = (int)(tix / tps), base = (int)(tix - (long long)s * tps) << 10;
        const cplx *Vs = V + (long long)s * v_rs;
        const double *xs = x + (long long)s * x_rs;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const int m = base / 2 - 1 + t + 256 * k;                       // pair (2m - 1, 2m), local
            const int mc = m > mhi ? mhi : m;                               // (m >= -1 always)
            const int i1 = 2 * m, i0 = 2 * m - 1;
            const int c0 = i0 < -2 ? -2 : (i0 > xhi ? xhi : i0), c1 = i1 > xhi ? xhi : i1;
            if (flat) {
                const double *Hs = (const double *)Vs;
                pv[k] = make_double2(Hs[c1], Hs[c0]);
            } else
                pv[k] = Vs[mc];
            px0[k] = xs[c0];
            px1[k] = xs[c1];
        }
    };
    long long tix = blockIdx.x;
    if (tix < ntiles) prefetch(tix);
    for (; tix < ntiles; tix += gridDim.x) {
        const int s = (int)(tix / tps), base = (int)(tix - (long long)s * tps) << 10;
        const long long gseg = g0 + (long long)s * g_stride;                // global index of the segment's sample 0
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            const int mi = t + 256 * k;
            const int m = base / 2 - 1 + mi;
            const int i1 = 2 * m, i0 = 2 * m - 1;
            const long long q0 = gseg + i0, q1 = gseg + i1;
            const double e0 = (q0 >= 0 && q0 < N && i0 <= xhi) ? env_abs(px0[k], pv[k].y) : 0.0;
            const double e1 = (q1 >= 0 && q1 < N && i1 <= xhi) ? env_abs(px1[k], pv[k].x) : 0.0;
            if (mi < 515) {
                if (mi > 0) tile[2 * mi - 1] = e0;
                if (mi < 514) tile[2 * mi] = e1;
            }
        }
        __syncthreads();
        if (tix + gridDim.x < ntiles) prefetch(tix + gridDim.x);
        double *eo = env + (long long)s * seg_len;
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int j = 2 * t + 512 * u;
            const double w0 = tile[j], w1 = tile[j + 1], w2 = tile[j + 2], w3 = tile[j + 3], w4 = tile[j + 4], w5 = tile[j + 5];
            double r[2];
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                double a = h ? w1 : w0, b = h ? w2 : w1, c = h ? w3 : w2, d = h ? w4 : w3, e = h ? w5 : w4;
                cswap_d(a, b);
                cswap_d(d, e);
                cswap_d(a, d);
                cswap_d(b, e);
                cswap_d(b, c);
                cswap_d(c, d);
                cswap_d(b, c);
                r[h] = c;
            }
            if (base + j + 1 < seg_len)
                *(double2 *)(eo + base + j) = make_double2(r[0], r[1]);      // (seg_len and base + j are even: 16-byte aligned)
            else if (base + j < seg_len)
                eo[base + j] = r[0];
            if (l0hist) {
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    if (base + j + h >= seg_len || gseg + base + j + h >= N) continue;
                    const unsigned dg = (unsigned)(wfx_f64_key(r[h]) >> 53);
                    if (dg == run_digit)
                        ++run_count;
                    else {
                        if (run_count) atomicAdd(&h0[run_digit], run_count);
                        run_digit = dg;
                        run_count = 1;
                    }
                }
            }
        }
    }
    if (l0hist) {
        if (run_count) atomicAdd(&h0[run_digit], run_count);
        __syncthreads();
        for (int i = t; i < WFX_SEL_BINS; i += 256)
            if (h0[i]) atomicAdd(&l0hist[i], h0[i]);
    }
}

int wfx_dev_env_median_segs(wfx_ctx *ctx, const cplx *V_rows, long long v_rs, const double *x_rows, long long x_rs, int nseg, int seg_len, long long g0,
                            long long g_stride, uint64_t n_total, double *env, unsigned *l0hist, int flat)
{
    if (nseg < 1 || seg_len < 2 || (seg_len & 1)) return wfx_fail(ctx, WFX_ERR_BAD_ARG, "envelope segments of %d samples", seg_len);
    const long long ntiles = (long long)nseg * ((seg_len + 1023) >> 10);
    const unsigned grid = (unsigned)std::min<long long>(ntiles, 1024);
    WFX_LAUNCH(ctx, K_ENV_MEDIAN, hconv_env_median_segs, dim3(grid), dim3(256), V_rows, v_rs, x_rows, x_rs, nseg, seg_len, g0, g_stride, (long long)n_total, env,
               l0hist, flat);
    return 0;
}

// One rank's block [s0, s1) of a sharded capture of ODD length: the packed inverse delivers H itself, point q = (H[2q], H[2q + 1]) --
// a flat array of doubles.  H (through V) and x are indexed by global sample index and valid two samples beyond the block on
// either side; zeros beyond the capture's true ends.
__global__ void __launch_bounds__(256) hconv_env_median_block_plain(const cplx *__restrict__ V, const double *__restrict__ x, long long N, long long s0,
                                                                   long long s1, double *__restrict__ env, unsigned *__restrict__ l0hist)
{
    __shared__ double tile[1024 + 8];
    __shared__ unsigned h0[WFX_SEL_BINS];
    const int t = threadIdx.x;
    unsigned run_digit = 0, run_count = 0;
    if (l0hist)
        for (int i = t; i < WFX_SEL_BINS; i += 256) h0[i] = 0;
    const long long step = (long long)gridDim.x * 1024;
    for (long long base = s0 + (long long)blockIdx.x * 1024; base < s1; base += step) {
        __syncthreads();
        for (int i = t; i < 1024 + 4; i += 256) {
            const long long n = base - 2 + i;
            tile[i] = (n >= 0 && n < N && n >= s0 - 2 && n < s1 + 2) ? env_abs(x[n], ((const double *)V)[n]) : 0.0;
        }
        __syncthreads();
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int j = t + 256 * u;
            if (base + j >= s1) continue;
            double a = tile[j], b = tile[j + 1], c = tile[j + 2], d = tile[j + 3], e = tile[j + 4];
            cswap_d(a, b);
            cswap_d(d, e);
            cswap_d(a, d);
            cswap_d(b, e);
            cswap_d(b, c);
            cswap_d(c, d);
            cswap_d(b, c);
            env[base + j - s0] = c;
            if (l0hist) {
                const unsigned dg = (unsigned)(wfx_f64_key(c) >> 53);
                if (dg == run_digit)
                    ++run_count;
                else {
                    if (run_count) atomicAdd(&h0[run_digit], run_count);
                    run_digit = dg;
                    run_count = 1;
                }
            }
        }
    }
    if (l0hist) {
        if (run_count) atomicAdd(&h0[run_digit], run_count);
        __syncthreads();
        for (int i = t; i < WFX_SEL_BINS; i += 256)
            if (h0[i]) atomicAdd(&l0hist[i], h0[i]);
    }
}

int wfx_dev_env_median_block_plain(wfx_ctx *ctx, const cplx *V_global, const double *x_global, uint64_t n_total, uint64_t s0, uint64_t s1, double *env_block,
                                   unsigned *l0hist)
{
    if (s1 <= s0) return wfx_fail(ctx, WFX_ERR_BAD_ARG, "empty envelope block");
    const unsigned grid = std::min(wfx_stream_grid(s1 - s0, 1024), 1024u);
    WFX_LAUNCH(ctx, K_ENV_MEDIAN, hconv_env_median_block_plain, dim3(grid), dim3(256), V_global, x_global, (long long)n_total, (long long)s0, (long long)s1,
               env_block, l0hist);
    return 0;
}

static int hilbert_conv(wfx_ctx *ctx, const double *x, uint64_t n, cplx **W_out, int *packed_out, uint64_t *L_out)
{
    // even N with a 13-smooth N/2: unpadded mixed-radix transforms, closed-form kernel spectrum
    if ((n & 1) == 0 && n >= 4 && !ctx->force_pow2 && wfx_mr_supported(n / 2)) {
        *packed_out = 1;
        *L_out = n / 2;
        return wfx_dev_hilbert_conv_mr(ctx, x, n, W_out);
    }
    // any other even N: the same packed convolution zero-padded to a 13-smooth M >= N - 1 on the mixed-radix passes
    if ((n & 1) == 0 && n >= 8192 && !ctx->force_pow2 && !WFX_LAB_ENV("WFX_NO_SMOOTH_PAD")) {
        int handled = 0;
        WFX_TRY(wfx_dev_hilbert_conv_mr_padded(ctx, x, n, W_out, &handled));
        if (handled) {
            *packed_out = 1;
            *L_out = n / 2;
            return 0;
        }
    }
    // odd N: samples and kernel are real -- two packed transforms of M/2 >= N points and a glue pass (wfx_mrfft.hip, round 4);
    // the result is H as a flat array of doubles (*packed_out = 2).  x[n] (the odd sample's partner in the last pair) must be zero:
    // every caller hands over a context buffer with slack behind the n samples
    if ((n & 1) == 1 && n >= 8192 && !ctx->force_pow2 && !WFX_LAB_ENV("WFX_NO_REAL_ODD")) {
        WFX_HIP(ctx, hipMemsetAsync((void *)(x + n), 0, sizeof(double), ctx->stream));
        int handled = 0;
        WFX_TRY(wfx_dev_hilbert_conv_mr_real(ctx, x, n, W_out, &handled));
        if (handled) {
            *packed_out = 2;
            *L_out = n;
            return 0;
        }
    }
    wfx_bs_plan *pl = nullptr;
    WFX_TRY(get_hplan(ctx, n, &pl, packed_out, L_out));
    const uint64_t M = 1ull << pl->log2m;
    WFX_TRY(wfx_reserve(ctx, ctx->b_work, M * sizeof(cplx)));
    cplx *W = (cplx *)ctx->b_work.p;
    // the first forward pass reads x directly (packed or plain) and the last inverse pass
    // only stores the L values that are read afterwards
    fft_io io;
    io.in_mode = *packed_out ? 1 : 2;
    io.xin = x;
    io.in_len = *L_out;
    io.out_limit = *L_out;
    WFX_TRY(bs_convolve(ctx, W, pl, io));
    *W_out = W;
    return 0;
}

int wfx_dev_hilbert_env_fft(wfx_ctx *ctx, const double *x, uint64_t n, double *env_raw)
{
    cplx *W = nullptr;
    int packed = 0;
    uint64_t L = 0;
    WFX_TRY(hilbert_conv(ctx, x, n, &W, &packed, &L));
    WFX_LAUNCH(ctx, K_ENV_MEDIAN, hconv_env, dim3(wfx_stream_grid(n, 256)), dim3(256), (const cplx *)W, x, (long long)n, (long long)L, packed, env_raw);
    return 0;
}

// envelope and median in one kernel (the decode path)
int wfx_dev_hilbert_envmed_fft(wfx_ctx *ctx, const double *x, uint64_t n, double *env, unsigned *l0hist)
{
    cplx *W = nullptr;
    int packed = 0;
    uint64_t L = 0;
    WFX_TRY(hilbert_conv(ctx, x, n, &W, &packed, &L));
    // 4 workgroups per CU: every workgroup ends with one global atomic per non-empty level-0 bin, all on the same
    // handful of addresses, and those serialise at the memory side (2048 workgroups: 47 us, 1024: 39 us)
    const unsigned grid = std::min(wfx_stream_grid(n, 1024), 1024u);
    WFX_LAUNCH(ctx, K_ENV_MEDIAN, hconv_env_median, dim3(grid), dim3(256), (const cplx *)W, x, (long long)n, (long long)L, packed, env, l0hist);
    return 0;
}

// The same operator through two Bluestein DFTs (fft -> h -> ifft literally); kept as
// an independent cross-check of the convolution form above (WFX_HILBERT_BLUESTEIN).
int wfx_dev_hilbert_env_bluestein(wfx_ctx *ctx, const double *x, uint64_t n, double *env_raw)
{
    wfx_bs_plan *pl = nullptr;
    WFX_TRY(get_plan(ctx, n, &pl));
    const uint64_t M = 1ull << pl->log2m;
    WFX_TRY(wfx_reserve(ctx, ctx->b_work, M * sizeof(cplx)));
    cplx *W = (cplx *)ctx->b_work.p;
    const dim3 gm(wfx_stream_grid(M, 256)), blk(256);
    WFX_LAUNCH(ctx, K_BS_CHIRP, bs_prologue_real, gm, blk, x, W, n, M);
    WFX_TRY(bs_convolve(ctx, W, pl));
    WFX_LAUNCH(ctx, K_BS_CHIRP, hilbert_mid, gm, blk, W, n, M);
    WFX_TRY(bs_convolve(ctx, W, pl));
    WFX_LAUNCH(ctx, K_ENV_MEDIAN, hilbert_abs, dim3(wfx_stream_grid(n, 256)), blk, (const cplx *)W, n, 1.0 / (double)n, env_raw);
    return 0;
}

// ---------------------------------------------------------------------------
// scipy.signal.resample(x, num) for real x (wefax.py:384):
//   X = rfft(x); Y[:nyq] = X[:nyq] (nyq = min(num, n0)//2 + 1), Nyquist bin doubled
//   (down) or halved (up) when min(num, n0) is even; y = irfft(Y, num) * (num / n0)
// ---------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
resample_gather(const cplx *__restrict__ V1, uint64_t n0, uint64_t num, cplx *__restrict__ U2, uint64_t M2)
{
    const uint64_t nmin = num < n0 ? num : n0;
    const uint64_t nyq = nmin / 2 + 1;
    for (uint64_t k = blockIdx.x * 256ull + threadIdx.x; k < M2; k += (uint64_t)gridDim.x * 256ull) {
        cplx u = make_double2(0.0, 0.0);
        if (k < num) {
            const bool mirror = k > num / 2;
            const uint64_t kk = mirror ? num - k : k;
            if (kk < nyq) {
                cplx X = cmul(V1[kk], cconj(chirp(kk, n0)));     // DFT_n0(x)[kk]
                if ((nmin & 1) == 0 && kk == nmin / 2) {
                    const double f = num < n0 ? 2.0 : (n0 < num ? 0.5 : 1.0);
                    X.x *= f;
                    X.y *= f;
                }
                // irfft ignores the imaginary part of the DC bin and of the Nyquist bin (num even)
                if (kk == 0 || ((num & 1) == 0 && kk == num / 2)) X.y = 0.0;
                const cplx Z = mirror ? cconj(X) : X;
                u = cmul(cconj(Z), cconj(chirp(k, num)));       // Bluestein input of DFT(conj(Z))
            }
        }
        U2[k] = u;
    }
}

__global__ void __launch_bounds__(256)
resample_final(const cplx *__restrict__ V2, uint64_t num, double inv_num, double ratio, double *__restrict__ y)
{
    for (uint64_t i = blockIdx.x * 256ull + threadIdx.x; i < num; i += (uint64_t)gridDim.x * 256ull) {
        // IDFT value = conj(V2[i] conj(w_i)) / num; its real part:
        const cplx w = chirp(i, num);
        const cplx a = V2[i];
        const double re = fma(a.x, w.x, a.y * w.y);
        y[i] = (re * inv_num) * ratio;
    }
}

// amp[k] = |X[k] / h|, k < h = N / 2 (data_packet.py:388-406: abs(fft[:h] / h)); |X[k]| = |V[k]|, the chirp has modulus 1
__global__ void __launch_bounds__(256) spectrum_abs(const cplx *__restrict__ V, uint64_t h, double *__restrict__ amp)
{
    const double hd = (double)h;
    for (uint64_t k = blockIdx.x * 256ull + threadIdx.x; k < h; k += (uint64_t)gridDim.x * 256ull) {
        const cplx a = V[k];
        amp[k] = hypot(a.x / hd, a.y / hd);
    }
}

// one-sided amplitude spectrum of a real sequence of any length (Bluestein DFT on the power-of-two passes)
int wfx_dev_spectrum_abs(wfx_ctx *ctx, const double *x, uint64_t n, double *amp)
{
    if (n < 2) return wfx_fail(ctx, WFX_ERR_BAD_ARG, "spectrum: at least two samples needed");
    wfx_bs_plan *pl = nullptr;
    WFX_TRY(get_plan(ctx, n, &pl));
    const uint64_t M = 1ull << pl->log2m;
    WFX_TRY(wfx_reserve(ctx, ctx->b_work, M * sizeof(cplx)));
    cplx *W = (cplx *)ctx->b_work.p;
    WFX_LAUNCH(ctx, K_BS_CHIRP, bs_prologue_real, dim3(wfx_stream_grid(M, 256)), dim3(256), x, W, n, M);
    WFX_TRY(bs_convolve(ctx, W, pl));
    WFX_LAUNCH(ctx, K_ENV_MEDIAN, spectrum_abs, dim3(wfx_stream_grid(n / 2, 256)), dim3(256), (const cplx *)W, n / 2, amp);
    return 0;
}

int wfx_dev_resample_fft(wfx_ctx *ctx, const double *x, uint64_t n0, uint64_t num, double *out, bool x_is_i16)
{
    if (n0 < 1 || num < 1) return wfx_fail(ctx, WFX_ERR_BAD_ARG, "resample: empty input or output");
    // even lengths with 13-smooth halves: packed real transforms on the mixed-radix passes (no padding, no chirps)
    if (!ctx->force_pow2 && wfx_mr_resample_supported(n0, num)) return wfx_dev_resample_mr(ctx, x, n0, num, out, x_is_i16);
    // any other lengths: two chirp-z transforms on the mixed-radix passes (WFX_NO_CZT=1: the power-of-two Bluestein form below)
    if (!ctx->force_pow2 && !getenv("WFX_NO_CZT")) {
        int handled = 0;
        WFX_TRY(wfx_dev_resample_czt(ctx, x, x_is_i16, n0, num, out, &handled));
        if (handled) return 0;
    }
    if (x_is_i16) {      // (the power-of-two form reads float64: only reached when the forms above were switched off)
        WFX_TRY(wfx_reserve(ctx, ctx->b_x, n0 * 8));
        WFX_TRY(wfx_dev_i16_to_f64(ctx, (const int16_t *)x, n0, (double *)ctx->b_x.p));
        x = (const double *)ctx->b_x.p;
    }
    wfx_bs_plan *p1 = nullptr;
    WFX_TRY(get_plan(ctx, n0, &p1));
    const int log2m1 = p1->log2m;
    const uint64_t M1 = 1ull << log2m1;
    WFX_TRY(wfx_reserve(ctx, ctx->b_work, M1 * sizeof(cplx)));
    cplx *W1 = (cplx *)ctx->b_work.p;
    const dim3 blk(256);
    WFX_LAUNCH(ctx, K_BS_CHIRP, bs_prologue_real, dim3(wfx_stream_grid(M1, 256)), blk, x, W1, n0, M1);
    WFX_TRY(bs_convolve(ctx, W1, p1));
    wfx_bs_plan *p2 = nullptr;
    WFX_TRY(get_plan(ctx, num, &p2));       // may evict p1's filter: p1 is not used below
    const uint64_t M2 = 1ull << p2->log2m;
    WFX_TRY(wfx_reserve(ctx, ctx->b_work2, M2 * sizeof(cplx)));
    cplx *W2 = (cplx *)ctx->b_work2.p;
    WFX_LAUNCH(ctx, K_RESAMPLE_PW, resample_gather, dim3(wfx_stream_grid(M2, 256)), blk, (const cplx *)W1, n0, num, W2, M2);
    WFX_TRY(bs_convolve(ctx, W2, p2));
    WFX_LAUNCH(ctx, K_RESAMPLE_PW, resample_final, dim3(wfx_stream_grid(num, 256)), blk, (const cplx *)W2, num,
               1.0 / (double)num, (double)num / (double)n0, out);
    return 0;
}
