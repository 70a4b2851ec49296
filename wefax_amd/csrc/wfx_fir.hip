// FIR mode of the analytic-signal stage (wefax.py:174): the sliding-window,
// LDS-tiled Hilbert transformer that BASELINE.json's north_star names.
//
//   H[n] = sum_{m odd, 0 < m <= half} h[m] * (x[n-m] - x[n+m])      (circular in n)
//   h[m] = (2/N) cot(pi m / N)   (N even)   |   (1/N) cot(pi m / (2N))   (N odd)
//
// h is the circular-convolution kernel of the FFT Hilbert transform truncated to
// `taps` = 2*half+1 taps (even lags are exactly zero for even N and O(m/N^2) for odd
// N, where they are dropped).  It is the halo-local operator used when a capture is
// sharded across GPUs; it matches the exact DFT path to <= 1 LSB on clean captures
// at 4095 taps but not on noisy ones, so the exact path
// stays the default.
//
// Mapping (wave64, fp32 VALU -- no MFMA by design): a workgroup owns 4096 outputs,
// a lane 16 consecutive ones (16 accumulators).  One x value fetched from LDS feeds
// 8 FMAs (the 8 outputs of the block whose lag to it is odd), so the loop is
// FMA-bound: 2 ds_read_b64 + 32 v_fma_f32 per pair of source samples.  A lane's
// block starts every 16 floats; LDS rows are padded by 2 floats per 16 so that the
// stride-16 accesses of a wave hit 32 distinct bank pairs.
#include "wfx_internal.h"

#define FIR_R 16
#define FIR_TILE 4096            // 4 waves x 64 lanes x 16 outputs

__device__ __forceinline__ int fir_pad(int j) { return j + 2 * (j >> 4); }

__global__ void __launch_bounds__(256)
fir_hilbert_kernel(const double *__restrict__ xf, long long n, const float *__restrict__ taps_g, int ntap_padded, int halo, int usteps,
                   double *__restrict__ env_raw)
{
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int xs_len = FIR_TILE + 2 * halo;
    float *xs = smem;                                   // padded sample window
    float *ts = smem + ((fir_pad(xs_len) + 3) & ~3);    // taps, index i+8 holds T[i] = h[2i+1]
    const int t = threadIdx.x;
    const long long base = (long long)blockIdx.x * FIR_TILE;

    for (int i = t; i < xs_len; i += 256) {
        long long g = (base - halo + i) % n;
        if (g < 0) g += n;
        xs[fir_pad(i)] = (float)xf[g];
    }
    for (int i = t; i < ntap_padded; i += 256) ts[i] = taps_g[i];
    __syncthreads();

    const int i0 = halo + t * FIR_R;     // LDS index of this lane's first output
    float acc[FIR_R];
#pragma unroll
    for (int r = 0; r < FIR_R; ++r) acc[r] = 0.f;

    // pair-step u handles source offsets t = 2u (a) and t = 2u-1 (b) on both sides;
    // u runs from -7 in groups of 8, tap window Tw[j] = T[u0 - 1 + j]
    for (int g8 = 0; g8 < usteps; ++g8) {
        const int u0 = -7 + 8 * g8;
        float tw[17];
        const float4 *tp = (const float4 *)(ts + (u0 + 7));
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float4 v = tp[q];
            tw[4 * q] = v.x;
            tw[4 * q + 1] = v.y;
            tw[4 * q + 2] = v.z;
            tw[4 * q + 3] = v.w;
        }
        tw[16] = ts[u0 + 7 + 16];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int u = u0 + j;
            // minus side: x[n - m]
            const float2 lo = *(const float2 *)(xs + fir_pad(i0 - 2 * u));        // (t=2u, t=2u-1)
            // plus side: x[n + m], mirrored block index
            const float2 hi = *(const float2 *)(xs + fir_pad(i0 + 14 + 2 * u));   // (t=2u-1, t=2u)
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                acc[2 * k + 1] = fmaf(tw[j + 1 + k], lo.x, acc[2 * k + 1]);      // T[u+k]   * a
                acc[2 * k] = fmaf(tw[j + k], lo.y, acc[2 * k]);                  // T[u-1+k] * b
                acc[14 - 2 * k] = fmaf(-tw[j + 1 + k], hi.y, acc[14 - 2 * k]);   // T[u+k]   * a'
                acc[15 - 2 * k] = fmaf(-tw[j + k], hi.x, acc[15 - 2 * k]);       // T[u-1+k] * b'
            }
        }
    }

    const long long o0 = base + (long long)t * FIR_R;
#pragma unroll
    for (int r = 0; r < FIR_R; ++r) {
        const long long o = o0 + r;
        if (o < n) env_raw[o] = hypot(xf[o], (double)acc[r]);
    }
}

int wfx_dev_hilbert_env_fir(wfx_ctx *ctx, const double *x, uint64_t n, int taps, double *env_raw, uint64_t n_global)
{
    const uint64_t nk = n_global ? n_global : n;      // the kernel is that of the WHOLE signal
    if (taps < 3 || (taps & 1) == 0) return wfx_fail(ctx, WFX_ERR_BAD_ARG, "fir_taps must be odd and >= 3");
    int half = (taps - 1) / 2;
    if ((uint64_t)half > (n - 1) / 2) half = (int)((n - 1) / 2);      // kernel cannot be longer than the signal
    if (half < 1) half = 1;
    const int ntap = (half + 1) / 2;                 // odd lags 1, 3, ..., <= half
    const int U = (half + 1) / 2;                    // last pair-step index
    const int usteps = (U + 8 + 7) / 8;              // groups of 8 pair-steps starting at u = -7
    const int ntap_padded = 8 + 8 * usteps + 24;
    const int halo = ((half + 16 + 16 * 0) + 15 + 16) & ~15;   // >= half + 16, multiple of 16
    std::vector<float> h((size_t)ntap_padded, 0.f);
    const double N = (double)nk;
    for (int i = 0; i < ntap; ++i) {
        const double m = 2.0 * i + 1.0;
        double v;
        if ((nk & 1) == 0)
            v = (2.0 / N) / tan(M_PI * m / N);
        else
            v = (1.0 / N) / tan(M_PI * m / (2.0 * N));
        h[(size_t)i + 8] = (float)v;
    }
    WFX_TRY(wfx_reserve(ctx, ctx->b_taps, h.size() * sizeof(float)));
    WFX_HIP(ctx, hipMemcpyAsync(ctx->b_taps.p, h.data(), h.size() * sizeof(float), hipMemcpyHostToDevice, ctx->stream));
    WFX_HIP(ctx, hipStreamSynchronize(ctx->stream));   // h is a local vector
    const int xs_len = FIR_TILE + 2 * halo;
    const size_t lds_floats = (size_t)(((xs_len + 2 * (xs_len >> 4)) + 3) & ~3) + 8 + (size_t)ntap_padded;
    const size_t lds_bytes = lds_floats * sizeof(float);
    if (lds_bytes > 160 * 1024) return wfx_fail(ctx, WFX_ERR_BAD_ARG, "fir_taps %d needs %zu bytes of LDS (max 160 KiB)", taps, lds_bytes);
    static bool attr_set = false;
    if (!attr_set) {
        hipFuncSetAttribute((const void *)fir_hilbert_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr_set = true;
    }
    const unsigned grid = (unsigned)((n + FIR_TILE - 1) / FIR_TILE);
    wfx_prof_begin(ctx, K_FIR_ANALYTIC);
    hipLaunchKernelGGL(fir_hilbert_kernel, dim3(grid), dim3(256), lds_bytes, ctx->stream, x, (long long)n,
                       (const float *)ctx->b_taps.p, ntap_padded, halo, usteps, env_raw);
    wfx_prof_end(ctx);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return wfx_fail_hip(ctx, e, "launch fir_hilbert_kernel");
    return 0;
}
