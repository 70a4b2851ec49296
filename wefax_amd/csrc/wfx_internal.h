// Internal declarations shared by the translation units of libwefax_hip.so.
// gfx950 (MI355X) only: wave64, 160 KiB LDS per CU, no portability shims.
#pragma once

#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include <cstdint>
#include <cstdlib>
#include <cstdio>
#include <map>
#include <string>
#include <utility>
#include <vector>

#include "wefax_hip.h"

typedef double2 cplx;   // interleaved (re, im), one 16-byte vector access
struct wfx_comm;

// ---- kernel ids for the per-kernel HIP-event profile ------------------------
enum wfx_kernel_id {
    K_MERGE = 0,
    K_NOTCH,
    K_BS_CHIRP,        // Bluestein pointwise kernels (prologue / mid / fill)
    K_FFT_FWD,         // power-of-two FFT pass, forward (DIF)
    K_FFT_INV,         // power-of-two FFT pass, inverse (DIT)
    K_ENV_MEDIAN,      // Bluestein epilogue + |z| + 5-tap median
    K_FIR_ANALYTIC,    // sliding-window FIR Hilbert + |z| (FIR mode)
    K_MEDIAN,          // stand-alone 5-tap median (FIR mode)
    K_SELECT_HIST,
    K_SELECT_SCAN,
    K_QUANTISE,
    K_SYNC_CORR,
    K_SYNC_PICK,
    K_IMAGE,
    K_RESAMPLE_PW,     // resampler pointwise kernels
    K_POLYPHASE_IN,    // time-domain front end, stage that reads the raw int16 capture (merge fused)
    K_POLYPHASE,       // time-domain front end, later stages (float64 in)
    K_DIST_COPY,       // sharded decode: pack / unpack copies around an exchange (wfx_dist.hip)
    K_FMM_UP,          // multipole Hilbert transform: notch + P2M + M2M of the leaf workgroups (wfx_fmm.hip)
    K_FMM_MID,         // ... the tiers above them and the top of the tree
    K_FMM_TREE,        // ... the leaf workgroups' six levels downwards
    K_FMM_LEAF,        // ... near field + L2P + envelope + median
    K_RS_UP,           // multipole resampler: weights + P2M + M2M
    K_RS_LEAF,         // ... near field + L2P
    K_COUNT
};

struct wfx_devbuf {
    void *p = nullptr;
    size_t cap = 0;
};

struct wfx_bs_plan {          // Bluestein plan for one transform length N
    uint64_t n = 0;
    int log2m = 0;
    wfx_devbuf bhat;          // FFT_M(chirp filter) / M, in the pass engine's output order
};

// scalars that live on the device between the kernels of one decode
struct wfx_dev_scalars {
    double low, high;
    unsigned long long nan_count;
    int npeaks, hit_limit, no_group, n_phasing;
    long long start_frame;
    int height, pad_;
    long long peak_pos[WFX_MAX_PEAKS + 1];
    long long first_pos[WFX_MAX_PEAKS + 1];
    long long phasing[WFX_MAX_PEAKS + 1];
    // radix-select state: 4 queries
    unsigned long long sel_prefix[4];    // after level 0 (11 bits)
    unsigned long long sel_rank[4];
    unsigned long long sel_prefix2[4];   // after level 1 (22 bits)
    unsigned long long sel_rank2[4];
    double sel_value[4];
    long long dbg[8];                    // diagnostic counters (tools/), not part of the ABI
};

struct wfx_prof_rec {
    hipEvent_t a, b;
    int kid;
};

struct wfx_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    std::string err;

    // named device buffers (grown on demand, reused across calls)
    wfx_devbuf b_in, b_x, b_audio, b_work, b_work2, b_envraw, b_env, b_dig, b_corr,
        b_img, b_hist, b_tmp, b_tmp2, b_w256, b_scal, b_taps, b_cand, b_pcoef, b_seg, b_png, b_synth;
    unsigned long long synth_key = 0;       // recipe whose chunk phases b_synth holds (wfx_synth_capture)
    void *h_png = nullptr;        // pinned host image of the last PNG file (wfx_decode_png)
    size_t h_png_cap = 0;
    bool w256_ready = false;
    void *ext_img = nullptr;     // wfx_decode_bind_image: {16-byte header, image} target owned by the caller (a collective's send slot)
    size_t ext_img_cap = 0;
    bool img_in_ext = false;     // the last decode wrote its image there
    bool force_pow2 = false;     // WFX_HILBERT_FFT_POW2: always use the zero-padded power-of-two convolution
    std::map<uint64_t, wfx_bs_plan> plans;    // Bluestein chirp filters (resampler, cross-check mode)
    std::map<uint64_t, wfx_bs_plan> hplans;   // Hilbert convolution kernels
    // filter tables of the time-domain front end, keyed by content: a table is uploaded (and the stream synchronised) the first
    // time a decode uses it, never again -- repeated decodes enqueue their stages without touching the host
    struct coef_entry {
        uint64_t hash;
        size_t bytes;
        void *dev;
    };
    std::vector<coef_entry> coef_cache;
    std::vector<coef_entry> coef_retired;                            // the generation before: freed at the NEXT eviction, so a table a caller just fetched outlives 64 further misses
    std::vector<std::pair<uint64_t, const double *>> fmm_tables;      // per capture length: tables of the fast-multipole Hilbert transform (wfx_fmm.hip)

    // decode state
    wfx_decode_params dp{};
    bool have_input = false;
    bool ran = false;
    const void *ext_in = nullptr;           // wfx_decode_attach: the capture lives in caller-owned device memory
    wfx_decode_info *h_info = nullptr;      // pinned
    wfx_dev_scalars *h_scal = nullptr;      // pinned mirror

    // measurement
    hipEvent_t t0 = nullptr, t1 = nullptr;
    bool prof = false;
    std::vector<wfx_prof_rec> prof_recs;
    std::vector<hipEvent_t> ev_pool;
    uint64_t prof_count[K_COUNT] = {};
    double prof_ms[K_COUNT] = {};
};

// ---- one pass of the mixed-radix transform engine (wfx_mrfft.hip) -------------------------------------------
#define MR_MAXF 8
struct mr_pass_desc {
    int ra, rb;                 // R = ra * rb has a register-resident two-level pass (0, 0: per-prime LDS stages only)
    int R, nf, f[MR_MAXF];
    long long P, ncol, L;       // layout: columns j in [0, ncol), k = j mod P, in x[j + m ncol], out y[(j - k) R + k + q P]
    int T, log2t;
    // twiddle / spectrum indices when the array is one rank's part of a distributed transform (wfx_dist.hip); dist == 0: plain
    int dist, B, kb0, kc0, kb1, kscale, kstep;
    long long Ptw, Ltw;         // the pass's P and the length L of the GLOBAL plan (Hilbert spectrum); == P, L when plain
    // first pass (P == 1) of a distributed transform: output q of column j goes to ((cplx *)qmap[q].base)[j * qmap[q].stride]
    // instead of out[j R + q] -- straight into the send buffers of the exchange that follows (no packing copy); null: off
    const struct mr_qmap *qmap;
    // outputs with skip_lo < index < skip_hi are not stored (skip_hi == 0: all are): the bins of a forward spectrum that
    // scipy.signal.resample's down-sampling never reads (wefax.py:384 keeps the lowest num/2 + 1 bins and their mirrors)
    long long skip_lo, skip_hi;
    // zero-padded convolution on these passes (any even capture length, wfx_dev_hilbert_conv_mr_padded): a first pass from packed
    // reals reads only the first in_len points, the rest of the array counts as zero (0: no limit); a last forward pass of
    // OUT_MODE 4 multiplies output o by gtab[o] (the transformed kernel)
    long long in_len;
    const double2 *gtab;
    // pair passes of a distributed transform whose neighbours live in the columns layout (wfx_dist.hip, round 4): the first pass
    // reads row m of its input at in + m * in_rs (elements) instead of m * ncol, the last inverse pass (P == ncol) writes output
    // row q at out + q * out_rs -- rows that carry a halo between them.  0: the dense strides
    long long in_rs, out_rs;
    // the pass reads its input with non-temporal loads: set by wfx_mr_launch_pair for arrays the Infinity Cache cannot hold anyway (there
    // the passes of the 60-minute captures run 10-13 % faster with it, those of the 10-minute one -- 57 MB arrays -- 3 % slower)
    int nt_in;
};
struct mr_qmap {
    unsigned long long base;
    long long stride;
};

// ---- error helpers ----------------------------------------------------------
int wfx_fail(wfx_ctx *ctx, int code, const char *fmt, ...);
int wfx_fail_hip(wfx_ctx *ctx, hipError_t e, const char *what);
void wfx_set_global_error(const char *msg);

#define WFX_HIP(ctx, call)                                         \
    do {                                                           \
        hipError_t e_ = (call);                                    \
        if (e_ != hipSuccess) return wfx_fail_hip(ctx, e_, #call); \
    } while (0)

#define WFX_TRY(expr)             \
    do {                          \
        int rc_ = (expr);         \
        if (rc_ != 0) return rc_; \
    } while (0)

const float *wfx_coef_device(wfx_ctx *ctx, const float *host, size_t count);      // nullptr + error set on failure
int wfx_reserve(wfx_ctx *ctx, wfx_devbuf &b, size_t bytes);
bool wfx_ctx_alive(const wfx_ctx *ctx);      // false once wfx_destroy has run on it

// ---- profiling-aware launch ---------------------------------------------------
void wfx_prof_begin(wfx_ctx *ctx, int kid);
void wfx_prof_events(wfx_ctx *ctx, int kid, hipEvent_t *a, hipEvent_t *b);      // a record whose events the launch itself fills
void wfx_prof_end(wfx_ctx *ctx);

// While the per-kernel profile is on, a launch carries its own start / stop events (hipExtLaunchKernelGGL: the time stamps of the
// dispatch itself, what rocprofv3's kernel trace reports); event records queued around a launch read ~2 us long on a 25 us kernel.
#define WFX_LAUNCH(ctx, kid, kern, grid, block, ...)                                                            \
    do {                                                                                                        \
        hipEvent_t ea_ = nullptr, eb_ = nullptr;                                                                \
        if ((ctx)->prof) wfx_prof_events(ctx, kid, &ea_, &eb_);                                                 \
        if (ea_)                                                                                                \
            hipExtLaunchKernelGGL(kern, grid, block, 0, (ctx)->stream, ea_, eb_, 0, __VA_ARGS__);               \
        else                                                                                                    \
            hipLaunchKernelGGL(kern, grid, block, 0, (ctx)->stream, __VA_ARGS__);                               \
        hipError_t e_ = hipGetLastError();                                                                      \
        if (e_ != hipSuccess) return wfx_fail_hip(ctx, e_, "launch " #kern);                                    \
    } while (0)

// Loads of arrays that are read ONCE and are larger than the Infinity Cache can hold anyway: non-temporal.  The transform passes gain
// 10-13 % with them on the 60-minute captures (wfx_mrfft.hip), the resampler's glue pass 18 %, the percentile select's two passes 4-7 %;
// kernels whose lanes share lines with their neighbours, or that find their input still in the caches from the kernel before, do not
// (envelope + median: 8 % slower; the notch: no change; the quantiser: 75 % slower) and keep the default loads.  `nt` is a kernel argument
// the launcher sets with wfx_nt_for(bytes of the array); WFX_MR_NT=0|1 forces it off / on (tests run small captures with 1).
#ifdef __HIPCC__
typedef double wfx_v2d __attribute__((ext_vector_type(2)));
__device__ __forceinline__ double wfx_ld(const double *p, int nt) { return nt ? __builtin_nontemporal_load(p) : *p; }
__device__ __forceinline__ short wfx_ld(const short *p, int nt) { return nt ? __builtin_nontemporal_load(p) : *p; }
__device__ __forceinline__ double2 wfx_ld(const double2 *p, int nt)
{
    if (nt) {
        const wfx_v2d q = __builtin_nontemporal_load((const wfx_v2d *)p);
        return make_double2(q.x, q.y);
    }
    return *p;
}
#endif
static inline int wfx_nt_for(double bytes)
{
    const char *e = getenv("WFX_MR_NT");
    return e ? atoi(e) : (bytes > 128.0 * 1048576.0 ? 1 : 0);
}

static inline unsigned wfx_blocks(uint64_t n, unsigned per_block)
{
    return (unsigned)((n + per_block - 1) / per_block);
}

// grid size for grid-stride streaming kernels: enough workgroups to fill 256 CUs
static inline unsigned wfx_stream_grid(uint64_t n, unsigned per_block)
{
    uint64_t b = (n + per_block - 1) / per_block;
    if (b < 1) b = 1;
    if (b > 2048) b = 2048;
    return (unsigned)b;
}

// Lab switches (A/B experiments, diagnostics that CHANGE results) exist only in variant builds: `bash tools/build_variant.sh <name> <source>
// -DWFX_LAB` -- in the shipped library WFX_LAB_ENV(x) is a constant null pointer and the branches behind it are compiled away.  What remains
// readable from the environment are settings a deployment may want (WFX_LINK_GBS, WFX_LINK_LAT_US, WFX_SHARD_CHUNKS, WFX_SHARD_ROWS,
// WFX_PNG_*, WFX_PLACE_TRIES, WFX_DEBUG) and test hooks that select another, EQUIVALENT code path (WFX_INGEST_NI, WFX_INGEST_TILE, WFX_NO_CZT,
// WFX_NO_I16_RESAMPLE, WFX_MR_NT, WFX_PICK_SEG, WFX_COMM_ASYNC): none of them can make a result wrong.
#ifdef WFX_LAB
#define WFX_LAB_ENV(name) getenv(name)
#define WFX_LAB_FLAGS(expr) (expr)
#else
#define WFX_LAB_ENV(name) ((const char *)nullptr)
#define WFX_LAB_FLAGS(expr) (0)
#endif

// ---- radix-select helpers shared by the kernels that fuse the level-0 histogram ----
#define WFX_SEL_BINS 2048
#ifdef __HIPCC__
__device__ __forceinline__ unsigned long long wfx_f64_key(double v)
{
    unsigned long long u = (unsigned long long)__double_as_longlong(v);
    return (u >> 63) ? ~u : (u | 0x8000000000000000ull);     // order-preserving map to unsigned
}
// add one digit to an LDS histogram.  Up to three distinct digits among the wave's valid
// lanes (smooth envelopes, coarse digits) are peeled off with one atomic each; whatever
// is left falls back to one atomic per lane.
__device__ __forceinline__ void wfx_sel_count(unsigned *h, unsigned digit, bool valid, int lane)
{
    unsigned long long m = __ballot(valid);
#pragma unroll
    for (int round = 0; round < 3; ++round) {
        if (m == 0) return;
        const int first = __ffsll((long long)m) - 1;
        const unsigned d0 = (unsigned)__builtin_amdgcn_readlane((int)digit, first);
        const unsigned long long same = __ballot(valid && digit == d0);
        if (lane == first) atomicAdd(&h[d0], (unsigned)__popcll(same));
        if (digit == d0) valid = false;
        m &= ~same;
    }
    if (valid) atomicAdd(&h[digit], 1u);
}
// 5-tap median as scipy.signal.medfilt(., 5) orders it (wefax.py:175); the same exchange network in every kernel that fuses it
__device__ __forceinline__ void wfx_cswap(double &a, double &b)
{
    const double lo = fmin(a, b), hi = fmax(a, b);
    a = lo;
    b = hi;
}
__device__ __forceinline__ double wfx_median5(double a, double b, double c, double d, double e)
{
    wfx_cswap(a, b);
    wfx_cswap(d, e);
    wfx_cswap(a, d);      // a is the smallest of a, b, d, e -> not the median
    wfx_cswap(b, e);      // e is the largest of a, b, d, e  -> not the median
    wfx_cswap(b, c);      // remaining: b, c, d -> median of three
    wfx_cswap(c, d);
    wfx_cswap(b, c);
    return c;
}
#endif

// ---- device-level stage functions (device pointers in, device pointers out) --
// wfx_fft.hip
int wfx_dev_fft_plan_radices(int log2m, int *ra_bits, int max_passes);   // host only
int wfx_dev_hilbert_env_fft(wfx_ctx *ctx, const double *x, uint64_t n, double *env_raw);
int wfx_dev_spectrum_abs(wfx_ctx *ctx, const double *x, uint64_t n, double *amp);
int wfx_dev_hilbert_envmed_fft(wfx_ctx *ctx, const double *x, uint64_t n, double *env, unsigned *l0hist);
int wfx_dev_hilbert_env_bluestein(wfx_ctx *ctx, const double *x, uint64_t n, double *env_raw);
// |x + i H| + median 5 + level-0 histogram of the block [s0, s1) of a sharded capture: V_global[m] = (H[2m], H[2m-1]) and x_global[i]
// are pointers pre-offset to GLOBAL indices, valid two samples beyond the block on either side; env_block[i - s0]
int wfx_dev_env_median_block_plain(wfx_ctx *ctx, const cplx *V_global, const double *x_global, uint64_t n_total, uint64_t s0, uint64_t s1, double *env_block,
                                   unsigned *l0hist);                // odd captures: H as a flat array, ((const double *)V_global)[n] = H[n]
int wfx_dev_env_median_block(wfx_ctx *ctx, const cplx *V_global, const double *x_global, uint64_t n_total, uint64_t s0, uint64_t s1, double *env_block,
                             unsigned *l0hist);
// one rank's COLUMNS: nseg segments of seg_len samples; V_rows / x_rows point at the first OWN pair / sample of row 0 (halos on both sides)
int wfx_dev_env_median_segs(wfx_ctx *ctx, const cplx *V_rows, long long v_rs, const double *x_rows, long long x_rs, int nseg, int seg_len, long long g0,
                            long long g_stride, uint64_t n_total, double *env, unsigned *l0hist, int flat = 0);
// x_is_i16: x points at int16 samples; valid only when wfx_mr_resample_supported(n0, num) (the mixed-radix form reads them in place)
int wfx_dev_resample_fft(wfx_ctx *ctx, const double *x, uint64_t n0, uint64_t num, double *out, bool x_is_i16 = false);

// wfx_stages.hip
int wfx_dev_merge(wfx_ctx *ctx, const int16_t *lr, uint64_t n, double *out);
int wfx_dev_merge_any(wfx_ctx *ctx, const void *lr, int in_kind, uint64_t n, double *out);
inline bool wfx_kind_is_stereo(int k) { return k == WFX_IN_I16_STEREO || k == WFX_IN_U8_STEREO || k == WFX_IN_I32_STEREO || k == WFX_IN_F32_STEREO; }
inline size_t wfx_kind_frame_bytes(int k)
{
    switch (k) {
    case WFX_IN_I16_MONO: case WFX_IN_U8_STEREO: return 2;
    case WFX_IN_I16_STEREO: case WFX_IN_F32_MONO: return 4;
    default: return 8;       // float64 mono, int32 / float32 stereo
    }
}
int wfx_dev_i16_to_f64(wfx_ctx *ctx, const int16_t *in, uint64_t n, double *out);
// clear (optional): device scalars the kernel zeroes on its way (saves the decode's memset launch);
// *cleared tells whether the form that ran did it
// ext18 (optional): the odd extension's 9 + 9 samples when the caller evaluated them (in the capture's own dtype)
int wfx_dev_notch(wfx_ctx *ctx, const void *in, int in_kind, uint64_t n, const double b[3],
                  const double a[3], double *out, wfx_dev_scalars *clear = nullptr, bool *cleared = nullptr, const double *ext18 = nullptr);
int wfx_dev_median5(wfx_ctx *ctx, const double *env_raw, uint64_t n, double *env, unsigned *l0hist);
int wfx_dev_select(wfx_ctx *ctx, const double *env, uint64_t n, const uint64_t ranks[4],
                   wfx_dev_scalars *d_scal);
int wfx_dev_percentile_lerp(wfx_ctx *ctx, double gamma_lo, double gamma_hi,
                            wfx_dev_scalars *d_scal);
int wfx_dev_percentiles(wfx_ctx *ctx, const double *env, uint64_t n, const uint64_t ranks[4],
                        double gamma_lo, double gamma_hi, wfx_dev_scalars *d_scal);
int wfx_dev_select_workspace(wfx_ctx *ctx, uint64_t n, unsigned **ws);
int wfx_dev_percentiles_fused(wfx_ctx *ctx, const double *env, uint64_t n, const uint64_t ranks[4],
                              double gamma_lo, double gamma_hi, wfx_dev_scalars *d_scal);
int wfx_dev_quantise(wfx_ctx *ctx, const double *env, uint64_t n, const wfx_dev_scalars *d_scal, uint8_t *out, wfx_dev_scalars *d_scal_out,
                     double eps = 0.0);
int wfx_dev_median3(wfx_ctx *ctx, const double *env_raw, uint64_t n, double *env);
int wfx_dev_sync_corr(wfx_ctx *ctx, const uint8_t *d, uint64_t n, int n1, int n0, int32_t *corr);
int wfx_dev_quantise_corr(wfx_ctx *ctx, const double *env, uint64_t n, wfx_dev_scalars *d_scal, uint8_t *out, int n1, int n0);
int wfx_dev_sync_pick_precomputed(wfx_ctx *ctx, uint64_t n, int n1, int n0, int64_t mindistance, double frame_samples, int width,
                                  wfx_dev_scalars *d_scal);
int wfx_dev_sync_pick(wfx_ctx *ctx, const uint8_t *d, uint64_t n, int n1, int n0,
                      int64_t mindistance, double frame_samples, int width,
                      wfx_dev_scalars *d_scal);
// mirror (optional): pinned host copy of the scalars, written by the kernel itself (saves the D2H blit launch)
// hdr (optional): {bytes, width} header of an exported image, written by the kernel (room = bytes available behind it)
int wfx_dev_image(wfx_ctx *ctx, const uint8_t *d, uint64_t n, int w, int h_max,
                  const wfx_dev_scalars *d_scal, uint8_t *img, wfx_dev_scalars *mirror = nullptr, long long *hdr = nullptr, long long room = 0);
int wfx_dev_image_rows(wfx_ctx *ctx, const uint8_t *d, uint64_t g0, uint64_t start, int w, int h_total, int y0, int rows,
                       uint8_t *img);
int wfx_dev_notch_fir_only(wfx_ctx *ctx, const void *in, int in_kind, uint64_t n, const double b[3], const double a[3], double *out,
                           int edge_flags, const double *ext18 = nullptr);
int wfx_dev_select_level(wfx_ctx *ctx, const double *env, uint64_t n, int level, const uint64_t prefix[4], unsigned *hist);
// The radix select in the steps the sharded decode separates with collectives (the histograms are summed over the ranks between them,
// the candidate lists gathered): workspace words [0, 2048) = level-0 histogram, [2048, 5 * 2048) = level-1 histograms.
#define WFX_SEL_H1_OFFSET 2048
#define WFX_SEL_H1_WORDS (4 * 2048)
int wfx_dev_select_sharded_ws(wfx_ctx *ctx, unsigned **ws);
int wfx_dev_select_l1(wfx_ctx *ctx, const double *env, uint64_t n, const uint64_t ranks[4], unsigned *ws, wfx_dev_scalars *d_scal);
// appends this rank's candidates to block[4][cap] (keys) and records the four counts behind them: block = {u64 keys[4][cap]; u32 count[4]; u32 pad[4]}
int wfx_dev_select_compact_block(wfx_ctx *ctx, const double *env, uint64_t n, unsigned *ws, wfx_dev_scalars *d_scal, void *block, uint64_t cap);
// all ranks' blocks (gathered) -> one candidate list per query; finishes the select; *overflow (device) += 1 when a rank's list did not fit
int wfx_dev_select_finish_blocks(wfx_ctx *ctx, unsigned *ws, wfx_dev_scalars *d_scal, const void *blocks, int nblocks, uint64_t cap, double gamma_lo,
                                 double gamma_hi, unsigned *overflow);
static inline size_t wfx_select_block_bytes(uint64_t cap) { return (size_t)cap * 32 + 32; }
int wfx_dev_add_u64(wfx_ctx *ctx, unsigned long long *dst, const unsigned long long *src, int n);

// wfx_mrfft.hip
// decomposition of a 13-smooth length into radix pairs with a register-resident pass (ascending radix); false if none
bool wfx_mr_pair_plan(long long L, std::vector<std::pair<int, int>> &pairs);
struct wfx_mr_radix {
    int R, ra, rb;              // ra = rb = 0: no register-resident pair pass for R (per-prime LDS stages)
};
// passes for any 13-smooth length (pairs where possible); descriptor and launch of either kind of pass
bool wfx_mr_general_plan(long long L, std::vector<wfx_mr_radix> &out);
void wfx_mr_general_desc(mr_pass_desc &d, const wfx_mr_radix &r, long long P, long long ncol, long long L);
int wfx_mr_launch(wfx_ctx *ctx, const mr_pass_desc &d, const cplx *tw, int in_mode, int out_mode, int dir, const void *src, cplx *dst);
bool wfx_mr_is_pair(int ra, int rb);
void wfx_mr_all_pairs(std::vector<std::pair<int, int>> &out);
// descriptor of one pair pass with the plain index maps (dist fields cleared, Ptw = P, Ltw = L)
void wfx_mr_pair_desc(mr_pass_desc &d, int ra, int rb, long long P, long long ncol, long long L);
// two-level twiddle table of W_mod = exp(-2 pi i / mod): lo[2048], hi[(mod >> 11) + 2]; returns the number of cplx entries written at `base`
size_t wfx_mr_table_elems(long long mod);
int wfx_mr_fill_table(wfx_ctx *ctx, cplx *base, long long mod);       // lo = base, hi = base + 2048
// launch one pair pass.  in_mode 0 complex, 1 packed reals swapped on load, 2 int16 pairs; out_mode 1: times the Hilbert spectrum; dir 1: inverse
int wfx_mr_launch_pair(wfx_ctx *ctx, const mr_pass_desc &d, const cplx *tw, int in_mode, int out_mode, int dir, const void *src, cplx *dst);
bool wfx_mr_supported(uint64_t L);
void wfx_mr_release(wfx_ctx *ctx);
int wfx_dev_hilbert_conv_mr(wfx_ctx *ctx, const double *x, uint64_t n, cplx **V_out);
// even n whose half is NOT 13-smooth: the same packed convolution zero-padded to the cheapest 13-smooth M >= n - 1 with a
// radix-pair plan (at most a few per cent above n - 1); *handled = 0 when no such M exists (tiny n): the caller pads to 2^k
int wfx_dev_hilbert_conv_mr_padded(wfx_ctx *ctx, const double *x, uint64_t n, cplx **V_out, int *handled);
// odd n: the real sequence against scipy's real kernel as two PACKED transforms of M/2 >= n points (13-smooth) and one glue pass;
// H[i] = ((double *)*V_out)[i].  x[n] must be readable and ZERO (the last packed pair).  *handled = 0 when n is too small
int wfx_dev_hilbert_conv_mr_real(wfx_ctx *ctx, const double *x, uint64_t n, cplx **V_out, int *handled);
long long wfx_mr_padded_length(long long min_len);
void wfx_mr_smooth_numbers(long long lo, long long hi, std::vector<long long> &out);       // ascending 13-smooth numbers in [lo, hi]
int wfx_dev_hilbert_kernel_rows(wfx_ctx *ctx, cplx *dst, long long p0, long long count, long long N, long long M);
int wfx_dev_hilbert_kernel_rows_real(wfx_ctx *ctx, cplx *dst, long long p0, long long count, long long N, long long Mh);       // odd N: pairs (g[2q], g[2q+1]) / 2 Mh
bool wfx_mr_resample_supported(uint64_t n0, uint64_t num);
int wfx_dev_resample_mr(wfx_ctx *ctx, const double *x, uint64_t n0, uint64_t num, double *out, bool x_is_i16 = false);
// any lengths: two chirp-z transforms on the mixed-radix passes (x: float64 or int16 samples); *handled = 0 when the lengths are out
// of its range (too short, no plan) and nothing was enqueued
int wfx_dev_resample_czt(wfx_ctx *ctx, const void *x, bool x_is_i16, uint64_t n0, uint64_t num, double *out, int *handled);
bool wfx_czt_resample_supported(wfx_ctx *ctx, uint64_t n0, uint64_t num);

int wfx_dev_export_header(wfx_ctx *ctx, const wfx_dev_scalars *d_scal, long long fixed, int width, long long room, long long *hdr);

// wfx_polyphase.hip
// nbatch > 1: that many equally shaped jobs in one launch -- member b reads in + b * in_stride frames (a multiple of 16 bytes) and
// writes out + b * out_stride
int wfx_dev_decimate_fir64(wfx_ctx *ctx, const void *in, int in_kind, uint64_t n_in, int64_t first, int M, const double *coef, int ntaps,
                           double *out, uint64_t n_out, int fix_shift, int *exact_out, int nbatch = 1, uint64_t in_stride = 0, uint64_t out_stride = 0);

// wfx_fmm.hip: |x + i H| (out_env) or H = imag(scipy.signal.hilbert(x)) for even n by near field + fast multipole far field; *handled = 0
// for lengths it does not take (odd, short)
int wfx_dev_hilbert_fmm(wfx_ctx *ctx, const double *x, uint64_t n, double *out, int out_mode, unsigned *l0hist, int *handled);
int wfx_dev_resample_fmm(wfx_ctx *ctx, const double *x, uint64_t n0, uint64_t num, double *y, int *handled);
// one capture over several GPUs by the multipole form (wfx_shard.hip plan 3; the kernels are wfx_fmm.hip's)
struct wfx_fmm_shard_geo {
    int L, ltop, lg, smax;       // levels; the leaf workgroups' roots; the gather level (ranks own whole boxes of it); the largest leaf
};
int wfx_fmm_shard_geometry(uint64_t n, wfx_fmm_shard_geo *geo);      // host only; -1: no multipole form for n
long long wfx_fmm_leaf_first_host(uint64_t n, int L, long long k);   // first sample of leaf k
int wfx_fmm_shard_weights(wfx_ctx *ctx, uint64_t n, int lev, long long b, double **ptr);
int wfx_fmm_shard_edges(wfx_ctx *ctx, uint64_t n, long long wg, double **ptr);
int wfx_fmm_shard_up(wfx_ctx *ctx, const void *raw, long long raw_index0, int raw_kind, const double b[3], const double a[3], const double *ext18, double *audio,
                     long long audio_index0, uint64_t n, long long gb_lo, long long gb_hi, wfx_dev_scalars *clear);
int wfx_fmm_shard_down(wfx_ctx *ctx, const double *audio, long long audio_index0, uint64_t n, long long gb_lo, long long gb_hi, double *env, long long env_index0,
                       unsigned *l0hist);
int wfx_fmm_shard_seams(wfx_ctx *ctx, uint64_t n, long long gb_lo, long long gb_hi, double *env, long long env_index0, unsigned *l0hist);
// the resampler's multipole form, sharded (wfx_fmm.hip; plan 3 in front of a resampler)
int wfx_rs_shard_geometry(uint64_t n0, uint64_t num, wfx_fmm_shard_geo *geo);
int wfx_rs_shard_up(wfx_ctx *ctx, const double *x, long long x_index0, uint64_t n0, uint64_t num, long long gb_lo, long long gb_hi, double **gsum);
int wfx_rs_shard_down(wfx_ctx *ctx, const double *x, long long x_index0, uint64_t n0, uint64_t num, long long gb_lo, long long gb_hi, double *y, long long y_index0);
int wfx_dev_notch_hilbert_fmm(wfx_ctx *ctx, const void *in, int in_kind, uint64_t n, const double b[3], const double a[3], const double *ext18, double *audio,
                              double *env, unsigned *l0hist, wfx_dev_scalars *clear, int *handled);      // a6 + a7 in one chain of kernels (wfx_fmm.hip)   // out_mode 0: H, 1: |x + iH|, 2: its 5-tap median + level-0 histogram of the select

// wfx_ingest.hip: the streaming form of the ingest (factor 32, int16 frames, taps on a 2^-s grid) with the float64 stage behind it
// fused (factor2 = 2 or 3; 0: stage 1 alone).  *handled = 0 and nothing enqueued when the shapes are not its own
int wfx_dev_ingest_stream(wfx_ctx *ctx, const void *in, int in_kind, uint64_t n_in, int factor, const double *coef1, int ntaps1, int fix_shift,
                          int factor2, const double *coef2, int ntaps2, double *out, uint64_t n_out, int nbatch, uint64_t in_stride,
                          uint64_t out_stride, int *handled);

// GB/s of a plain read of `bytes` at `dev` (best of `reps` launches, HIP events): the box's read-stream ceiling, measured beside the ingest
int wfx_dev_read_rate(wfx_ctx *ctx, const void *dev, uint64_t bytes, int reps, double *gbs);
int wfx_dev_stream_rate(wfx_ctx *ctx, const void *dev, uint64_t bytes, double *out, int reps, double *gbs);

// wfx_comm.hip: the communicator behind the sharded decode (RCCL bound directly, or every rank in this process)
struct wfx_xfer {            // one message pair of a personalised exchange; several entries per peer are matched in order
    int peer;
    const void *send;
    size_t send_bytes;
    void *recv;
    size_t recv_bytes;
};
int wfx_comm_world(const wfx_comm *c);
int wfx_comm_rank(const wfx_comm *c);
bool wfx_comm_is_local(const wfx_comm *c);
void wfx_comm_label(wfx_comm *c, const char *name);      // names the next collective in the wire statistics (wfx_comm_wire_stats)
int wfx_comm_exchange(wfx_comm *c, wfx_ctx *ctx, const wfx_xfer *list, int n);
// an exchange that may overlap the caller's next kernels (RCCL: on the communicator's own stream; others: completed on return);
// wfx_comm_wait orders the context's stream behind it.  slot in [0, 64)
int wfx_comm_exchange_async(wfx_comm *c, wfx_ctx *ctx, const wfx_xfer *list, int n, int slot);
int wfx_comm_wait(wfx_comm *c, wfx_ctx *ctx, int slot);
unsigned long long wfx_comm_async_count(const wfx_comm *c);
int wfx_comm_allreduce_u32(wfx_comm *c, wfx_ctx *ctx, unsigned *buf, size_t count);
int wfx_comm_allgather(wfx_comm *c, wfx_ctx *ctx, const void *send, void *recv, size_t bytes_per_rank);

