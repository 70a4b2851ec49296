// C ABI of libwefax_hip.so: stage entry points (host buffers in/out) and the fused
// device-resident decode.  See include/wefax_hip.h for the contract and the
// reference lines each entry point replaces.
#include <cstdlib>
#include <cstring>

#include "wfx_internal.h"
#include <atomic>
#include <thread>
#include <vector>
#include <unistd.h>

#define CHECK_CTX(ctx)                                                        \
    do {                                                                      \
        if (!(ctx)) return wfx_fail(nullptr, WFX_ERR_BAD_ARG, "null context"); \
        (void)hipSetDevice((ctx)->device);                                        \
    } while (0)

static int h2d(wfx_ctx *ctx, void *dst, const void *src, size_t bytes)
{
    if (bytes == 0) return 0;
    WFX_HIP(ctx, hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, ctx->stream));
    return 0;
}

static int d2h_sync(wfx_ctx *ctx, void *dst, const void *src, size_t bytes)
{
    if (bytes) WFX_HIP(ctx, hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, ctx->stream));
    WFX_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return 0;
}

static int ensure_scal(wfx_ctx *ctx)
{
    WFX_TRY(wfx_reserve(ctx, ctx->b_scal, sizeof(wfx_dev_scalars)));
    return 0;
}

extern "C" {

// ---- a4 ---------------------------------------------------------------------
int wfx_merge_channels(wfx_ctx *ctx, const int16_t *lr, size_t n, double *out)
{
    CHECK_CTX(ctx);
    if (!lr || !out) return wfx_fail(ctx, WFX_ERR_BAD_ARG, "null buffer");
    ctx->ran = false;
    if (n == 0) return 0;
    WFX_TRY(wfx_reserve(ctx, ctx->b_in, n * 4));
    WFX_TRY(wfx_reserve(ctx, ctx->b_x, n * 8));
    WFX_TRY(h2d(ctx, ctx->b_in.p, lr, n * 4));
    WFX_TRY(wfx_dev_merge(ctx, (const int16_t *)ctx->b_in.p, n, (double *)ctx->b_x.p));
    return d2h_sync(ctx, out, ctx->b_x.p, n * 8);
}

int wfx_merge_channels_any(wfx_ctx *ctx, const void *lr, int in_kind, size_t n, double *out)
{
    CHECK_CTX(ctx);
    if (!lr || !out) return wfx_fail(ctx, WFX_ERR_BAD_ARG, "null buffer");
    if (!wfx_kind_is_stereo(in_kind)) return wfx_fail(ctx, WFX_ERR_BAD_ARG, "merge: input kind %d has one channel", in_kind);
    ctx->ran = false;
    if (n == 0) return 0;
    const size_t fb = wfx_kind_frame_bytes(in_kind);
    WFX_TRY(wfx_reserve(ctx, ctx->b_in, n * fb));
    WFX_TRY(wfx_reserve(ctx, ctx->b_x, n * 8));
    WFX_TRY(h2d(ctx, ctx->b_in.p, lr, n * fb));
    WFX_TRY(wfx_dev_merge_any(ctx, ctx->b_in.p, in_kind, n, (double *)ctx->b_x.p));
    return d2h_sync(ctx, out, ctx->b_x.p, n * 8);
}

// ---- a5 ---------------------------------------------------------------------
int wfx_resample(wfx_ctx *ctx, const double *x, size_t n0, size_t num, double *out)
{
    CHECK_CTX(ctx);
    if (!x || !out) return wfx_fail(ctx, WFX_ERR_BAD_ARG, "null buffer");
    if (n0 == 0 || num == 0) return wfx_fail(ctx, WFX_ERR_BAD_ARG, "resample: empty input or output");
    ctx->ran = false;
    WFX_TRY(wfx_reserve(ctx, ctx->b_x, n0 * 8));
    WFX_TRY(wfx_reserve(ctx, ctx->b_tmp, num * 8));
    WFX_TRY(h2d(ctx, ctx->b_x.p, x, n0 * 8));
    WFX_TRY(wfx_dev_resample_fft(ctx, (const double *)ctx->b_x.p, n0, num, (double *)ctx->b_tmp.p));
    return d2h_sync(ctx, out, ctx->b_tmp.p, num * 8);
}

// ---- a6 ---------------------------------------------------------------------
int wfx_notch_filtfilt(wfx_ctx *ctx, const void *in, int in_kind, size_t n, const double b[3], const double a[3], double *out)
{
    CHECK_CTX(ctx);
    if (!in || !out || !b || !a) return wfx_fail(ctx, WFX_ERR_BAD_ARG, "null buffer");
    if (in_kind != WFX_IN_I16_MONO && in_kind != WFX_IN_F64_MONO)
        return wfx_fail(ctx, WFX_ERR_BAD_ARG, "notch: input must be int16 mono or float64");
    ctx->ran = false;
    const size_t esz = in_kind == WFX_IN_I16_MONO ? 2 : 8;
    WFX_TRY(wfx_reserve(ctx, ctx->b_x, n * esz + 16));
    WFX_TRY(wfx_reserve(ctx, ctx->b_audio, n * 8 + 16));
    WFX_TRY(h2d(ctx, ctx->b_x.p, in, n * esz));
    WFX_TRY(wfx_dev_notch(ctx, ctx->b_x.p, in_kind, n, b, a, (double *)ctx->b_audio.p));
    return d2h_sync(ctx, out, ctx->b_audio.p, n * 8);
}

int wfx_notch_filtfilt_ext(wfx_ctx *ctx, const void *in, int in_kind, size_t n, const double b[3], const double a[3], const double ext_left[9],
                           const double ext_right[9], double *out)
{
    CHECK_CTX(ctx);
    if (!in || !out || !b || !a || !ext_left || !ext_right) return wfx_fail(ctx, WFX_ERR_BAD_ARG, "null buffer");
    if (in_kind != WFX_IN_I16_MONO && in_kind != WFX_IN_F64_MONO) return wfx_fail(ctx, WFX_ERR_BAD_ARG, "notch: input must be int16 mono or float64");
    ctx->ran = false;
    const size_t esz = in_kind == WFX_IN_I16_MONO ? 2 : 8;
    double ext18[18];
    for (int i = 0; i < 9; ++i) {
        ext18[i] = ext_left[i];
        ext18[9 + i] = ext_right[i];
    }
    WFX_TRY(wfx_reserve(ctx, ctx->b_x, n * esz + 16));
    WFX_TRY(wfx_reserve(ctx, ctx->b_audio, n * 8 + 16));
    WFX_TRY(h2d(ctx, ctx->b_x.p, in, n * esz));
    WFX_TRY(wfx_dev_notch(ctx, ctx->b_x.p, in_kind, n, b, a, (double *)ctx->b_audio.p, nullptr, nullptr, ext18));
    return d2h_sync(ctx, out, ctx->b_audio.p, n * 8);
}

// ---- a7 ---------------------------------------------------------------------
static int analytic_env_dev(wfx_ctx *ctx, const double *x, uint64_t n, int mode, double *env_raw, double *env, unsigned *l0hist = nullptr)
{
    if (mode == WFX_HILBERT_FFT || mode == WFX_HILBERT_FFT_POW2) {
        ctx->force_pow2 = mode == WFX_HILBERT_FFT_POW2;
        return wfx_dev_hilbert_envmed_fft(ctx, x, n, env, l0hist);
    }
    else if (mode == WFX_HILBERT_FMM) {
        // near field + fast multipole far field (wfx_fmm.hip); lengths it does not take (odd, short) run the transform path
        // (envelope, 5-tap median and the select's level-0 histogram are fused into its leaf kernel)
        int handled = 0;
        WFX_TRY(wfx_dev_hilbert_fmm(ctx, x, n, env, 2, l0hist, &handled));
        if (!handled) {
            ctx->force_pow2 = false;
            return wfx_dev_hilbert_envmed_fft(ctx, x, n, env, l0hist);
        }
        return 0;
    }
    else if (mode == WFX_HILBERT_BLUESTEIN)
        WFX_TRY(wfx_dev_hilbert_env_bluestein(ctx, x, n, env_raw));
    else
        return wfx_fail(ctx, WFX_ERR_BAD_ARG, "unknown hilbert mode %d", mode);
    return wfx_dev_median5(ctx, env_raw, n, env, l0hist);
}

int wfx_analytic_env(wfx_ctx *ctx, const double *x, size_t n, int hilbert_mode, double *env_out)
{
    CHECK_CTX(ctx);
    if (!x || !env_out) return wfx_fail(ctx, WFX_ERR_BAD_ARG, "null buffer");
    if (n == 0) return wfx_fail(ctx, WFX_ERR_BAD_ARG, "N must be positive.");
    ctx->ran = false;
    WFX_TRY(wfx_reserve(ctx, ctx->b_audio, n * 8 + 16));
    WFX_TRY(wfx_reserve(ctx, ctx->b_envraw, n * 8));
    WFX_TRY(wfx_reserve(ctx, ctx->b_env, n * 8 + 64));
    WFX_TRY(h2d(ctx, ctx->b_audio.p, x, n * 8));
    WFX_TRY(analytic_env_dev(ctx, (const double *)ctx->b_audio.p, n, hilbert_mode, (double *)ctx->b_envraw.p, (double *)ctx->b_env.p));
    return d2h_sync(ctx, env_out, ctx->b_env.p, n * 8);
}

// ---- a8 ---------------------------------------------------------------------
int wfx_order_stats(wfx_ctx *ctx, const double *env, size_t n, const uint64_t *ranks, int nranks, double *out)
{
    CHECK_CTX(ctx);
    if (!env || !ranks || !out) return wfx_fail(ctx, WFX_ERR_BAD_ARG, "null buffer");
    if (n == 0 || nranks < 1) return wfx_fail(ctx, WFX_ERR_BAD_ARG, "order_stats: empty input");
    ctx->ran = false;
    WFX_TRY(ensure_scal(ctx));
    WFX_TRY(wfx_reserve(ctx, ctx->b_env, n * 8 + 64));
    WFX_TRY(h2d(ctx, ctx->b_env.p, env, n * 8));
    wfx_dev_scalars *ds = (wfx_dev_scalars *)ctx->b_scal.p;
    for (int base = 0; base < nranks; base += 4) {
        uint64_t r[4];
        for (int q = 0; q < 4; ++q) r[q] = ranks[base + q < nranks ? base + q : nranks - 1];
        WFX_TRY(wfx_dev_select(ctx, (const double *)ctx->b_env.p, n, r, ds));
        WFX_TRY(d2h_sync(ctx, ctx->h_scal, ds, sizeof(wfx_dev_scalars)));
        for (int q = 0; q < 4 && base + q < nranks; ++q) out[base + q] = ctx->h_scal->sel_value[q];
    }
    return 0;
}

int wfx_quantise(wfx_ctx *ctx, const double *env, size_t n, double low, double high, uint8_t *out, uint64_t *nan_count)
{
    CHECK_CTX(ctx);
    if (!env || !out) return wfx_fail(ctx, WFX_ERR_BAD_ARG, "null buffer");
    ctx->ran = false;
    if (nan_count) *nan_count = 0;
    if (n == 0) return 0;
    WFX_TRY(ensure_scal(ctx));
    WFX_TRY(wfx_reserve(ctx, ctx->b_env, n * 8 + 64));
    WFX_TRY(wfx_reserve(ctx, ctx->b_dig, n + 64));
    WFX_TRY(h2d(ctx, ctx->b_env.p, env, n * 8));
    memset(ctx->h_scal, 0, sizeof(wfx_dev_scalars));
    ctx->h_scal->low = low;
    ctx->h_scal->high = high;
    wfx_dev_scalars *ds = (wfx_dev_scalars *)ctx->b_scal.p;
    WFX_TRY(h2d(ctx, ds, ctx->h_scal, sizeof(wfx_dev_scalars)));
    WFX_TRY(wfx_dev_quantise(ctx, (const double *)ctx->b_env.p, n, ds, (uint8_t *)ctx->b_dig.p, ds));
    WFX_TRY(d2h_sync(ctx, out, ctx->b_dig.p, n));
    WFX_TRY(d2h_sync(ctx, ctx->h_scal, ds, sizeof(wfx_dev_scalars)));
    if (nan_count) *nan_count = ctx->h_scal->nan_count;
    return 0;
}

// ---- a9 ---------------------------------------------------------------------
int wfx_sync_corr(wfx_ctx *ctx, const uint8_t *d, size_t n, int n1, int n0, int32_t *corr_out)
{
    CHECK_CTX(ctx);
    if (!d || !corr_out) return wfx_fail(ctx, WFX_ERR_BAD_ARG, "null buffer");
    ctx->ran = false;
    const size_t L = (size_t)(2 * n1 + n0);
    if (n <= L) return 0;
    WFX_TRY(wfx_reserve(ctx, ctx->b_dig, n + 64));
    WFX_TRY(wfx_reserve(ctx, ctx->b_corr, n * 4 + 256));
    WFX_TRY(h2d(ctx, ctx->b_dig.p, d, n));
    WFX_TRY(wfx_dev_sync_corr(ctx, (const uint8_t *)ctx->b_dig.p, n, n1, n0, (int32_t *)ctx->b_corr.p));
    return d2h_sync(ctx, corr_out, ctx->b_corr.p, (n - L) * 4);
}

int wfx_sync_peaks(wfx_ctx *ctx, const uint8_t *d, size_t n, int n1, int n0, int64_t mindistance, int64_t *peak_pos, int64_t *first_pos,
                   int *npeaks, int *hit_limit)
{
    CHECK_CTX(ctx);
    if (!d || !peak_pos || !first_pos || !npeaks || !hit_limit) return wfx_fail(ctx, WFX_ERR_BAD_ARG, "null buffer");
    ctx->ran = false;
    WFX_TRY(ensure_scal(ctx));
    WFX_TRY(wfx_reserve(ctx, ctx->b_dig, n + 64));
    WFX_TRY(h2d(ctx, ctx->b_dig.p, d, n));
    wfx_dev_scalars *ds = (wfx_dev_scalars *)ctx->b_scal.p;
    WFX_HIP(ctx, hipMemsetAsync(ds, 0, sizeof(wfx_dev_scalars), ctx->stream));
    WFX_TRY(wfx_dev_sync_pick(ctx, (const uint8_t *)ctx->b_dig.p, n, n1, n0, mindistance, 0.0, 0, ds));
    WFX_TRY(d2h_sync(ctx, ctx->h_scal, ds, sizeof(wfx_dev_scalars)));
    *npeaks = ctx->h_scal->npeaks;
    *hit_limit = ctx->h_scal->hit_limit;
    for (int i = 0; i < ctx->h_scal->npeaks; ++i) {
        peak_pos[i] = ctx->h_scal->peak_pos[i];
        first_pos[i] = ctx->h_scal->first_pos[i];
    }
    return 0;
}

// ---- a10 --------------------------------------------------------------------
int wfx_lines_to_image(wfx_ctx *ctx, const uint8_t *d, size_t n, size_t start, int w, uint8_t *img)
{
    CHECK_CTX(ctx);
    if (!d || !img) return wfx_fail(ctx, WFX_ERR_BAD_ARG, "null buffer");
    if (w <= 0 || start > n) return wfx_fail(ctx, WFX_ERR_BAD_ARG, "lines_to_image: bad width or start");
    ctx->ran = false;
    const int h = (int)((n - start) / (size_t)w);
    if (h == 0) return 0;
    WFX_TRY(ensure_scal(ctx));
    WFX_TRY(wfx_reserve(ctx, ctx->b_dig, n + 64));
    WFX_TRY(wfx_reserve(ctx, ctx->b_img, (size_t)w * 4 * h));
    WFX_TRY(h2d(ctx, ctx->b_dig.p, d, n));
    memset(ctx->h_scal, 0, sizeof(wfx_dev_scalars));
    ctx->h_scal->start_frame = (long long)start;
    ctx->h_scal->height = h;
    wfx_dev_scalars *ds = (wfx_dev_scalars *)ctx->b_scal.p;
    WFX_TRY(h2d(ctx, ds, ctx->h_scal, sizeof(wfx_dev_scalars)));
    WFX_TRY(wfx_dev_image(ctx, (const uint8_t *)ctx->b_dig.p, n, w, h, ds, (uint8_t *)ctx->b_img.p));
    return d2h_sync(ctx, img, ctx->b_img.p, (size_t)w * 4 * h);
}

// ---- fused decode ---------------------------------------------------------------
static size_t in_bytes(const wfx_decode_params *p)
{
    return (size_t)p->n0 * wfx_kind_frame_bytes(p->in_kind);
}

static int check_params(wfx_ctx *ctx, const wfx_decode_params *p)
{
    if (p->in_kind < WFX_IN_I16_MONO || p->in_kind > WFX_IN_F32_STEREO || p->in_kind == WFX_IN_F32_MONO)
        return wfx_fail(ctx, WFX_ERR_BAD_ARG, "bad in_kind %d", p->in_kind);
    if (p->n0 == 0 || p->n == 0) return wfx_fail(ctx, WFX_ERR_BAD_ARG, "empty capture");
    if (!p->resample && p->n != p->n0) return wfx_fail(ctx, WFX_ERR_BAD_ARG, "n != n0 without resampling");
    if (p->n <= 9) return wfx_fail(ctx, WFX_ERR_BAD_ARG, "The length of the input vector x must be greater than padlen, which is 9.");
    if (p->n > (1ull << 31)) return wfx_fail(ctx, WFX_ERR_BAD_ARG, "capture too long");
    if (p->width <= 0) return wfx_fail(ctx, WFX_ERR_BAD_ARG, "bad line width");
    if (p->rank_lo[0] >= p->n || p->rank_lo[1] >= p->n || p->rank_hi[0] >= p->n || p->rank_hi[1] >= p->n)
        return wfx_fail(ctx, WFX_ERR_BAD_ARG, "percentile rank out of range");
    if (p->hilbert_mode != WFX_HILBERT_FFT && p->hilbert_mode != WFX_HILBERT_BLUESTEIN && p->hilbert_mode != WFX_HILBERT_FFT_POW2 &&
        p->hilbert_mode != WFX_HILBERT_FMM)
        return wfx_fail(ctx, WFX_ERR_BAD_ARG, "unknown hilbert mode %d", p->hilbert_mode);
    return 0;
}

int wfx_decode_upload(wfx_ctx *ctx, const void *host_in, const wfx_decode_params *p)
{
    CHECK_CTX(ctx);
    if (!host_in || !p) return wfx_fail(ctx, WFX_ERR_BAD_ARG, "null argument");
    WFX_TRY(check_params(ctx, p));
    ctx->dp = *p;
    ctx->have_input = false;
    ctx->ran = false;
    ctx->ext_in = nullptr;
    const size_t nb = in_bytes(p);
    WFX_TRY(wfx_reserve(ctx, ctx->b_in, nb + 64));
    WFX_TRY(h2d(ctx, ctx->b_in.p, host_in, nb));
    WFX_HIP(ctx, hipStreamSynchronize(ctx->stream));
    ctx->have_input = true;
    return 0;
}

// wefax.py:349 (scipy.io.wavfile.read) + the upload, as ONE pipeline for the capture formats that reach the device as they lie in
// the file (16-bit PCM, one or two channels): a few threads pread() slices of the data chunk into the caller's page-locked staging
// buffer, and every slice goes to the device by DMA the moment it is complete -- the copy out of the page cache and the transfer over
// PCIe overlap instead of adding up (the 345 MB wav of BASELINE configs[2]: 13 + 7 ms one after the other).
int wfx_decode_upload_fd(wfx_ctx *ctx, int fd, uint64_t file_offset, void *pinned, size_t pinned_bytes, const wfx_decode_params *p)
{
    CHECK_CTX(ctx);
    if (fd < 0 || !pinned || !p) return wfx_fail(ctx, WFX_ERR_BAD_ARG, "null argument");
    WFX_TRY(check_params(ctx, p));
    if (p->in_kind != WFX_IN_I16_MONO && p->in_kind != WFX_IN_I16_STEREO)
        return wfx_fail(ctx, WFX_ERR_BAD_ARG, "upload from a file: 16-bit PCM captures only (kind %d)", p->in_kind);
    ctx->dp = *p;
    ctx->have_input = false;
    ctx->ran = false;
    ctx->ext_in = nullptr;
    const size_t nb = in_bytes(p);
    if (pinned_bytes < nb) return wfx_fail(ctx, WFX_ERR_BAD_ARG, "upload from a file: staging buffer of %zu bytes for %zu", pinned_bytes, nb);
    WFX_TRY(wfx_reserve(ctx, ctx->b_in, nb + 64));
    const size_t slice = (size_t)4 << 20;
    const size_t nsl = (nb + slice - 1) / slice;
    unsigned nthr = std::thread::hardware_concurrency();
    nthr = nthr < 1 ? 1 : nthr > 16 ? 16 : nthr;
    if (nthr > nsl) nthr = (unsigned)(nsl ? nsl : 1);
    std::vector<std::atomic<int>> done(nsl ? nsl : 1);
    for (auto &d : done) d.store(0, std::memory_order_relaxed);
    std::atomic<size_t> next{0};
    std::atomic<int> failed{0};
    auto reader = [&]() {
        for (;;) {
            const size_t k = next.fetch_add(1);
            if (k >= nsl || failed.load()) return;
            size_t off = k * slice;
            const size_t end = off + slice < nb ? off + slice : nb;
            while (off < end) {
                const ssize_t got = pread(fd, (unsigned char *)pinned + off, end - off, (off_t)(file_offset + off));
                if (got <= 0) {
                    failed.store(1);
                    done[k].store(2, std::memory_order_release);
                    return;
                }
                off += (size_t)got;
            }
            done[k].store(1, std::memory_order_release);
        }
    };
    // (a std::thread constructor may throw std::system_error at the process's thread limit: nothing may cross the C boundary, so
    // whatever could not be started is made up for by this thread, which then reads the remaining slices itself -- ADVICE r5)
    std::vector<std::thread> pool;
    try {
        for (unsigned i = 0; i < nthr; ++i) pool.emplace_back(reader);
    } catch (...) {
    }
    if (pool.empty()) reader();
    int rc = 0;
    for (size_t k = 0; k < nsl && rc == 0; ++k) {
        int st;
        while ((st = done[k].load(std::memory_order_acquire)) == 0) std::this_thread::yield();
        if (st != 1) break;
        const size_t off = k * slice, len = off + slice < nb ? slice : nb - off;
        if (hipMemcpyAsync((unsigned char *)ctx->b_in.p + off, (const unsigned char *)pinned + off, len, hipMemcpyHostToDevice, ctx->stream) != hipSuccess) rc = WFX_ERR_HIP;
    }
    if (rc != 0) failed.store(1);
    for (auto &th : pool) th.join();
    if (rc != 0) return wfx_fail(ctx, WFX_ERR_HIP, "upload from a file: DMA failed");
    if (failed.load()) {
        (void)hipStreamSynchronize(ctx->stream);
        return wfx_fail(ctx, WFX_ERR_SHORT_FILE, "Incomplete wav file: data chunk is shorter than its header says");
    }
    WFX_HIP(ctx, hipStreamSynchronize(ctx->stream));
    ctx->have_input = true;
    return 0;
}

int wfx_decode_reload(wfx_ctx *ctx, const void *host_in, size_t bytes, const double *ext_left, const double *ext_right)
{
    CHECK_CTX(ctx);
    if (!host_in) return wfx_fail(ctx, WFX_ERR_BAD_ARG, "null argument");
    if (!ctx->have_input || ctx->ext_in) return wfx_fail(ctx, WFX_ERR_STATE, "decode_reload needs a capture uploaded with wfx_decode_upload");
    const size_t nb = in_bytes(&ctx->dp);
    if (bytes != nb)
        return wfx_fail(ctx, WFX_ERR_BAD_ARG, "decode_reload: the uploaded capture has %zu bytes (%llu frames of kind %d), %zu given", nb,
                        (unsigned long long)ctx->dp.n0, ctx->dp.in_kind, bytes);
    // results of the previous capture are no longer fetchable, and filtfilt's odd extension (evaluated by the host in the
    // file's own dtype for float64 hand-overs) belongs to the capture: the new one brings its own or has none
    ctx->ran = false;
    if (ext_left && ext_right) {
        ctx->dp.has_ext = 1;
        for (int i = 0; i < 9; ++i) {
            ctx->dp.ext_left[i] = ext_left[i];
            ctx->dp.ext_right[i] = ext_right[i];
        }
    } else {
        ctx->dp.has_ext = 0;
    }
    return h2d(ctx, ctx->b_in.p, host_in, nb);
}

int wfx_decode_fetch_async(wfx_ctx *ctx, int buffer_id, void *host_out, size_t bytes)
{
    CHECK_CTX(ctx);
    if (!ctx->ran) return wfx_fail(ctx, WFX_ERR_STATE, "no decode has been enqueued on this context");
    if (!host_out) return wfx_fail(ctx, WFX_ERR_BAD_ARG, "null buffer");
    const uint64_t n = ctx->dp.n;
    const void *src = nullptr;
    size_t nb = 0;
    switch (buffer_id) {
    case WFX_BUF_AUDIO: src = ctx->b_audio.p; nb = n * 8; break;
    case WFX_BUF_ENVELOPE: src = ctx->b_env.p; nb = n * 8; break;
    case WFX_BUF_DIGITAL: src = ctx->b_dig.p; nb = n; break;
    case WFX_BUF_IMAGE:
        src = ctx->img_in_ext ? (const void *)((const char *)ctx->ext_img + 16) : ctx->b_img.p;
        nb = (size_t)ctx->dp.width * 4 * (size_t)(n / (uint64_t)ctx->dp.width);
        break;
    default: return wfx_fail(ctx, WFX_ERR_BAD_ARG, "unknown buffer id %d", buffer_id);
    }
    if (bytes < nb) return wfx_fail(ctx, WFX_ERR_BAD_ARG, "fetch_async: %zu bytes needed, %zu given", nb, bytes);
    if (nb) WFX_HIP(ctx, hipMemcpyAsync(host_out, src, nb, hipMemcpyDeviceToHost, ctx->stream));
    return 0;
}

int wfx_decode_attach(wfx_ctx *ctx, const void *dev_in, const wfx_decode_params *p)
{
    CHECK_CTX(ctx);
    if (!dev_in || !p) return wfx_fail(ctx, WFX_ERR_BAD_ARG, "null argument");
    WFX_TRY(check_params(ctx, p));
    ctx->dp = *p;
    ctx->ran = false;
    ctx->ext_in = dev_in;
    ctx->have_input = true;
    return 0;
}

int wfx_decode_run(wfx_ctx *ctx)
{
    CHECK_CTX(ctx);
    if (!ctx->have_input) return wfx_fail(ctx, WFX_ERR_STATE, "decode_run before decode_upload");
    const wfx_decode_params &p = ctx->dp;
    const uint64_t n0 = p.n0, n = p.n;
    const int w = p.width;
    const int h_max = (int)(n / (uint64_t)w);
    WFX_TRY(ensure_scal(ctx));
    WFX_TRY(wfx_reserve(ctx, ctx->b_audio, n * 8 + 64));
    WFX_TRY(wfx_reserve(ctx, ctx->b_envraw, n * 8 + 64));
    WFX_TRY(wfx_reserve(ctx, ctx->b_env, n * 8 + 64));
    WFX_TRY(wfx_reserve(ctx, ctx->b_dig, n + 64));
    const size_t img_max = (size_t)w * 4 * (size_t)(h_max > 0 ? h_max : 1);
    const bool to_ext = ctx->ext_img != nullptr && ctx->ext_img_cap >= 16 + img_max;       // straight into the caller's {header, image} slot
    if (!to_ext) WFX_TRY(wfx_reserve(ctx, ctx->b_img, img_max));
    ctx->img_in_ext = to_ext;
    wfx_dev_scalars *ds = (wfx_dev_scalars *)ctx->b_scal.p;

    const void *cur = ctx->ext_in ? ctx->ext_in : ctx->b_in.p;      // caller-owned device input (wfx_decode_attach) or the uploaded copy
    int cur_kind = p.in_kind;
    if (wfx_kind_is_stereo(p.in_kind)) {
        WFX_TRY(wfx_reserve(ctx, ctx->b_x, n0 * 8));
        WFX_TRY(wfx_dev_merge_any(ctx, cur, p.in_kind, n0, (double *)ctx->b_x.p));
        cur = ctx->b_x.p;
        cur_kind = WFX_IN_F64_MONO;
    }
    ctx->force_pow2 = p.hilbert_mode == WFX_HILBERT_FFT_POW2;       // (also selects the padded resampler)
    bool resampled = false;
    wfx_fmm_shard_geo rs_geo;
    if (p.resample && p.hilbert_mode == WFX_HILBERT_FMM && wfx_rs_shard_geometry(n0, n, &rs_geo) == 0) {
        // the multipole route takes the resampler's multipole form too (wfx_fmm.hip rs_*): slower than the transform on ONE GPU, but the very
        // arithmetic of the sharded plan 3 -- a decode in this mode gives that plan's bytes
        if (cur_kind == WFX_IN_I16_MONO) {
            WFX_TRY(wfx_reserve(ctx, ctx->b_x, n0 * 8));
            WFX_TRY(wfx_dev_i16_to_f64(ctx, (const int16_t *)cur, n0, (double *)ctx->b_x.p));
            cur = ctx->b_x.p;
            cur_kind = WFX_IN_F64_MONO;
        }
        WFX_TRY(wfx_reserve(ctx, ctx->b_tmp, n * 8));
        int handled = 0;
        WFX_TRY(wfx_dev_resample_fmm(ctx, (const double *)cur, n0, n, (double *)ctx->b_tmp.p, &handled));
        if (handled) {
            cur = ctx->b_tmp.p;
            resampled = true;
        }
    }
    if (p.resample && !resampled) {
        // an int16 capture whose lengths the mixed-radix resampler takes is read in place by its first pass
        // (so is one of any other length by the chirp-z form's prologue)
        const bool in_place16 = cur_kind == WFX_IN_I16_MONO && !ctx->force_pow2 && !getenv("WFX_NO_I16_RESAMPLE") &&
                                (wfx_mr_resample_supported(n0, n) || wfx_czt_resample_supported(ctx, n0, n));
        if (cur_kind == WFX_IN_I16_MONO && !in_place16) {
            WFX_TRY(wfx_reserve(ctx, ctx->b_x, n0 * 8));
            WFX_TRY(wfx_dev_i16_to_f64(ctx, (const int16_t *)cur, n0, (double *)ctx->b_x.p));
            cur = ctx->b_x.p;
        }
        WFX_TRY(wfx_reserve(ctx, ctx->b_tmp, n * 8));
        WFX_TRY(wfx_dev_resample_fft(ctx, (const double *)cur, n0, n, (double *)ctx->b_tmp.p, in_place16));
        cur = ctx->b_tmp.p;
        cur_kind = WFX_IN_F64_MONO;
    }
    // the notch is the first kernel that can see the device scalars: its edge workgroup zeroes them
    bool cleared = false;
    double ext18[18];
    const bool use_ext = p.has_ext && !p.resample && p.in_kind != WFX_IN_I16_STEREO;       // (it describes the capture as handed over)
    for (int i = 0; i < 9; ++i) {
        ext18[i] = p.ext_left[i];
        ext18[9 + i] = p.ext_right[i];
    }
    unsigned *sel_ws = nullptr;
    WFX_TRY(wfx_dev_select_workspace(ctx, n, &sel_ws));           // level-0 histogram is fused into the envelope kernel
    int fused = 0;
    if (p.hilbert_mode == WFX_HILBERT_FMM)
        // a6 + a7 as one chain: the notch runs inside the multipole form's first kernel (which also zeroes the scalars), the median and the
        // histogram inside its last
        WFX_TRY(wfx_dev_notch_hilbert_fmm(ctx, cur, cur_kind, n, p.notch_b, p.notch_a, use_ext ? ext18 : nullptr, (double *)ctx->b_audio.p, (double *)ctx->b_env.p,
                                          sel_ws, ds, &fused));
    if (!fused) {
        WFX_TRY(wfx_dev_notch(ctx, cur, cur_kind, n, p.notch_b, p.notch_a, (double *)ctx->b_audio.p, ds, &cleared, use_ext ? ext18 : nullptr));
        if (!cleared) WFX_HIP(ctx, hipMemsetAsync(ds, 0, sizeof(wfx_dev_scalars), ctx->stream));
        WFX_TRY(analytic_env_dev(ctx, (const double *)ctx->b_audio.p, n, p.hilbert_mode, (double *)ctx->b_envraw.p, (double *)ctx->b_env.p, sel_ws));
    }
    const uint64_t ranks[4] = {p.rank_lo[0], p.rank_lo[1], p.rank_hi[0], p.rank_hi[1]};
    WFX_TRY(wfx_dev_percentiles_fused(ctx, (const double *)ctx->b_env.p, n, ranks, p.gamma_lo, p.gamma_hi, ds));
    WFX_TRY(wfx_dev_quantise_corr(ctx, (const double *)ctx->b_env.p, n, ds, (uint8_t *)ctx->b_dig.p, p.n1, p.n0_gap));   // a8 + correlation of a9
    WFX_TRY(wfx_dev_sync_pick_precomputed(ctx, n, p.n1, p.n0_gap, p.mindistance, p.frame_samples, w, ds));
    // the image kernel also writes the scalars to the pinned host copy wfx_decode_result reads
    if (to_ext)
        WFX_TRY(wfx_dev_image(ctx, (const uint8_t *)ctx->b_dig.p, n, w, h_max, ds, (uint8_t *)ctx->ext_img + 16, ctx->h_scal, (long long *)ctx->ext_img,
                              (long long)(ctx->ext_img_cap - 16)));
    else
        WFX_TRY(wfx_dev_image(ctx, (const uint8_t *)ctx->b_dig.p, n, w, h_max, ds, (uint8_t *)ctx->b_img.p, ctx->h_scal));
    ctx->ran = true;
    return 0;
}

int wfx_decode_result(wfx_ctx *ctx, wfx_decode_info *info)
{
    CHECK_CTX(ctx);
    if (!info) return wfx_fail(ctx, WFX_ERR_BAD_ARG, "null info");
    if (!ctx->ran) return wfx_fail(ctx, WFX_ERR_STATE, "decode_result before decode_run");
    WFX_HIP(ctx, hipStreamSynchronize(ctx->stream));
    const wfx_dev_scalars &s = *ctx->h_scal;
    memset(info, 0, sizeof *info);
    info->n = ctx->dp.n;
    info->low = s.low;
    info->high = s.high;
    info->nan_count = s.nan_count;
    info->npeaks = s.npeaks;
    info->hit_limit = s.hit_limit;
    info->no_group = s.no_group;
    info->n_phasing = s.n_phasing;
    info->start_frame = s.start_frame;
    info->width = ctx->dp.width;
    info->height = s.height;
    if (getenv("WFX_DEBUG"))
        fprintf(stderr, "[wfx] sync_pick: form %lld ops %lld chunks %lld pick_cycles %lld loop_cycles %lld total_cycles %lld | fused steps %lld: reads %lld\n",
                s.dbg[7], s.dbg[0], s.dbg[1], s.dbg[2], s.dbg[3], s.dbg[4], s.dbg[6], s.dbg[5]);
    for (int i = 0; i <= WFX_MAX_PEAKS; ++i) {
        info->peak_pos[i] = s.peak_pos[i];
        info->first_pos[i] = s.first_pos[i];
        info->phasing[i] = s.phasing[i];
    }
    return 0;
}

int wfx_debug_counters(wfx_ctx *ctx, long long out[8])
{
    CHECK_CTX(ctx);
    if (!out) return wfx_fail(ctx, WFX_ERR_BAD_ARG, "null buffer");
    WFX_HIP(ctx, hipStreamSynchronize(ctx->stream));
    for (int i = 0; i < 8; ++i) out[i] = ctx->h_scal->dbg[i];
    return 0;
}

static int buffer_of(wfx_ctx *ctx, int id, void **p, size_t *bytes)
{
    if (!ctx->ran) return wfx_fail(ctx, WFX_ERR_STATE, "no decode has run on this context");
    WFX_HIP(ctx, hipStreamSynchronize(ctx->stream));
    const uint64_t n = ctx->dp.n;
    switch (id) {
    case WFX_BUF_AUDIO: *p = ctx->b_audio.p; *bytes = n * 8; return 0;
    case WFX_BUF_ENVELOPE: *p = ctx->b_env.p; *bytes = n * 8; return 0;
    case WFX_BUF_DIGITAL: *p = ctx->b_dig.p; *bytes = n; return 0;
    case WFX_BUF_IMAGE:
        *p = ctx->img_in_ext ? (void *)((char *)ctx->ext_img + 16) : ctx->b_img.p;
        *bytes = (size_t)ctx->dp.width * 4 * (size_t)ctx->h_scal->height;
        return 0;
    default: return wfx_fail(ctx, WFX_ERR_BAD_ARG, "unknown buffer id %d", id);
    }
}

int wfx_decode_fetch(wfx_ctx *ctx, int buffer_id, void *host_out, size_t bytes)
{
    CHECK_CTX(ctx);
    void *p = nullptr;
    size_t nb = 0;
    WFX_TRY(buffer_of(ctx, buffer_id, &p, &nb));
    if (bytes != nb) return wfx_fail(ctx, WFX_ERR_BAD_ARG, "fetch: expected %zu bytes, got %zu", nb, bytes);
    if (nb && !host_out) return wfx_fail(ctx, WFX_ERR_BAD_ARG, "null buffer");
    return d2h_sync(ctx, host_out, p, nb);
}

int wfx_decode_device_ptr(wfx_ctx *ctx, int buffer_id, void **dev_ptr, size_t *bytes)
{
    CHECK_CTX(ctx);
    if (!dev_ptr || !bytes) return wfx_fail(ctx, WFX_ERR_BAD_ARG, "null argument");
    return buffer_of(ctx, buffer_id, dev_ptr, bytes);
}

// ---- device-resident stage calls (sample-range sharding) ------------------------------
int wfx_dev_malloc(wfx_ctx *ctx, size_t bytes, void **dev_ptr)
{
    CHECK_CTX(ctx);
    if (!dev_ptr) return wfx_fail(ctx, WFX_ERR_BAD_ARG, "null argument");
    *dev_ptr = nullptr;
    hipError_t e = hipMalloc(dev_ptr, bytes < 256 ? 256 : bytes + 256);     // padded: kernels may read a few bytes past a byte stream
    if (e != hipSuccess) return wfx_fail(ctx, WFX_ERR_OOM, "hipMalloc(%zu bytes) failed: %s", bytes, hipGetErrorString(e));
    return 0;
}

int wfx_dev_free(wfx_ctx *ctx, void *dev_ptr)
{
    CHECK_CTX(ctx);
    if (!dev_ptr) return 0;
    WFX_HIP(ctx, hipStreamSynchronize(ctx->stream));
    WFX_HIP(ctx, hipFree(dev_ptr));
    return 0;
}

int wfx_dev_upload(wfx_ctx *ctx, void *dst_dev, const void *src_host, size_t bytes)
{
    CHECK_CTX(ctx);
    if (bytes && (!dst_dev || !src_host)) return wfx_fail(ctx, WFX_ERR_BAD_ARG, "null buffer");
    WFX_TRY(h2d(ctx, dst_dev, src_host, bytes));
    WFX_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return 0;
}

int wfx_dev_download(wfx_ctx *ctx, void *dst_host, const void *src_dev, size_t bytes)
{
    CHECK_CTX(ctx);
    if (bytes && (!dst_host || !src_dev)) return wfx_fail(ctx, WFX_ERR_BAD_ARG, "null buffer");
    return d2h_sync(ctx, dst_host, src_dev, bytes);
}

int wfx_dev_copy(wfx_ctx *ctx, void *dst_dev, const void *src_dev, size_t bytes)
{
    CHECK_CTX(ctx);
    if (bytes && (!dst_dev || !src_dev)) return wfx_fail(ctx, WFX_ERR_BAD_ARG, "null buffer");
    if (bytes) WFX_HIP(ctx, hipMemcpyAsync(dst_dev, src_dev, bytes, hipMemcpyDeviceToDevice, ctx->stream));
    WFX_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return 0;
}

int wfx_d_notch_fir(wfx_ctx *ctx, const int16_t *in_dev, size_t n, const double b[3], const double a[3], double *out_dev, int edge_flags)
{
    CHECK_CTX(ctx);
    if (!in_dev || !out_dev || !b || !a) return wfx_fail(ctx, WFX_ERR_BAD_ARG, "null buffer");
    return wfx_dev_notch_fir_only(ctx, in_dev, WFX_IN_I16_MONO, n, b, a, out_dev, edge_flags);
}

int wfx_d_notch_fir_f64(wfx_ctx *ctx, const double *in_dev, size_t n, const double b[3], const double a[3], double *out_dev, int edge_flags)
{
    CHECK_CTX(ctx);
    if (!in_dev || !out_dev || !b || !a) return wfx_fail(ctx, WFX_ERR_BAD_ARG, "null buffer");
    return wfx_dev_notch_fir_only(ctx, in_dev, WFX_IN_F64_MONO, n, b, a, out_dev, edge_flags);
}

int wfx_d_decimate_fir64(wfx_ctx *ctx, const void *in_dev, int in_kind, size_t n_in, int64_t first, int factor, const double *coef,
                         int ntaps, double *out_dev, size_t n_out, int fix_shift, int *exact)
{
    CHECK_CTX(ctx);
    if (!in_dev || !out_dev || !coef) return wfx_fail(ctx, WFX_ERR_BAD_ARG, "null buffer");
    return wfx_dev_decimate_fir64(ctx, in_dev, in_kind, n_in, first, factor, coef, ntaps, out_dev, n_out, fix_shift, exact);
}

int wfx_d_decimate_fir64_batch(wfx_ctx *ctx, const void *in_dev, int in_kind, size_t n_in, int64_t first, int factor, const double *coef,
                               int ntaps, double *out_dev, size_t n_out, int fix_shift, int *exact, int nbatch, size_t in_stride, size_t out_stride)
{
    CHECK_CTX(ctx);
    if (!in_dev || !out_dev || !coef) return wfx_fail(ctx, WFX_ERR_BAD_ARG, "null buffer");
    return wfx_dev_decimate_fir64(ctx, in_dev, in_kind, n_in, first, factor, coef, ntaps, out_dev, n_out, fix_shift, exact, nbatch, in_stride, out_stride);
}

int wfx_d_ingest_chain(wfx_ctx *ctx, const void *in_dev, int in_kind, size_t n_in, int factor, const double *coef1, int ntaps1, int fix_shift,
                       int factor2, const double *coef2, int ntaps2, double *out_dev, size_t n_out, int nbatch, size_t in_stride, size_t out_stride,
                       int *handled)
{
    CHECK_CTX(ctx);
    if (!in_dev || !out_dev || !coef1 || !handled || (factor2 && !coef2)) return wfx_fail(ctx, WFX_ERR_BAD_ARG, "null buffer");
    return wfx_dev_ingest_stream(ctx, in_dev, in_kind, n_in, factor, coef1, ntaps1, fix_shift, factor2, coef2, ntaps2, out_dev, n_out, nbatch, in_stride,
                                 out_stride, handled);
}

int wfx_d_hilbert_fmm(wfx_ctx *ctx, const double *x_dev, size_t n, double *out_dev, int out_env, int *handled)
{
    CHECK_CTX(ctx);
    if (!x_dev || !out_dev || !handled) return wfx_fail(ctx, WFX_ERR_BAD_ARG, "null buffer");
    return wfx_dev_hilbert_fmm(ctx, x_dev, n, out_dev, out_env, nullptr, handled);
}

int wfx_plan_resample_direct(uint64_t n0, uint64_t num)
{
    return wfx_mr_resample_supported(n0, num) ? 1 : 0;
}

int wfx_d_resample_fmm(wfx_ctx *ctx, const double *x_dev, size_t n0, size_t num, double *y_dev, int *handled)
{
    CHECK_CTX(ctx);
    if (!x_dev || !y_dev || !handled) return wfx_fail(ctx, WFX_ERR_BAD_ARG, "null buffer");
    return wfx_dev_resample_fmm(ctx, x_dev, n0, num, y_dev, handled);
}

int wfx_d_read_rate(wfx_ctx *ctx, const void *dev, size_t bytes, int reps, double *gbs)
{
    CHECK_CTX(ctx);
    if (!dev || !gbs) return wfx_fail(ctx, WFX_ERR_BAD_ARG, "null buffer");
    return wfx_dev_read_rate(ctx, dev, bytes, reps, gbs);
}

int wfx_d_stream_rate(wfx_ctx *ctx, const void *dev, size_t bytes, double *out_dev, int reps, double *gbs)
{
    CHECK_CTX(ctx);
    if (!dev || !gbs) return wfx_fail(ctx, WFX_ERR_BAD_ARG, "null buffer");
    return wfx_dev_stream_rate(ctx, dev, bytes, out_dev, reps, gbs);
}

int wfx_d_median5(wfx_ctx *ctx, const double *in_dev, size_t n, double *out_dev)
{
    CHECK_CTX(ctx);
    if (!in_dev || !out_dev) return wfx_fail(ctx, WFX_ERR_BAD_ARG, "null buffer");
    return wfx_dev_median5(ctx, in_dev, n, out_dev, nullptr);
}

int wfx_d_select_hist(wfx_ctx *ctx, const double *env_dev, size_t n, int level, const uint64_t prefix[4], uint32_t *hist_dev)
{
    CHECK_CTX(ctx);
    if (!env_dev || !prefix || !hist_dev) return wfx_fail(ctx, WFX_ERR_BAD_ARG, "null buffer");
    return wfx_dev_select_level(ctx, env_dev, n, level, prefix, hist_dev);
}

int wfx_d_quantise(wfx_ctx *ctx, const double *env_dev, size_t n, double low, double high, uint8_t *out_dev, uint64_t *nan_count)
{
    CHECK_CTX(ctx);
    if (!env_dev || !out_dev) return wfx_fail(ctx, WFX_ERR_BAD_ARG, "null buffer");
    if (nan_count) *nan_count = 0;
    if (n == 0) return 0;
    ctx->ran = false;             // the context's scalar block is reused: a previous fused decode is no longer fetchable
    WFX_TRY(ensure_scal(ctx));
    memset(ctx->h_scal, 0, sizeof(wfx_dev_scalars));
    ctx->h_scal->low = low;
    ctx->h_scal->high = high;
    wfx_dev_scalars *ds = (wfx_dev_scalars *)ctx->b_scal.p;
    WFX_TRY(h2d(ctx, ds, ctx->h_scal, sizeof(wfx_dev_scalars)));
    WFX_TRY(wfx_dev_quantise(ctx, env_dev, n, ds, out_dev, ds));
    WFX_TRY(d2h_sync(ctx, ctx->h_scal, ds, sizeof(wfx_dev_scalars)));
    if (nan_count) *nan_count = ctx->h_scal->nan_count;
    return 0;
}

static void fill_info(wfx_decode_info *info, const wfx_dev_scalars &s, uint64_t n, int width)
{
    memset(info, 0, sizeof *info);
    info->n = n;
    info->low = s.low;
    info->high = s.high;
    info->nan_count = s.nan_count;
    info->npeaks = s.npeaks;
    info->hit_limit = s.hit_limit;
    info->no_group = s.no_group;
    info->n_phasing = s.n_phasing;
    info->start_frame = s.start_frame;
    info->width = width;
    info->height = s.height;
    for (int i = 0; i <= WFX_MAX_PEAKS; ++i) {
        info->peak_pos[i] = s.peak_pos[i];
        info->first_pos[i] = s.first_pos[i];
        info->phasing[i] = s.phasing[i];
    }
}

int wfx_d_sync_search(wfx_ctx *ctx, const uint8_t *d_dev, size_t n, size_t n_total, int n1, int n0_gap, int64_t mindistance, double frame_samples,
                      int width, wfx_decode_info *info)
{
    CHECK_CTX(ctx);
    if (!d_dev || !info) return wfx_fail(ctx, WFX_ERR_BAD_ARG, "null argument");
    if (width <= 0 || n_total < n) return wfx_fail(ctx, WFX_ERR_BAD_ARG, "bad width or total length");
    ctx->ran = false;
    WFX_TRY(ensure_scal(ctx));
    wfx_dev_scalars *ds = (wfx_dev_scalars *)ctx->b_scal.p;
    WFX_HIP(ctx, hipMemsetAsync(ds, 0, sizeof(wfx_dev_scalars), ctx->stream));
    WFX_TRY(wfx_dev_sync_pick(ctx, d_dev, n, n1, n0_gap, mindistance, frame_samples, width, ds));
    WFX_TRY(d2h_sync(ctx, ctx->h_scal, ds, sizeof(wfx_dev_scalars)));
    fill_info(info, *ctx->h_scal, n_total, width);
    // the picker saw only the first n samples: the line count refers to the whole capture
    info->height = ctx->h_scal->no_group ? 0 : (int)((n_total - (uint64_t)ctx->h_scal->start_frame) / (uint64_t)width);
    return 0;
}

int wfx_d_image_rows(wfx_ctx *ctx, const uint8_t *d_dev, size_t n, uint64_t g0, uint64_t start, int width, int h_total, int y0, int rows,
                     uint8_t *img_dev)
{
    CHECK_CTX(ctx);
    if (!d_dev || !img_dev) return wfx_fail(ctx, WFX_ERR_BAD_ARG, "null buffer");
    if (rows <= 0) return 0;
    // the rows draw on source lines max(y0-2,0) .. min(y0+rows+1, h_total-1): they must lie inside the slice
    const long long ylo = y0 - 2 < 0 ? 0 : y0 - 2, yhi = y0 + rows + 1 > h_total - 1 ? h_total - 1 : y0 + rows + 1;
    const unsigned long long first = start + (unsigned long long)ylo * width, last = start + (unsigned long long)(yhi + 1) * width;
    if (first < g0 || last > g0 + n) return wfx_fail(ctx, WFX_ERR_BAD_ARG, "image rows need samples [%llu, %llu) outside the slice [%llu, %llu)",
                                                     first, last, (unsigned long long)g0, (unsigned long long)(g0 + n));
    return wfx_dev_image_rows(ctx, d_dev, g0, start, width, h_total, y0, rows, img_dev);
}

int wfx_decode_copy_to_device(wfx_ctx *ctx, int buffer_id, void *dst_dev, size_t capacity, size_t *copied)
{
    CHECK_CTX(ctx);
    void *p = nullptr;
    size_t nb = 0;
    WFX_TRY(buffer_of(ctx, buffer_id, &p, &nb));
    if (nb > capacity) nb = capacity;
    if (nb && !dst_dev) return wfx_fail(ctx, WFX_ERR_BAD_ARG, "null destination");
    if (nb) WFX_HIP(ctx, hipMemcpyAsync(dst_dev, p, nb, hipMemcpyDeviceToDevice, ctx->stream));
    WFX_HIP(ctx, hipStreamSynchronize(ctx->stream));
    if (copied) *copied = nb;
    return 0;
}

// ---- live path, one audio packet (data_packet.py:408-464) ---------------------------------
int wfx_packet_process(wfx_ctx *ctx, const void *samples, int in_kind, size_t n, const double b[3], const double a[3],
                       const uint64_t ranks[4], double gamma_lo, double gamma_hi, uint8_t *out, double *low, double *high)
{
    CHECK_CTX(ctx);
    if (!samples || !b || !a || !ranks || !out) return wfx_fail(ctx, WFX_ERR_BAD_ARG, "null buffer");
    if (in_kind != WFX_IN_I16_MONO && in_kind != WFX_IN_F64_MONO) return wfx_fail(ctx, WFX_ERR_BAD_ARG, "packet: int16 or float64 samples");
    if (n == 0) return wfx_fail(ctx, WFX_ERR_BAD_ARG, "N must be positive.");
    ctx->ran = false;
    const size_t esz = in_kind == WFX_IN_I16_MONO ? 2 : 8;
    WFX_TRY(ensure_scal(ctx));
    WFX_TRY(wfx_reserve(ctx, ctx->b_x, n * esz + 16));
    WFX_TRY(wfx_reserve(ctx, ctx->b_audio, n * 8 + 16));
    WFX_TRY(wfx_reserve(ctx, ctx->b_envraw, n * 8 + 64));
    WFX_TRY(wfx_reserve(ctx, ctx->b_env, n * 8 + 64));
    WFX_TRY(wfx_reserve(ctx, ctx->b_dig, n + 64));
    WFX_TRY(h2d(ctx, ctx->b_x.p, samples, n * esz));
    wfx_dev_scalars *ds = (wfx_dev_scalars *)ctx->b_scal.p;
    WFX_HIP(ctx, hipMemsetAsync(ds, 0, sizeof(wfx_dev_scalars), ctx->stream));
    WFX_TRY(wfx_dev_notch(ctx, ctx->b_x.p, in_kind, n, b, a, (double *)ctx->b_audio.p));                 // :420-434
    ctx->force_pow2 = false;
    WFX_TRY(wfx_dev_hilbert_env_fft(ctx, (const double *)ctx->b_audio.p, n, (double *)ctx->b_envraw.p)); // :445
    WFX_TRY(wfx_dev_median3(ctx, (const double *)ctx->b_envraw.p, n, (double *)ctx->b_env.p));           // :446
    WFX_TRY(wfx_dev_percentiles(ctx, (const double *)ctx->b_env.p, n, ranks, gamma_lo, gamma_hi, ds));   // :457
    WFX_TRY(wfx_dev_quantise(ctx, (const double *)ctx->b_env.p, n, ds, (uint8_t *)ctx->b_dig.p, ds, 0.000001));   // :458-463
    WFX_TRY(d2h_sync(ctx, out, ctx->b_dig.p, n));
    WFX_TRY(d2h_sync(ctx, ctx->h_scal, ds, sizeof(wfx_dev_scalars)));
    if (low) *low = ctx->h_scal->low;
    if (high) *high = ctx->h_scal->high;
    return 0;
}

// `count` packets of n samples each (contiguous), decoded back to back on the context's stream: one upload, no host
// synchronisation between packets, one download (re-decoding a recorded packet stream; the live path makes one a second)
int wfx_packets_process(wfx_ctx *ctx, const void *samples, int in_kind, size_t n, size_t count, const double b[3], const double a[3],
                        const uint64_t ranks[4], double gamma_lo, double gamma_hi, uint8_t *out, double *low, double *high)
{
    CHECK_CTX(ctx);
    if (!samples || !b || !a || !ranks || !out) return wfx_fail(ctx, WFX_ERR_BAD_ARG, "null buffer");
    if (in_kind != WFX_IN_I16_MONO && in_kind != WFX_IN_F64_MONO) return wfx_fail(ctx, WFX_ERR_BAD_ARG, "packet: int16 or float64 samples");
    if (n == 0 || count == 0) return wfx_fail(ctx, WFX_ERR_BAD_ARG, "N must be positive.");
    ctx->ran = false;
    const size_t esz = in_kind == WFX_IN_I16_MONO ? 2 : 8;
    const size_t pitch = (n + 63) / 64 * 64 + 64;                    // bytes between two packets' outputs on the device
    WFX_TRY(wfx_reserve(ctx, ctx->b_scal, count * sizeof(wfx_dev_scalars)));
    WFX_TRY(wfx_reserve(ctx, ctx->b_x, count * n * esz + 16));
    WFX_TRY(wfx_reserve(ctx, ctx->b_audio, n * 8 + 16));
    WFX_TRY(wfx_reserve(ctx, ctx->b_envraw, n * 8 + 64));
    WFX_TRY(wfx_reserve(ctx, ctx->b_env, n * 8 + 64));
    WFX_TRY(wfx_reserve(ctx, ctx->b_dig, count * pitch));
    WFX_TRY(h2d(ctx, ctx->b_x.p, samples, count * n * esz));
    wfx_dev_scalars *ds = (wfx_dev_scalars *)ctx->b_scal.p;
    WFX_HIP(ctx, hipMemsetAsync(ds, 0, count * sizeof(wfx_dev_scalars), ctx->stream));
    ctx->force_pow2 = false;
    for (size_t p = 0; p < count; ++p) {
        const void *xin = (const char *)ctx->b_x.p + p * n * esz;
        uint8_t *dig = (uint8_t *)ctx->b_dig.p + p * pitch;
        WFX_TRY(wfx_dev_notch(ctx, xin, in_kind, n, b, a, (double *)ctx->b_audio.p));
        WFX_TRY(wfx_dev_hilbert_env_fft(ctx, (const double *)ctx->b_audio.p, n, (double *)ctx->b_envraw.p));
        WFX_TRY(wfx_dev_median3(ctx, (const double *)ctx->b_envraw.p, n, (double *)ctx->b_env.p));
        WFX_TRY(wfx_dev_percentiles(ctx, (const double *)ctx->b_env.p, n, ranks, gamma_lo, gamma_hi, ds + p));
        WFX_TRY(wfx_dev_quantise(ctx, (const double *)ctx->b_env.p, n, ds + p, dig, ds + p, 0.000001));
    }
    WFX_HIP(ctx, hipMemcpy2DAsync(out, n, ctx->b_dig.p, pitch, n, count, hipMemcpyDeviceToHost, ctx->stream));
    std::vector<wfx_dev_scalars> hs(count);
    WFX_TRY(d2h_sync(ctx, hs.data(), ds, count * sizeof(wfx_dev_scalars)));
    for (size_t p = 0; p < count; ++p) {
        if (low) low[p] = hs[p].low;
        if (high) high[p] = hs[p].high;
    }
    return 0;
}

// ---- live path, detectors: one-sided amplitude spectrum of a packet (data_packet.py:388-406) ----
int wfx_packet_spectrum(wfx_ctx *ctx, const void *samples, int in_kind, size_t n, double *amp_out)
{
    CHECK_CTX(ctx);
    if (!samples || !amp_out) return wfx_fail(ctx, WFX_ERR_BAD_ARG, "null buffer");
    if (in_kind != WFX_IN_I16_MONO && in_kind != WFX_IN_F64_MONO) return wfx_fail(ctx, WFX_ERR_BAD_ARG, "packet: int16 or float64 samples");
    if (n < 2) return wfx_fail(ctx, WFX_ERR_BAD_ARG, "spectrum: at least two samples needed");
    ctx->ran = false;
    const size_t esz = in_kind == WFX_IN_I16_MONO ? 2 : 8;
    WFX_TRY(wfx_reserve(ctx, ctx->b_x, n * esz + 16));
    WFX_TRY(wfx_reserve(ctx, ctx->b_audio, n * 8 + 16));
    WFX_TRY(wfx_reserve(ctx, ctx->b_env, (n / 2) * 8 + 64));
    WFX_TRY(h2d(ctx, ctx->b_x.p, samples, n * esz));
    const double *x = (const double *)ctx->b_x.p;
    if (in_kind == WFX_IN_I16_MONO) {
        WFX_TRY(wfx_dev_i16_to_f64(ctx, (const int16_t *)ctx->b_x.p, n, (double *)ctx->b_audio.p));
        x = (const double *)ctx->b_audio.p;
    }
    WFX_TRY(wfx_dev_spectrum_abs(ctx, x, n, (double *)ctx->b_env.p));
    return d2h_sync(ctx, amp_out, ctx->b_env.p, (n / 2) * 8);
}

// ---- asynchronous export for a collective ------------------------------------------------
int wfx_stream_handle(wfx_ctx *ctx, void **stream)
{
    CHECK_CTX(ctx);
    if (!stream) return wfx_fail(ctx, WFX_ERR_BAD_ARG, "null argument");
    *stream = (void *)ctx->stream;
    return 0;
}

int wfx_decode_bind_image(wfx_ctx *ctx, void *dst_dev, size_t capacity)
{
    CHECK_CTX(ctx);
    if (dst_dev && capacity < 16) return wfx_fail(ctx, WFX_ERR_BAD_ARG, "bind_image: destination of at least 16 bytes needed");
    ctx->ext_img = dst_dev;
    ctx->ext_img_cap = dst_dev ? capacity : 0;
    return 0;
}

int wfx_decode_export_async(wfx_ctx *ctx, int buffer_id, void *dst_dev, size_t capacity)
{
    CHECK_CTX(ctx);
    if (!ctx->ran) return wfx_fail(ctx, WFX_ERR_STATE, "no decode has been enqueued on this context");
    if (!dst_dev || capacity < 16) return wfx_fail(ctx, WFX_ERR_BAD_ARG, "export: destination of at least 16 bytes needed");
    const uint64_t n = ctx->dp.n;
    const void *src = nullptr;
    long long fixed = -1;            // byte count known on the host, or -1: 4 * height * width with the height still on the device
    size_t maxb = 0;
    switch (buffer_id) {
    case WFX_BUF_AUDIO: src = ctx->b_audio.p; fixed = (long long)(n * 8); maxb = n * 8; break;
    case WFX_BUF_ENVELOPE: src = ctx->b_env.p; fixed = (long long)(n * 8); maxb = n * 8; break;
    case WFX_BUF_DIGITAL: src = ctx->b_dig.p; fixed = (long long)n; maxb = n; break;
    case WFX_BUF_IMAGE:
        if (ctx->img_in_ext && dst_dev == ctx->ext_img) return 0;       // the decode wrote header and image there itself
        src = ctx->img_in_ext ? (const void *)((const char *)ctx->ext_img + 16) : ctx->b_img.p;
        maxb = (size_t)ctx->dp.width * 4 * (size_t)(n / (uint64_t)ctx->dp.width);
        break;
    default: return wfx_fail(ctx, WFX_ERR_BAD_ARG, "unknown buffer id %d", buffer_id);
    }
    const size_t room = capacity - 16;
    const size_t nb = maxb < room ? maxb : room;
    WFX_TRY(wfx_dev_export_header(ctx, (const wfx_dev_scalars *)ctx->b_scal.p, fixed, ctx->dp.width, (long long)room, (long long *)dst_dev));
    if (nb) WFX_HIP(ctx, hipMemcpyAsync((char *)dst_dev + 16, src, nb, hipMemcpyDeviceToDevice, ctx->stream));
    return 0;
}

}  // extern "C"
