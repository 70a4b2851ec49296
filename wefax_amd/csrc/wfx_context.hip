// Context, device-buffer pool, error reporting, HIP-event stopwatch and the
// per-kernel HIP-event profile of libwefax_hip.so.
#include <cstdarg>
#include <cstring>

#include "wfx_internal.h"

#include <mutex>
#include <set>

static thread_local std::string g_err;

// live contexts: objects that outlive their context (a shard destroyed late by a garbage collector) must not touch it
static std::set<const wfx_ctx *> g_live;
static std::mutex g_live_mutex;
bool wfx_ctx_alive(const wfx_ctx *ctx)
{
    std::lock_guard<std::mutex> lock(g_live_mutex);
    return g_live.count(ctx) != 0;
}

void wfx_set_global_error(const char *msg) { g_err = msg ? msg : ""; }

int wfx_fail(wfx_ctx *ctx, int code, const char *fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (ctx) ctx->err = buf;
    g_err = buf;
    return code;
}

int wfx_fail_hip(wfx_ctx *ctx, hipError_t e, const char *what)
{
    return wfx_fail(ctx, e == hipErrorOutOfMemory ? WFX_ERR_OOM : WFX_ERR_HIP, "HIP error %d (%s) in %s",
                    (int)e, hipGetErrorString(e), what);
}

int wfx_reserve(wfx_ctx *ctx, wfx_devbuf &b, size_t bytes)
{
    if (bytes <= b.cap && b.p) return 0;
    if (b.p) {
        hipStreamSynchronize(ctx->stream);
        hipFree(b.p);
        b.p = nullptr;
        b.cap = 0;
    }
    size_t want = bytes < 256 ? 256 : bytes;
    hipError_t e = hipMalloc(&b.p, want);
    if (e != hipSuccess) {
        b.p = nullptr;
        return wfx_fail(ctx, WFX_ERR_OOM, "hipMalloc(%zu bytes) failed: %s", want, hipGetErrorString(e));
    }
    b.cap = want;
    return 0;
}

static void free_buf(wfx_devbuf &b)
{
    if (b.p) hipFree(b.p);
    b.p = nullptr;
    b.cap = 0;
}

// Device copy of a host table of floats, cached by content (FNV-1a over the bytes + an exact compare is not needed: a
// colliding table of the same size would have to come from the same filter design code).
const float *wfx_coef_device(wfx_ctx *ctx, const float *host, size_t count)
{
    const size_t bytes = count * sizeof(float);
    uint64_t h = 1469598103934665603ull;
    const unsigned char *b = (const unsigned char *)host;
    for (size_t i = 0; i < bytes; ++i) h = (h ^ b[i]) * 1099511628211ull;
    for (auto &e : ctx->coef_cache)
        if (e.hash == h && e.bytes == bytes) return (const float *)e.dev;
    if (ctx->coef_cache.size() >= 64) {           // a long-lived context fed ever new filters: start over
        if (hipStreamSynchronize(ctx->stream) != hipSuccess) return nullptr;
        // the tables of this generation may be in a caller's hands (a launcher fetches two tables before it launches: ADVICE r5):
        // they are retired, and what was retired one eviction ago is freed
        for (auto &e : ctx->coef_retired) (void)hipFree(e.dev);
        ctx->coef_retired.swap(ctx->coef_cache);
        ctx->coef_cache.clear();
    }
    void *d = nullptr;
    hipError_t e = hipMalloc(&d, bytes ? bytes : 4);
    if (e == hipSuccess) e = hipMemcpyAsync(d, host, bytes, hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);       // the host table belongs to the caller
    if (e != hipSuccess) {
        if (d) (void)hipFree(d);
        wfx_fail_hip(ctx, e, "filter table upload");
        return nullptr;
    }
    ctx->coef_cache.push_back({h, bytes, d});
    return (const float *)d;
}

extern "C" {

const char *wfx_version(void) { return "wefax_hip 0.1 (gfx950)"; }

int wfx_device_pci_bus_id(wfx_ctx *ctx, char *out, int cap)
{
    if (!ctx || !out || cap < 16) return wfx_fail(ctx, WFX_ERR_BAD_ARG, "pci bus id: buffer of at least 16 bytes");
    WFX_HIP(ctx, hipDeviceGetPCIBusId(out, cap, ctx->device));
    for (char *p = out; *p; ++p)
        if (*p >= 'A' && *p <= 'F') *p = (char)(*p - 'A' + 'a');
    return 0;
}

int wfx_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

wfx_ctx *wfx_create(int device, int flags)
{
    (void)flags;
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0) {
        wfx_fail(nullptr, WFX_ERR_HIP, "no HIP device available (%s)",
                 e == hipSuccess ? "device count 0" : hipGetErrorString(e));
        return nullptr;
    }
    if (device < 0 || device >= n) {
        wfx_fail(nullptr, WFX_ERR_BAD_ARG, "device %d out of range [0, %d)", device, n);
        return nullptr;
    }
    if ((e = hipSetDevice(device)) != hipSuccess) {
        wfx_fail_hip(nullptr, e, "hipSetDevice");
        return nullptr;
    }
    wfx_ctx *ctx = new wfx_ctx();
    ctx->device = device;
    if ((e = hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking)) != hipSuccess) {
        wfx_fail_hip(nullptr, e, "hipStreamCreate");
        delete ctx;
        return nullptr;
    }
    hipEventCreate(&ctx->t0);
    hipEventCreate(&ctx->t1);
    if (hipHostMalloc((void **)&ctx->h_info, sizeof(wfx_decode_info), hipHostMallocDefault) != hipSuccess ||
        hipHostMalloc((void **)&ctx->h_scal, sizeof(wfx_dev_scalars), hipHostMallocDefault) != hipSuccess) {
        wfx_fail(nullptr, WFX_ERR_OOM, "pinned host allocation failed");
        wfx_destroy(ctx);
        return nullptr;
    }
    memset(ctx->h_info, 0, sizeof(wfx_decode_info));
    memset(ctx->h_scal, 0, sizeof(wfx_dev_scalars));
    {
        std::lock_guard<std::mutex> lock(g_live_mutex);
        g_live.insert(ctx);
    }
    return ctx;
}

void wfx_destroy(wfx_ctx *ctx)
{
    if (!ctx) return;
    {
        std::lock_guard<std::mutex> lock(g_live_mutex);
        g_live.erase(ctx);
    }
    hipSetDevice(ctx->device);
    if (ctx->stream) hipStreamSynchronize(ctx->stream);
    wfx_devbuf *bufs[] = {&ctx->b_in, &ctx->b_x, &ctx->b_audio, &ctx->b_work, &ctx->b_work2, &ctx->b_envraw,
                          &ctx->b_env, &ctx->b_dig, &ctx->b_corr, &ctx->b_img, &ctx->b_hist, &ctx->b_tmp,
                          &ctx->b_tmp2, &ctx->b_w256, &ctx->b_scal, &ctx->b_taps, &ctx->b_cand, &ctx->b_pcoef, &ctx->b_seg, &ctx->b_png, &ctx->b_synth};
    if (ctx->h_png) hipHostFree(ctx->h_png);
    for (auto *b : bufs) free_buf(*b);
    for (auto &e : ctx->coef_cache) (void)hipFree(e.dev);
    for (auto &e : ctx->coef_retired) (void)hipFree(e.dev);
    for (auto &e : ctx->fmm_tables) (void)hipFree((void *)e.second);
    for (auto &kv : ctx->plans) free_buf(kv.second.bhat);
    for (auto &kv : ctx->hplans) free_buf(kv.second.bhat);
    wfx_mr_release(ctx);
    for (auto &r : ctx->prof_recs) {
        hipEventDestroy(r.a);
        hipEventDestroy(r.b);
    }
    for (auto ev : ctx->ev_pool) hipEventDestroy(ev);
    if (ctx->t0) hipEventDestroy(ctx->t0);
    if (ctx->t1) hipEventDestroy(ctx->t1);
    if (ctx->h_info) hipHostFree(ctx->h_info);
    if (ctx->h_scal) hipHostFree(ctx->h_scal);
    if (ctx->stream) hipStreamDestroy(ctx->stream);
    delete ctx;
}

const char *wfx_last_error(wfx_ctx *ctx) { return ctx ? ctx->err.c_str() : g_err.c_str(); }

int wfx_sync(wfx_ctx *ctx)
{
    if (!ctx) return wfx_fail(nullptr, WFX_ERR_BAD_ARG, "null context");
    WFX_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return 0;
}

// pinned (page-locked) host memory for captures and images that cross PCIe at DMA speed
void *wfx_host_alloc(size_t bytes)
{
    void *p = nullptr;
    if (hipHostMalloc(&p, bytes ? bytes : 1, hipHostMallocDefault) != hipSuccess) {
        wfx_fail(nullptr, WFX_ERR_OOM, "pinned host allocation of %zu bytes failed", bytes);
        return nullptr;
    }
    return p;
}

void wfx_host_free(void *p)
{
    if (p) (void)hipHostFree(p);
}

int wfx_timer_start(wfx_ctx *ctx)
{
    if (!ctx) return wfx_fail(nullptr, WFX_ERR_BAD_ARG, "null context");
    WFX_HIP(ctx, hipEventRecord(ctx->t0, ctx->stream));
    return 0;
}

int wfx_timer_stop(wfx_ctx *ctx, float *ms)
{
    if (!ctx || !ms) return wfx_fail(ctx, WFX_ERR_BAD_ARG, "null argument");
    WFX_HIP(ctx, hipEventRecord(ctx->t1, ctx->stream));
    WFX_HIP(ctx, hipEventSynchronize(ctx->t1));
    WFX_HIP(ctx, hipEventElapsedTime(ms, ctx->t0, ctx->t1));
    return 0;
}

// ---- per-kernel profile -------------------------------------------------------
static const char *k_names[K_COUNT] = {
    "merge_channels", "notch_filtfilt", "bluestein_pointwise", "fft_pass_fwd", "fft_pass_inv",
    "env_median",     "fir_analytic",   "median5",             "select_hist",  "select_scan",
    "quantise",       "sync_corr",      "sync_pick",           "lines_to_image", "resample_pointwise",
    "polyphase_ingest", "polyphase_stages", "dist_copy",
    "fmm_notch_p2m_m2m", "fmm_tiers_and_top", "fmm_tree_levels", "fmm_near_l2p_env_median", "resample_fmm_p2m_m2m", "resample_fmm_near_l2p"};

int wfx_profile_kernel_count(void) { return K_COUNT; }
const char *wfx_profile_kernel_name(int i) { return (i >= 0 && i < K_COUNT) ? k_names[i] : ""; }

static int prof_collect(wfx_ctx *ctx)
{
    if (ctx->prof_recs.empty()) return 0;
    WFX_HIP(ctx, hipStreamSynchronize(ctx->stream));
    for (auto &r : ctx->prof_recs) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, r.a, r.b) == hipSuccess) {
            ctx->prof_count[r.kid] += 1;
            ctx->prof_ms[r.kid] += ms;
        }
        ctx->ev_pool.push_back(r.a);
        ctx->ev_pool.push_back(r.b);
    }
    ctx->prof_recs.clear();
    return 0;
}

int wfx_profile_enable(wfx_ctx *ctx, int on)
{
    if (!ctx) return wfx_fail(nullptr, WFX_ERR_BAD_ARG, "null context");
    if (!on) WFX_TRY(prof_collect(ctx));
    ctx->prof = on != 0;
    return 0;
}

int wfx_profile_reset(wfx_ctx *ctx)
{
    if (!ctx) return wfx_fail(nullptr, WFX_ERR_BAD_ARG, "null context");
    WFX_TRY(prof_collect(ctx));
    memset(ctx->prof_count, 0, sizeof ctx->prof_count);
    memset(ctx->prof_ms, 0, sizeof ctx->prof_ms);
    return 0;
}

int wfx_profile_get(wfx_ctx *ctx, int i, uint64_t *launches, double *total_ms)
{
    if (!ctx || i < 0 || i >= K_COUNT) return wfx_fail(ctx, WFX_ERR_BAD_ARG, "bad kernel index");
    WFX_TRY(prof_collect(ctx));
    if (launches) *launches = ctx->prof_count[i];
    if (total_ms) *total_ms = ctx->prof_ms[i];
    return 0;
}

}  // extern "C"

static hipEvent_t take_event(wfx_ctx *ctx)
{
    if (!ctx->ev_pool.empty()) {
        hipEvent_t e = ctx->ev_pool.back();
        ctx->ev_pool.pop_back();
        return e;
    }
    hipEvent_t e = nullptr;
    hipEventCreate(&e);
    return e;
}

void wfx_prof_begin(wfx_ctx *ctx, int kid)
{
    if (!ctx->prof) return;
    wfx_prof_rec r;
    r.a = take_event(ctx);
    r.b = take_event(ctx);
    r.kid = kid;
    hipEventRecord(r.a, ctx->stream);
    ctx->prof_recs.push_back(r);
}

void wfx_prof_events(wfx_ctx *ctx, int kid, hipEvent_t *a, hipEvent_t *b)
{
    wfx_prof_rec r;
    r.a = take_event(ctx);
    r.b = take_event(ctx);
    r.kid = kid;
    ctx->prof_recs.push_back(r);
    *a = r.a;
    *b = r.b;
}

void wfx_prof_end(wfx_ctx *ctx)
{
    if (!ctx->prof) return;
    hipEventRecord(ctx->prof_recs.back().b, ctx->stream);
}
