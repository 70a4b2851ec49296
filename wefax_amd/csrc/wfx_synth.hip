// Test-signal synthesis on the device (measurement support, not on the decode path): the WEFAX transmission of
// wefax_amd/synth.py -- start tone, phasing lines, ramp image lines, stop tone, black tail; phase-continuous FM
// sin / exp(j 2 pi cumsum(f) / fs) + white noise -> int16 -- for any range of frames of a capture that is too large to
// build on the host or to fit a RIFF file (BASELINE configs[3]: 60 minutes at 1.536 MS/s IQ = 5.53 G frames, 22 GB).
//
//   1. chunk sums of the instantaneous frequency (65 536 frames per chunk, every chunk up to the end of the range)
//   2. one thread scans them (sum reduced modulo fs after every step: the phase only needs it modulo fs)
//   3. per chunk of the requested range: thread-local running sums + a block scan -> phase -> cos / sin + noise -> int16
//
// The noise of a frame depends only on (seed, frame index, channel): overlapping ranges produced by different ranks agree
// bit for bit.  It is a counter-based generator (splitmix64 -> Box-Muller), not NumPy's: with noise = 0 the output equals
// synth.synth_capture to +-1 count on rare rounding ties (tests/test_sharded.py).
#include "wfx_internal.h"

#define SY_CHUNK 65536
#define SY_PER_THREAD 256

struct synth_track {
    double fs, t_line, start_hz, t_start_end, t_phasing_end, t_image_end, t_stop_end;
};

__device__ __forceinline__ double synth_freq(const synth_track &k, long long i)
{
    const double black = 1500.0, white = 2300.0;
    const double t = (double)i / k.fs;
    if (t < k.t_start_end) {
        const double ph = fmod(t * k.start_hz, 1.0);
        return ph < 0.5 ? white : black;
    }
    if (t < k.t_phasing_end) {
        const double frac = fmod((t - k.t_start_end) / k.t_line, 1.0);
        return frac < 0.05 ? white : black;
    }
    if (t < k.t_image_end) {
        const double frac = fmod((t - k.t_phasing_end) / k.t_line, 1.0);
        return frac < 0.05 ? white : black + (white - black) * (frac - 0.05) / 0.95;
    }
    if (t < k.t_stop_end) {
        const double ph = fmod((t - k.t_image_end) * 450.0, 1.0);
        return ph < 0.5 ? white : black;
    }
    return black;
}

__global__ void __launch_bounds__(256) synth_chunk_sums(synth_track k, long long n_total, int nchunks, double *__restrict__ sums)
{
    __shared__ double part[256];
    const int c = blockIdx.x;
    const long long base = (long long)c * SY_CHUNK + (long long)threadIdx.x * SY_PER_THREAD;
    double s = 0.0;
    for (int j = 0; j < SY_PER_THREAD; ++j) {
        const long long i = base + j;
        if (i < n_total) s += synth_freq(k, i);
    }
    part[threadIdx.x] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        double tot = 0.0;
        for (int j = 0; j < 256; ++j) tot += part[j];
        sums[c] = tot;
    }
}

// offsets[c] = (sum of the chunks before c) mod fs
__global__ void synth_scan_chunks(const double *__restrict__ sums, int nchunks, double fs, double *__restrict__ offsets)
{
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    double acc = 0.0;
    for (int c = 0; c < nchunks; ++c) {
        offsets[c] = acc;
        acc = fmod(acc + sums[c], fs);
    }
}

__device__ __forceinline__ unsigned long long splitmix64(unsigned long long x)
{
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}

__device__ __forceinline__ double synth_gauss(unsigned long long seed, long long i, int channel)
{
    const unsigned long long a = splitmix64(seed ^ splitmix64((unsigned long long)i * 2ull + (unsigned long long)channel));
    const unsigned long long b = splitmix64(a);
    const double u1 = ((double)(a >> 11) + 1.0) * (1.0 / 9007199254740993.0);      // (0, 1)
    const double u2 = (double)(b >> 11) * (1.0 / 9007199254740992.0);
    double sn, cs;
    sincospi(2.0 * u2, &sn, &cs);
    return sqrt(-2.0 * log(u1)) * cs;
}

__device__ __forceinline__ short synth_i16(double v)
{
    double r = rint(v * 32767.0);
    r = r < -32768.0 ? -32768.0 : (r > 32767.0 ? 32767.0 : r);
    return (short)r;
}

// frames [lo, hi) of the capture (all inside [0, n_total)), written to out[(i - lo)] (mono) or out[2 (i - lo) + {0, 1}] (IQ)
__global__ void __launch_bounds__(256) synth_emit(synth_track k, long long lo, long long hi, int chunk0, const double *__restrict__ offsets, double amplitude,
                                                   double noise, unsigned long long seed, int iq, short *__restrict__ out)
{
    __shared__ double tot[256];
    const int c = chunk0 + blockIdx.x, t = threadIdx.x;
    const long long base = (long long)c * SY_CHUNK + (long long)t * SY_PER_THREAD;
    // pass 1: the thread's total; block scan of the totals
    double s = 0.0;
    for (int j = 0; j < SY_PER_THREAD; ++j) s += synth_freq(k, base + j);
    tot[t] = s;
    __syncthreads();
    double before = offsets[c];
    for (int j = 0; j < t; ++j) before += tot[j];
    // pass 2: inclusive running sum (numpy.cumsum), phase, samples
    double acc = before;
    const double w = 2.0 / k.fs;                      // phase / pi
    for (int j = 0; j < SY_PER_THREAD; ++j) {
        const long long i = base + j;
        acc += synth_freq(k, i);
        if (i < lo || i >= hi) continue;
        double sn, cs;
        sincospi(fmod(acc, k.fs) * w, &sn, &cs);
        if (iq) {
            double xi = amplitude * cs, xq = amplitude * sn;
            if (noise > 0.0) {
                xi += noise * synth_gauss(seed, i, 0);
                xq += noise * synth_gauss(seed, i, 1);
            }
            ((short2 *)out)[i - lo] = make_short2(synth_i16(xi), synth_i16(xq));
        } else {
            double x = amplitude * sn;
            if (noise > 0.0) x += noise * synth_gauss(seed, i, 0);
            out[i - lo] = synth_i16(x);
        }
    }
}

static void make_track(const wfx_synth_params &p, synth_track &k)
{
    k.fs = p.sample_rate;
    k.t_line = 60.0 / (double)p.lines_per_minute;
    k.start_hz = p.ioc == 576 ? 300.0 : 675.0;
    k.t_start_end = p.start_tone_s;
    k.t_phasing_end = k.t_start_end + p.phasing_lines * k.t_line;
    k.t_image_end = k.t_phasing_end + p.image_lines * k.t_line;
    k.t_stop_end = k.t_image_end + p.stop_tone_s;
}

extern "C" {

uint64_t wfx_synth_frames(const wfx_synth_params *p)
{
    if (!p || !(p->sample_rate > 0) || p->lines_per_minute <= 0) return 0;
    const double dur = p->start_tone_s + (p->phasing_lines + p->image_lines) * (60.0 / (double)p->lines_per_minute) + p->stop_tone_s + p->black_tail_s;
    return (uint64_t)llround(dur * p->sample_rate);
}

int wfx_synth_capture(wfx_ctx *ctx, const wfx_synth_params *p, uint64_t lo, uint64_t hi, void *dev_out)
{
    if (!ctx || !p || !dev_out) return wfx_fail(ctx, WFX_ERR_BAD_ARG, "null argument");
    (void)hipSetDevice(ctx->device);
    const uint64_t n_total = wfx_synth_frames(p);
    if (n_total == 0 || lo >= hi || hi > n_total) return wfx_fail(ctx, WFX_ERR_BAD_ARG, "synth: frames [%llu, %llu) outside the capture of %llu", (unsigned long long)lo, (unsigned long long)hi, (unsigned long long)n_total);
    synth_track k;
    make_track(*p, k);
    const int c0 = (int)(lo / SY_CHUNK), c1 = (int)((hi + SY_CHUNK - 1) / SY_CHUNK);       // chunks of the range
    // the phase at the start of every chunk of the capture: computed once per recipe and kept with the context (a rank of the
    // columns layout asks for 225 slices of one capture; the serial scan over 84 000 chunks of the 60-minute stream takes ~10 ms)
    const int call = (int)((n_total + SY_CHUNK - 1) / SY_CHUNK);
    unsigned long long key = 1469598103934665603ull;
    for (size_t i = 0; i < sizeof(wfx_synth_params); ++i) key = (key ^ ((const unsigned char *)p)[i]) * 1099511628211ull;
    if (!ctx->b_synth.p || ctx->synth_key != key) {
        WFX_TRY(wfx_reserve(ctx, ctx->b_synth, (size_t)call * 2 * sizeof(double) + 64));
        double *sums_all = (double *)ctx->b_synth.p + call;
        WFX_LAUNCH(ctx, K_MERGE, synth_chunk_sums, dim3(call), dim3(256), k, (long long)n_total, call, sums_all);
        WFX_LAUNCH(ctx, K_MERGE, synth_scan_chunks, dim3(1), dim3(64), (const double *)sums_all, call, k.fs, (double *)ctx->b_synth.p);
        ctx->synth_key = key;
    }
    const double *offs = (const double *)ctx->b_synth.p;
    WFX_LAUNCH(ctx, K_MERGE, synth_emit, dim3(c1 - c0), dim3(256), k, (long long)lo, (long long)hi, c0, (const double *)offs, p->amplitude, p->noise,
               (unsigned long long)p->seed, p->iq, (short *)dev_out);
    return 0;
}

}  // extern "C"
