/*
 * wefax_hip.h -- C ABI of libwefax_hip.so, the MI355X (gfx950) implementation of
 * the WEFAX file-decoding hot path  wojlin/WEFAX wefax.py: Demodulator.process().
 *
 * The reference has no FFI (it is pure Python); each entry point below replaces
 * the body of one reference function and cites it.  Plain C types only: no C++
 * exceptions, no torch types.  Every function returns 0 on success or a negative
 * wfx_status; the message is available from wfx_last_error().
 *
 * Ownership: the caller owns every host buffer; the library never keeps a host
 * pointer after a call returns.  Device memory belongs to the context and is
 * reused across calls.  A context is not thread-safe; use one per Demodulator
 * (several contexts may run concurrently from different threads).
 *
 * Reference binding a maintainer would add: see INTEGRATION.md (ctypes stub).
 */
#ifndef WEFAX_HIP_H
#define WEFAX_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct wfx_ctx wfx_ctx;

typedef enum {
    WFX_OK = 0,
    WFX_ERR_BAD_ARG = -1,
    WFX_ERR_HIP = -2,
    WFX_ERR_OOM = -3,
    WFX_ERR_STATE = -4,
    WFX_ERR_COMM = -5,
    WFX_ERR_SHORT_FILE = -6     /* a wav whose data chunk is shorter than its header says: the binding raises what scipy.io.wavfile.read raises (ValueError) */
} wfx_status;

/* sample formats accepted at ingest (wefax.py:348-373) */
typedef enum {
    WFX_IN_I16_MONO = 0,    /* int16[n]                                         */
    WFX_IN_I16_STEREO = 1,  /* int16[n][2] -> (int16)(L+R) wrapped, then /2      */
    WFX_IN_F64_MONO = 2,    /* float64[n] (any other wav dtype, converted on host) */
    WFX_IN_F32_MONO = 3,    /* float32[n]: only between the stages of the time-domain front end (wfx_d_decimate_fir ...) */
    /* two-channel wavs in the other sample formats scipy.io.wavfile returns: wefax.py:372 adds the two numpy scalars IN THE FILE'S
     * dtype -- uint8 wraps modulo 2^8, int32 modulo 2^32, float32 rounds to float32 -- then divides by 2 (fused decode and
     * wfx_merge_channels_any only; the merged samples continue as float64) */
    WFX_IN_U8_STEREO = 4,   /* uint8[n][2]   */
    WFX_IN_I32_STEREO = 5,  /* int32[n][2]   */
    WFX_IN_F32_STEREO = 6   /* float32[n][2] */
} wfx_in_kind;

/* how the analytic signal (scipy.signal.hilbert, wefax.py:174) is computed */
typedef enum {
    WFX_HILBERT_FFT = 0,    /* exact: circular convolution with ifft(h) (mixed-radix N/2-point or zero-padded power-of-two transforms) */
    /* 1 was WFX_HILBERT_FIR (rounds 1-2): a truncated sliding-window FIR.  It cannot stay within one grey level of the
     * reference's global FFT on noisy captures and no product path used it: removed in round 3. */
    WFX_HILBERT_BLUESTEIN = 2, /* exact, literal fft -> h -> ifft via two Bluestein DFTs (cross-check) */
    WFX_HILBERT_FFT_POW2 = 3, /* exact, like WFX_HILBERT_FFT but always the zero-padded power-of-two form
                                 (WFX_HILBERT_FFT picks the unpadded mixed-radix form when N/2 is 13-smooth) */
    WFX_HILBERT_FMM = 4       /* no transform over the capture: near field summed directly + fast multipole far field on 16 Chebyshev
                                 nodes per box (csrc/wfx_fmm.hip; 1e-14 relative against the transform forms); even N >= 32768,
                                 other lengths run WFX_HILBERT_FFT.  A capture that is resampled on the way takes the resampler's
                                 multipole form as well where its lengths have one (down-sampling to an even count): the arithmetic of
                                 the sharded plan 3, whose bytes a decode in this mode reproduces on one GPU */
} wfx_hilbert_mode;

#define WFX_MAX_PEAKS 100   /* wefax.py:251 */

/* ---- lifecycle ------------------------------------------------------- */
int         wfx_device_count(void);
wfx_ctx    *wfx_create(int device, int flags);          /* NULL on failure      */
void        wfx_destroy(wfx_ctx *ctx);
const char *wfx_last_error(wfx_ctx *ctx);               /* ctx may be NULL      */
int         wfx_sync(wfx_ctx *ctx);                     /* wait for the stream  */
const char *wfx_version(void);
/* PCI address of the context's GPU as the driver's sysfs tree names it ("0000:05:00.0"): lets a measurement harness read the clocks and power of
 * THE GPU the context runs on, not of the first card of the machine; WFX_ERR_BAD_ARG when `cap` is too small */
int         wfx_device_pci_bus_id(wfx_ctx *ctx, char *out, int cap);

/* ---- stage entry points (host in, host out), one per reference stage --- */

/* a4  wefax.py:360-373 __merge_channels: out[i] = (double)(int16)(L+R) / 2 */
int wfx_merge_channels(wfx_ctx *ctx, const int16_t *lr, size_t n, double *out);
/* the same for any two-channel kind (WFX_IN_I16_STEREO, WFX_IN_U8_STEREO, WFX_IN_I32_STEREO, WFX_IN_F32_STEREO) */
int wfx_merge_channels_any(wfx_ctx *ctx, const void *lr, int in_kind, size_t n, double *out);

/* a5  wefax.py:375-394 __resample -> scipy.signal.resample(x, num) (real input) */
int wfx_resample(wfx_ctx *ctx, const double *x, size_t n0, size_t num, double *out);

/* a6  wefax.py:68-72 iirnotch + filtfilt.  b/a: biquad (a[0] == 1).  For
 *     WFX_IN_I16_MONO the 9-sample odd extension wraps in int16 as it does inside
 *     scipy.signal.filtfilt when handed an int16 array. */
int wfx_notch_filtfilt(wfx_ctx *ctx, const void *in, int in_kind, size_t n,
                       const double b[3], const double a[3], double *out);
/* the same with the 9 + 9 samples of the odd extension given by the caller (computed in the capture's own dtype) */
int wfx_notch_filtfilt_ext(wfx_ctx *ctx, const void *in, int in_kind, size_t n, const double b[3], const double a[3],
                           const double ext_left[9], const double ext_right[9], double *out);

/* a7  wefax.py:166-183 __demodulate: medfilt(abs(hilbert(x)), 5) */
int wfx_analytic_env(wfx_ctx *ctx, const double *x, size_t n, int hilbert_mode, double *env_out);

/* a8  wefax.py:196 np.percentile: exact order statistics sorted(env)[rank[i]] */
int wfx_order_stats(wfx_ctx *ctx, const double *env, size_t n,
                    const uint64_t *ranks, int nranks, double *out);

/* a8  wefax.py:198-200,216: clamp(rint(255*(env-low)/(high-low)), 0, 255);
 *     *nan_count receives the number of NaN results (the reference raises) */
int wfx_quantise(wfx_ctx *ctx, const double *env, size_t n, double low, double high,
                 uint8_t *out, uint64_t *nan_count);

/* a9  wefax.py:225-236: corr[i] = dot(pattern-128, d[i:i+L]-128), L = 2*n1+n0,
 *     i in [0, n-L) */
int wfx_sync_corr(wfx_ctx *ctx, const uint8_t *d, size_t n, int n1, int n0,
                  int32_t *corr_out);

/* a9  wefax.py:226-261 pattern_search: peak positions, the index at which each
 *     peak was first appended (progress messages, wefax.py:245), count, and
 *     whether the 100-peak limit was hit */
int wfx_sync_peaks(wfx_ctx *ctx, const uint8_t *d, size_t n, int n1, int n0,
                   int64_t mindistance, int64_t *peak_pos, int64_t *first_pos,
                   int *npeaks, int *hit_limit);

/* a10 wefax.py:296-327 __convert_to_image: rows of w samples from d[start:],
 *     255-v, then PIL resize((w, 4h)) (vertical bicubic, 22-bit fixed point).
 *     img must hold w*4*h bytes, h = (n-start)/w. */
int wfx_lines_to_image(wfx_ctx *ctx, const uint8_t *d, size_t n, size_t start,
                       int w, uint8_t *img);

/* ---- live path, one audio packet: data_packet.py:408-464 DataPacket.__process_samples ----
 * notch filtfilt with b/a designed at the packet's own sample rate (:420-434), |hilbert| + medfilt 3 (:436-448),
 * per-packet np.percentile(., (0.5, 99.5)) (ranks / lerp weights from the host), rint(255 (x - low) / (delta + 1e-6)),
 * clip (:450-464).  samples: WFX_IN_I16_MONO or WFX_IN_F64_MONO, host memory; out: n bytes; low / high optional. */
int wfx_packet_process(wfx_ctx *ctx, const void *samples, int in_kind, size_t n, const double b[3], const double a[3],
                       const uint64_t ranks[4], double gamma_lo, double gamma_hi, uint8_t *out, double *low, double *high);

/* `count` packets of n samples each, contiguous in `samples`: the same decode per packet, enqueued back to back with one
 * upload and one download (no host synchronisation between packets).  out: count * n bytes; low / high: count each. */
int wfx_packets_process(wfx_ctx *ctx, const void *samples, int in_kind, size_t n, size_t count, const double b[3], const double a[3],
                        const uint64_t ranks[4], double gamma_lo, double gamma_hi, uint8_t *out, double *low, double *high);

/* ---- live path, detectors: data_packet.py:388-406 DataPacket.__fourier_transform ----
 * amp_out[k] = |FFT(samples)[k] / (n / 2)| for k < n / 2 (any n >= 2).  The caller normalises by max + 1e-4 and runs
 * the peak conditions of contain_start_tone / contain_stop_tone / find_sync_pulse (:301-386) on the n / 2 values
 * (wefax_amd/detect.py).  samples: WFX_IN_I16_MONO or WFX_IN_F64_MONO, host memory. */
int wfx_packet_spectrum(wfx_ctx *ctx, const void *samples, int in_kind, size_t n, double *amp_out);

/* ---- fused whole path: input resident in HBM -> image resident in HBM --- */

typedef struct {
    int      in_kind;          /* wfx_in_kind                                     */
    uint64_t n0;               /* input frames                                    */
    uint64_t n;                /* samples at 11 025 Hz: n0, or int(11025*length)  */
    int      resample;         /* 1 when sample_rate != 11025 (wefax.py:60)       */
    double   notch_b[3];       /* scipy.signal.iirnotch(2600, 1, 11025)           */
    double   notch_a[3];
    int      hilbert_mode;     /* wfx_hilbert_mode                                */
    /* sharded decode only (ignored by the one-GPU calls): which plan wfx_shard_* takes.  0 = the library's cost model decides
     * (distributed transforms, or -- where their exchanges would take longer than one GPU needs for the whole capture -- the
     * single plan: rank 0 alone -- or, round 6, plan 3 where its model is ahead); 1 = distributed whenever a distributed form exists; 2 = single;
     * 3 = the chunk-local plan: contiguous arcs of the capture, the Hilbert transform (and, for a capture at another rate, the resampler) by
     * their multipole forms on the rank's own arc, kilobytes exchanged before the one gather (even length >= 32768 at 11 025 Hz; down-sampling).
     * Bit 4 (value 16) set: the ROWS layout of rounds 2-3 instead of the columns layout (A/B runs only: every form, the padded ones too, has
     * the columns layout) */
    int      shard_plan;
    /* np.percentile(., (0.5, 99.5)) 'linear': rank pairs and lerp weights        */
    uint64_t rank_lo[2];
    uint64_t rank_hi[2];
    double   gamma_lo;
    double   gamma_hi;
    /* sync search constants (wefax.py:223-229,264-267), computed by the host in
     * the same double arithmetic as the reference */
    int      n1, n0_gap;
    int64_t  mindistance;
    double   frame_samples;    /* frame_len * sample_rate (float, e.g. 5512.5)    */
    int      width;            /* int(frame_len * sample_rate)                    */
    /* filtfilt's odd extension (wefax.py:72: 9 samples before the first and after the last sample) when it is NOT what
     * 2 x[0] - x[k] gives in the arithmetic of the samples handed over: scipy evaluates it in the wav file's own dtype, so for
     * uint8 / int32 captures it wraps and for float32 ones it is rounded to float32, while such captures reach this library
     * converted to float64.  has_ext = 1: use ext_left[i] (extended position i - 9) and ext_right[i] (position n + i). */
    int      has_ext;
    double   ext_left[9];
    double   ext_right[9];
} wfx_decode_params;

typedef struct {
    uint64_t n;
    double   low, high;        /* the two percentiles                             */
    uint64_t nan_count;        /* > 0: the reference raises ValueError (int(nan)) */
    int      npeaks;
    int      hit_limit;
    int      no_group;         /* 1: wefax.py:294 max() of an empty list          */
    int      n_phasing;        /* len(phasing_signals)                            */
    int64_t  start_frame;
    int      width, height;    /* image is width x 4*height                       */
    int64_t  peak_pos[WFX_MAX_PEAKS + 1];
    int64_t  first_pos[WFX_MAX_PEAKS + 1];
    int64_t  phasing[WFX_MAX_PEAKS + 1];
} wfx_decode_info;

/* stage buffers that can be copied back after a decode */
typedef enum {
    WFX_BUF_AUDIO = 0,         /* double[n]  after merge/resample/notch  (a6)     */
    WFX_BUF_ENVELOPE = 1,      /* double[n]  after medfilt               (a7)     */
    WFX_BUF_DIGITAL = 2,       /* uint8[n]                               (a8)     */
    WFX_BUF_IMAGE = 3          /* uint8[4*height][width]                 (a10)    */
} wfx_buffer_id;

/* copy the capture into HBM (not part of the timed region) */
int wfx_decode_upload(wfx_ctx *ctx, const void *host_in, const wfx_decode_params *p);
/* the same from an open file: `in_bytes` of 16-bit PCM samples at `file_offset` of `fd` (the data chunk of a wav file: wefax.py:349)
 * go through `pinned` -- page-locked host memory of the caller, at least that large -- to the device in slices, the reads (a few
 * threads) overlapping the DMA.  WFX_IN_I16_MONO / WFX_IN_I16_STEREO only; a short file is WFX_ERR_SHORT_FILE ("Incomplete wav file") */
int wfx_decode_upload_fd(wfx_ctx *ctx, int fd, uint64_t file_offset, void *pinned, size_t pinned_bytes, const wfx_decode_params *p);
/* enqueue every kernel of the path on the context's stream (asynchronous) */
/* like wfx_decode_upload, but the capture already lives in DEVICE memory owned by the caller (e.g. the output of the
 * time-domain front end): nothing is copied, the pointer must stay valid while decodes run */
int wfx_decode_attach(wfx_ctx *ctx, const void *dev_in, const wfx_decode_params *p);
int wfx_decode_run(wfx_ctx *ctx);
/* wait and read back the scalars */
int wfx_decode_result(wfx_ctx *ctx, wfx_decode_info *info);
/* diagnostics of the last decode (not part of the reference's interface): out[7] = which form of the peak scan
 * (wefax.py:226-261) produced the peaks: 1 = scans of overlapping segments joined at a common appended peak,
 * -1 = the join failed and the sequential scan ran, 0 = sequential scan only; out[0..6]: cycle stamps of a
 * `--pick-stats` build, else 0 */
int wfx_debug_counters(wfx_ctx *ctx, long long out[8]);
/* copy one stage buffer to the host (bytes must match the buffer's size) */
int wfx_decode_fetch(wfx_ctx *ctx, int buffer_id, void *host_out, size_t bytes);
/* device address of a stage buffer (for collectives on the final image) */
int wfx_decode_device_ptr(wfx_ctx *ctx, int buffer_id, void **dev_ptr, size_t *bytes);
/* copy a stage buffer into caller-owned DEVICE memory (e.g. the send buffer of a
 * collective) and wait for it; at most `capacity` bytes, *copied receives the size */
int wfx_decode_copy_to_device(wfx_ctx *ctx, int buffer_id, void *dst_dev, size_t capacity, size_t *copied);

/* save_output_image (wefax.py:407-408) without the host-side encoder: the PNG file of the decode's image -- 8-bit gray, width x 4 height,
 * no interlace, filter 0, STORED deflate blocks (valid PNG, identical pixels, not compressed) -- is assembled on the device (stream layout,
 * Adler-32 row sums, CRC-32 segments) and finished on the host in closed form.  wfx_decode_png returns the file image in pinned
 * memory owned by the context (valid until the next call); wfx_decode_save_png writes it to `path`. */
int wfx_decode_png(wfx_ctx *ctx, const void **file_bytes, size_t *nbytes);
int wfx_decode_save_png(wfx_ctx *ctx, const char *path, size_t *bytes_written);
/* the same with deflate != 0: the COMPRESSED file, also encoded on the device -- rows "Up"-filtered, chunks of whole rows as
 * dynamic-Huffman deflate blocks (literals + distance-1 runs, one code per image built from the token histogram, chunks joined by
 * empty stored blocks), i.e. what PIL's zlib stream is for the reference (wefax.py:408: identical pixels, a file of the same order
 * of size).  deflate == 0: the stored form above */
int wfx_decode_png_ex(wfx_ctx *ctx, int deflate, const void **file_bytes, size_t *nbytes);
int wfx_decode_save_png_ex(wfx_ctx *ctx, const char *path, int deflate, size_t *bytes_written);
/* page-locked host memory: captures uploaded from it and images fetched into it cross PCIe by DMA (no staging copies) */
void *wfx_host_alloc(size_t bytes);
void  wfx_host_free(void *p);
/* Planning queries (host only, no GPU needed).  wfx_plan_padded_length: the transform length the analytic-signal convolution of
 * a capture takes when N/2 is not 13-smooth -- the cheapest 13-smooth M >= min_len (= N - 1) with a radix-pair plan, at most
 * 8 % above it; 0 when min_len is below 4096 (such captures pad to a power of two).  wfx_plan_describe: the passes of the
 * transform of length L as text ("7x13,7x25,15x15"; "" when L is not 13-smooth), returns the number of passes. */
uint64_t wfx_plan_padded_length(uint64_t min_len);
/* 1: the transform-based resampler takes n0 -> num directly (both half-lengths 13-smooth: mixed-radix passes); 0: it needs the chirp-z form
 * (any length, ~3.5x the time) -- where the multipole form of the resampler exists (wfx_d_resample_fmm) that one is the faster */
int wfx_plan_resample_direct(uint64_t n0, uint64_t num);
int wfx_plan_describe(uint64_t L, char *buf, int cap);
/* copy a new capture of the same description into the context (asynchronous when host_in is pinned memory).  `bytes` must equal
 * the size of the capture described to wfx_decode_upload (n0 frames of in_kind) -- WFX_ERR_BAD_ARG otherwise, nothing is read.
 * ext_left / ext_right (9 values each, or both NULL): filtfilt's odd extension of THIS capture for float64 hand-overs of
 * uint8 / int32 / float32 files (wfx_decode_params.ext_left/right); NULL clears has_ext.  Results of the previous capture stop
 * being fetchable (WFX_ERR_STATE until the next wfx_decode_run). */
int wfx_decode_reload(wfx_ctx *ctx, const void *host_in, size_t bytes, const double *ext_left, const double *ext_right);
/* enqueue a copy of a stage buffer to the host WITHOUT waiting (host_out should be pinned); wfx_sync / wfx_decode_result waits.
 * For WFX_BUF_IMAGE `bytes` must be >= the largest possible image (4 * width * (n / width)); the height arrives with the result. */
int wfx_decode_fetch_async(wfx_ctx *ctx, int buffer_id, void *host_out, size_t bytes);

/* Let the NEXT decodes write {int64 bytes, int64 width, image} straight into caller-owned device memory (e.g. the send
 * slot of a collective): no copy after the decode, wfx_decode_export_async of WFX_BUF_IMAGE to the same address becomes
 * a no-op.  Used when capacity >= 16 + 4 * width * (n / width); NULL unbinds.  The pointer must stay valid while
 * decodes run.  (No counterpart in the reference, which decodes on one host.) */
int wfx_decode_bind_image(wfx_ctx *ctx, void *dst_dev, size_t capacity);
/* Asynchronous hand-over to a collective (one process per GPU, images gathered by RCCL while the next decode
 * runs): wfx_stream_handle returns the context's hipStream_t so that the caller can order its own stream against
 * it with events; wfx_decode_export_async enqueues on that stream, WITHOUT waiting, a 16-byte header
 * {int64 bytes, int64 width} built on the device followed by the stage buffer of the decode in flight
 * (at most capacity - 16 bytes) into caller-owned DEVICE memory. */
int wfx_stream_handle(wfx_ctx *ctx, void **stream);
int wfx_decode_export_async(wfx_ctx *ctx, int buffer_id, void *dst_dev, size_t capacity);

/* ---- device-resident stage calls (sample-range sharding of one capture) -------
 * Building blocks for decoding a slice [g0, g0+n) of a long capture on one GPU with
 * halos on both sides: every call takes DEVICE pointers obtained
 * from wfx_dev_malloc and runs on the context's stream.  The FIR forms are the
 * halo-local operators; results within the stated margin of the slice ends are invalid
 * and must be covered by the caller's halo. */
int wfx_dev_malloc(wfx_ctx *ctx, size_t bytes, void **dev_ptr);
int wfx_dev_free(wfx_ctx *ctx, void *dev_ptr);
int wfx_dev_upload(wfx_ctx *ctx, void *dst_dev, const void *src_host, size_t bytes);
int wfx_dev_download(wfx_ctx *ctx, void *dst_host, const void *src_dev, size_t bytes);
int wfx_dev_copy(wfx_ctx *ctx, void *dst_dev, const void *src_dev, size_t bytes);     /* device to device, waits */
/* a6 on a segment: 49-tap symmetric FIR == filtfilt away from the ends (valid for 24 <= i < n-24); with
 * edge_flags bit 0 / bit 1 the segment starts / ends at the capture's true start / end and gets filtfilt's exact edge */
int wfx_d_notch_fir(wfx_ctx *ctx, const int16_t *in_dev, size_t n, const double b[3], const double a[3], double *out_dev,
                    int edge_flags);
int wfx_d_notch_fir_f64(wfx_ctx *ctx, const double *in_dev, size_t n, const double b[3], const double a[3], double *out_dev,
                        int edge_flags);
/* a4 + a5 in halo-local form (oversampled captures, BASELINE configs[3]): the time-domain counterpart of wefax.py:360-394 down to
 * a hand-over rate above 11 025 Hz (the exact FFT resampler of the decode path takes the last step).  One stencil,
 *   out[i] = sum_{j<ntaps} coef[j] * in[first + i*factor + j],  factor <= 64, samples of `in` outside [0, n_in) read as zero,
 * with float64 taps and a float64 result.  in_kind WFX_IN_I16_MONO, WFX_IN_I16_STEREO (= interleaved IQ; merged as (int16)(L+R)/2
 * while loading) or WFX_IN_F64_MONO; coef is a HOST pointer, designed by the caller (wefax_amd/polyphase.py).  With fix_shift =
 * s > 0, int16 input and a power-of-two factor >= the samples per 16 bytes (8 mono, 4 IQ) the stencil is computed EXACTLY: the
 * taps are rounded to multiples of 2^-s and the products summed as integers (v_dot2_i32_i16) --
 *   out[i] = sum_j round(coef[j] * 2^s) * in[first + i*factor + j] / 2^s  (IQ: of the wrapped int16 sum, / 2)
 * with no rounding of the sum; *exact (may be NULL) is set to 1.  s must keep every tap below 2^27 steps and the sums of a
 * flush window inside 32 bits for the worst-case input (wefax_amd/polyphase.fix_shift_for picks it; 30 for the ingest filters).
 * Everything else (fix_shift 0, other inputs / factors, a shift the taps do not fit) runs float64 FMAs in one fixed order per
 * output (*exact = 0): |error| <= ntaps * 2^-53 * sum |coef * in|.  (The fp32 forms of rounds 1-2, wfx_d_decimate_fir and
 * wfx_d_resample_rational, were removed in round 4.) */
int wfx_d_decimate_fir64(wfx_ctx *ctx, const void *in_dev, int in_kind, size_t n_in, int64_t first, int factor, const double *coef,
                         int ntaps, double *out_dev, size_t n_out, int fix_shift, int *exact);
/* `nbatch` equally shaped jobs in ONE launch (the segments a rank owns in the columns layout of the sharded decode): member b
 * reads in_dev + b * in_stride frames (b * in_stride * frame bytes must be a multiple of 16) and writes out_dev + b * out_stride */
int wfx_d_decimate_fir64_batch(wfx_ctx *ctx, const void *in_dev, int in_kind, size_t n_in, int64_t first, int factor, const double *coef,
                               int ntaps, double *out_dev, size_t n_out, int fix_shift, int *exact, int nbatch, size_t in_stride, size_t out_stride);
/* The first TWO stages of such a chain in one kernel that reads the capture once and keeps the intermediate rate on chip (the
 * 1.536 MS/s ingest: / 32 integer-exact, / 3 float64 behind it):
 *   y1[i] = sum_j round(coef1[j] 2^s) * in[factor i + j] / 2^s,   out[k] = sum_j coef2[j] * y1[factor2 k + j]   (factor2 = 0: out = y1)
 * with `in` 16-byte aligned, frames beyond n_in reading as zero, and results BIT-IDENTICAL to the two wfx_d_decimate_fir64 calls
 * it replaces.  *handled = 0 and nothing enqueued for shapes it is not built for (factor != 32, factor2 not in {0, 2, 3}, more than
 * 256 / 120 taps, a misaligned pointer, taps the grid does not hold): the caller then makes those two calls.  Batches as above. */
int wfx_d_ingest_chain(wfx_ctx *ctx, const void *in_dev, int in_kind, size_t n_in, int factor, const double *coef1, int ntaps1, int fix_shift,
                       int factor2, const double *coef2, int ntaps2, double *out_dev, size_t n_out, int nbatch, size_t in_stride, size_t out_stride,
                       int *handled);
/* a7 on device memory without a transform over the whole capture: out = H = imag(scipy.signal.hilbert(x)) (out_env 0), |x + i H| (1) or
 * the 5-tap median of that envelope, zeros beyond both ends (2: wefax.py:174-175 complete), by the fast multipole form of WFX_HILBERT_FMM; *handled = 0 and nothing enqueued for lengths it does not take */
int wfx_d_hilbert_fmm(wfx_ctx *ctx, const double *x_dev, size_t n, double *out_dev, int out_env, int *handled);
/* a5's resampler (wefax.py:160-161: scipy.signal.resample(x, num)) on device memory without a transform over the capture: y_dev[0 .. num) from the
 * n0 real samples x_dev, by the multipole form of the periodic sinc sum (downsampling to an even count: the IQ hand-over and the 48 kHz captures);
 * *handled = 0 and nothing enqueued for other lengths (upsampling, odd counts, < 32768 samples: the transform route serves those) */
int wfx_d_resample_fmm(wfx_ctx *ctx, const double *x_dev, size_t n0, size_t num, double *y_dev, int *handled);
/* measurement aid: GB/s at which this GPU reads `bytes` (>= 1 MiB, 16-byte aligned) of device memory with a kernel that only loads
 * (16-byte loads, 16 in flight per lane, 64 KiB blocks), best of `reps` launches by HIP events; waits for the stream.  bench.py puts
 * the ingest kernel's rate beside it: a slow box shows here, a slow kernel in the ratio */
int wfx_d_read_rate(wfx_ctx *ctx, const void *dev, size_t bytes, int reps, double *gbs);
/* measurement aid: GB/s (input bytes per second) at which the streaming ingest itself (wfx_d_ingest_chain's / 32 -> / 3 kernel, outputs
 * into out_dev -- bytes / 48 of them -- or, out_dev == NULL, into a scratch allocation) works through `bytes` (>= 64 MiB, 16-byte aligned) of device memory taken as an IQ capture.
 * The kernel's time does not depend on the data but -- unlike the dense sweep above -- on WHERE the allocation lies (up to 15 %
 * between two buffers of one process; the output's place counts too): a caller that keeps its buffers can time a few allocations and keep the best
 * (wefax_amd/_native.py: Context.dev_malloc_placed) */
int wfx_d_stream_rate(wfx_ctx *ctx, const void *dev, size_t bytes, double *out_dev, int reps, double *gbs);
int wfx_d_median5(wfx_ctx *ctx, const double *in_dev, size_t n, double *out_dev);
/* a8: one level of the radix select: hist_dev[q*2048 + digit] += count over values whose bits above the level equal prefix[q] */
int wfx_d_select_hist(wfx_ctx *ctx, const double *env_dev, size_t n, int level, const uint64_t prefix[4], uint32_t *hist_dev);
int wfx_d_quantise(wfx_ctx *ctx, const double *env_dev, size_t n, double low, double high, uint8_t *out_dev, uint64_t *nan_count);
/* a9 on a device-resident stream: fills peaks, phasing, start_frame, no_group, height = (n_total - start_frame) / width */
int wfx_d_sync_search(wfx_ctx *ctx, const uint8_t *d_dev, size_t n, size_t n_total, int n1, int n0_gap, int64_t mindistance,
                      double frame_samples, int width, wfx_decode_info *info);
/* a10: output rows 4*y0 .. 4*(y0+rows) of the image of a capture with h_total lines whose line 0 starts at sample
 * `start`; d_dev[0] is global sample g0 (lines y0-2 .. y0+rows+1 must lie inside the slice) */
int wfx_d_image_rows(wfx_ctx *ctx, const uint8_t *d_dev, size_t n, uint64_t g0, uint64_t start, int width, int h_total, int y0,
                     int rows, uint8_t *img_dev);

/* ---- one capture over several GPUs: the exact path, sharded by sample range ---------------------------------
 * SURVEY.md section 8b item (4) / 8e.  The reference decodes on one host; its two global operators -- scipy.signal.hilbert
 * (wefax.py:174) and scipy.signal.resample (wefax.py:384) -- are computed here as DISTRIBUTED exact transforms (two
 * transposes per transform over RCCL), everything else on each rank's own sample range with a halo:
 *   rank r owns samples [own_lo, own_hi) at 11 025 Hz (and the matching input frames): [resample] -> notch filtfilt ->
 *   Hilbert envelope + median 5 -> np.percentile by radix select (two all-reduces of histograms + one all-gather of candidate
 *   keys) -> quantise -> ONE gather of the uint8 stream to rank 0 -> sync search + bicubic image on rank 0.
 * The result is the single-GPU exact path's (identical uint8 stream; float stages to rounding) and bit-identical for every
 * world size.  All collectives are enqueued on the context's stream by the library itself; the only host synchronisation
 * of a decode is the one in wfx_shard_result.
 *
 * Communicator: RCCL, bound directly.  Rank 0 calls wfx_comm_unique_id and passes the 128 bytes to the other processes
 * (wefax_amd/sharded.py: a loopback TCP socket); every rank then calls wfx_comm_create with its context.  Failures of the
 * transport return WFX_ERR_COMM.  wfx_comm_create_local makes `world` communicators whose ranks all live in the calling
 * process (contexts on any device): collectives complete when the last rank has posted its part -- drive the ranks phase
 * by phase with wfx_shard_phase.  It exists to run the N-rank form on a one-GPU box (tests, emulation). */
typedef struct wfx_comm wfx_comm;
typedef struct wfx_shard wfx_shard;
#define WFX_COMM_ID_BYTES 128
int wfx_comm_unique_id(void *id_out /* WFX_COMM_ID_BYTES */);
int wfx_comm_create(wfx_ctx *ctx, const void *id, int world, int rank, wfx_comm **out);
int wfx_comm_create_local(int world, wfx_comm **out /* [world] */);
/* Third transport: one PROCESS per rank on one host, messages staged through POSIX shared memory (/dev/shm/wfx_<job>_*), two
 * barriers per collective.  For boxes where RCCL cannot run the job (several ranks on ONE GPU) and for tests: real processes,
 * contexts and streams, ranks out of step with each other.  `job`: a name the ranks of one job share (letters, digits, '-',
 * '.'); rank 0 creates the control block, the others wait for it up to timeout_s.  ctx may be NULL: the collectives then move
 * HOST memory (protocol tests on a machine without a GPU).  A peer that dies, or disagrees about a message, is WFX_ERR_COMM
 * on every rank within timeout_s. */
int wfx_comm_create_shm(wfx_ctx *ctx, const char *job, int world, int rank, double timeout_s, wfx_comm **out);
/* `rounds` rounds of randomised exchanges / all-reduces / all-gathers with known answers and random pauses (every rank derives
 * the same plan from `seed`); RCCL or shm communicators; device buffers, or host buffers when ctx is NULL (shm only) */
int wfx_comm_selftest(wfx_comm *comm, wfx_ctx *ctx, int rounds, uint64_t seed);
int wfx_comm_info(wfx_comm *comm, int *world, int *rank, int *is_rccl);
int wfx_comm_destroy(wfx_comm *comm);
/* for drivers: wait until every rank has arrived (and this rank's stream is idle); gather `bytes` host bytes of every rank
 * into recv_host[world * bytes] on every rank (timings, checksums).  RCCL communicators and world 1 only. */
int wfx_comm_barrier(wfx_comm *comm, wfx_ctx *ctx);
int wfx_comm_allgather_host(wfx_comm *comm, wfx_ctx *ctx, const void *send_host, void *recv_host, size_t bytes);

typedef struct {
    int      world, rank;
    int      first_radix[2];   /* radix pair of the distributed first pass                                           */
    uint64_t in_lo, in_hi;     /* input frames [in_lo, in_hi) this rank must provide (global indices)                 */
    uint64_t own_lo, own_hi;   /* samples at 11 025 Hz this rank owns                                                 */
    /* Round 4, the COLUMNS layout (nseg > 1): a rank owns nseg equally long, equally spaced SEGMENTS instead of one range --
     * segment s = samples [own_lo + s * own_seg_stride, + own_seg_len) at 11 025 Hz (own_hi = the end of the last one), and
     * provides the input frames [in_lo + s * in_seg_stride - in_halo, in_lo + s * in_seg_stride + in_seg_len + in_halo) of
     * every segment, back to back: nseg * (in_seg_len + 2 in_halo) frames (frames outside the capture: anything).  The stage
     * buffers wfx_shard_fetch returns are the segments back to back.  nseg == 1: one range, as in_lo / in_hi / own_lo /
     * own_hi say (rows layout, single plan, one rank).  A capture of arbitrary length at 11 025 Hz (padded forms) is laid out on
     * a longer arrangement: its last segments reach past the capture's end -- provide zeros (anything) there, and ignore the slots
     * of the stage buffers whose sample index is >= n; in_halo is 192 for those (32 otherwise). */
    int      nseg, in_halo;
    uint64_t in_seg_len, in_seg_stride, own_seg_len, own_seg_stride;
    int      plan;             /* 0 single (rank 0 alone), 1 rows layout, 2 columns layout, 3 chunk-local multipole forms: ONE range
                                  [in_lo, in_hi) with in_halo frames on either side taken ROUND THE CIRCLE (320 at 11 025 Hz, 64 in
                                  front of a resampler: in_hi - in_lo + 2 in_halo frames in all)                         */
    int      plan_forced;      /* 1: wfx_decode_params.shard_plan asked for it; 0: the cost model chose                   */
    /* the cost model's figures for this capture and world size, seconds (0 when no distributed form exists) */
    double   model_single_s, model_dist_compute_s, model_dist_wire_s;
    uint64_t model_wire_bytes; /* bytes all ranks put on the wire per decode in the distributed form                     */
    char     plan_reason[160];
} wfx_shard_layout;

/* host only (no GPU needed): how a capture described by `p` is cut for `world` ranks.  Captures at 11 025 Hz shard at ANY
 * length (odd ones too: every own_lo / own_hi is even except the last rank's own_hi = n).  A capture
 * with no distributed form -- resampled with odd or non-13-smooth half-lengths, or too short for the world size -- gets the
 * SINGLE plan: first_radix = {0, 0}, rank 0 owns (and must provide) the whole capture and decodes it alone with the fused
 * one-GPU path, the other ranks provide nothing and receive the scalars; the calls below behave the same either way.
 * WFX_ERR_BAD_ARG only for parameters no path accepts */
int wfx_shard_layout_query(const wfx_decode_params *p, int world, int rank, wfx_shard_layout *out);
/* host only: builds every rank's exchange lists for `world` ranks with fake buffer addresses and checks that the two ends of
 * every message agree, that what a rank receives tiles its buffers exactly, and that every packing copy stays inside its
 * buffers (WFX_ERR_COMM with a description otherwise).  Lets a deployment -- and the CPU test suite -- validate a plan for
 * capture sizes and world sizes it cannot run */
int wfx_shard_dry_run(const wfx_decode_params *p, int world);
/* host only: what the plan puts on the wire.  One entry per collective of a decode, in order: name (<= 23 characters), the
 * bytes all ranks together send to OTHER ranks, the most any one rank sends, and the most that crosses one directed link
 * (rank a -> rank b).  Returns the number of collectives (entries beyond `cap` are not written), negative on error. */
typedef struct {
    char     name[24];
    uint64_t total_bytes, max_rank_bytes, max_link_bytes;
} wfx_wire_entry;
int wfx_shard_wire_plan(const wfx_decode_params *p, int world, wfx_wire_entry *out, int cap);
/* what this rank's communicator has moved since the last reset: one entry per collective in call order (ring buffer of the
 * last 256), total_bytes = bytes THIS rank sent to other ranks, max_rank_bytes = bytes it received from them, max_link_bytes =
 * the largest message; transport-independent (RCCL, shm and local communicators count alike).  Returns the number of
 * collectives since the reset. */
/* exchanges this communicator has run on its own stream so far (RCCL: the k1 subsets' E2 / E3, overlapping the slab passes of other
 * subsets; 0 on the transports that complete an exchange before returning) */
uint64_t wfx_comm_async_exchanges(wfx_comm *comm);
int wfx_comm_wire_reset(wfx_comm *comm);
int wfx_comm_wire_stats(wfx_comm *comm, wfx_wire_entry *out, int cap);
/* Time per collective, parallel to the entries of wfx_comm_wire_stats.  wfx_comm_wire_timing(comm, 1) resets the records and from
 * then on brackets every collective with a HIP-event pair on the stream it is enqueued on (RCCL) or reads the host clock around
 * it (the shm / local transports, which complete it before returning); for an exchange on the communicator's own stream a second
 * pair on the CONTEXT's stream brackets the wait for it.  After the streams have been synchronised wfx_comm_wire_times reports
 *   us       the collective itself (RCCL: from its first kernel's start to its last one's end -- waiting for a late peer included),
 *   wait_us  what the compute stream spent standing still for it: = us for a collective in stream order or on a blocking
 *            transport; for an overlapped exchange the time between reaching its wfx_comm_wait and being released (0 when it
 *            had already finished: fully hidden),
 * -1 where nothing was measured.  timed: 0 no, 1 HIP events, 2 host clock.  Returns the number of entries. */
typedef struct {
    double us, wait_us;
    int    on_comm_stream, timed;
} wfx_wire_time;
int wfx_comm_wire_timing(wfx_comm *comm, int on);
int wfx_comm_wire_times(wfx_comm *comm, wfx_wire_time *out, int cap);
/* `p` describes the WHOLE capture (as for wfx_decode_upload); hilbert_mode must be WFX_HILBERT_FFT */
int wfx_shard_create(wfx_ctx *ctx, wfx_comm *comm, const wfx_decode_params *p, wfx_shard **out);
/* this rank's input frames [in_lo, in_hi): host memory (copied) or caller-owned device memory (kept, not copied) */
int wfx_shard_upload(wfx_shard *sh, const void *host_frames);
int wfx_shard_attach(wfx_shard *sh, const void *dev_frames);
/* the decode is a fixed sequence of phases, each ending in at most one collective; wfx_decode_sharded enqueues all of them
 * (RCCL communicator, or world 1); with a local communicator call phase p for every rank before phase p + 1 for any */
int wfx_shard_phase_count(wfx_shard *sh);
int wfx_shard_phase(wfx_shard *sh, int phase);
int wfx_decode_sharded(wfx_shard *sh);
/* waits for this rank's stream.  Rank 0: the decode's scalars (as wfx_decode_result); other ranks: n, width, low, high only */
int wfx_shard_result(wfx_shard *sh, wfx_decode_info *info);
/* stage buffers: WFX_BUF_AUDIO / ENVELOPE / DIGITAL = this rank's own samples [own_lo, own_hi); WFX_BUF_IMAGE and the
 * whole WFX_BUF_DIGITAL stream (bytes == n) on rank 0 only */
int wfx_shard_fetch(wfx_shard *sh, int buffer_id, void *host_out, size_t bytes);
int wfx_shard_destroy(wfx_shard *sh);

/* ---- measurement ------------------------------------------------------ */
/* Test-signal synthesis straight into HBM (not on the decode path; the reference ships no generator): the WEFAX transmission
 * of wefax_amd/synth.py -- start tone, phasing lines, ramp image lines, stop tone, black tail, phase-continuous FM + white
 * noise -> int16 mono (iq = 0) or interleaved I/Q (iq = 1) -- for frames [lo, hi) of a capture of wfx_synth_frames() frames.
 * BASELINE configs[3] (60 minutes at 1.536 MS/s IQ, 22 GB) exists only this way. */
typedef struct {
    double   sample_rate;
    int      lines_per_minute, ioc;             /* ioc 576: 300 Hz start tone, 288: 675 Hz (README.md:27-97 of the reference) */
    double   start_tone_s;
    int      phasing_lines, image_lines;
    double   stop_tone_s, black_tail_s;
    double   amplitude, noise;                  /* in units of full scale */
    uint64_t seed;
    int      iq;
} wfx_synth_params;
uint64_t wfx_synth_frames(const wfx_synth_params *p);
int wfx_synth_capture(wfx_ctx *ctx, const wfx_synth_params *p, uint64_t lo, uint64_t hi, void *dev_out);

/* HIP-event stopwatch on the context's stream */
int wfx_timer_start(wfx_ctx *ctx);
int wfx_timer_stop(wfx_ctx *ctx, float *ms);
/* per-kernel HIP-event timing: enable, run, then read (count, total ms) by name;
 * names are listed by wfx_profile_kernel_name(i), i in [0, wfx_profile_kernel_count()) */
int         wfx_profile_enable(wfx_ctx *ctx, int on);
int         wfx_profile_reset(wfx_ctx *ctx);
int         wfx_profile_kernel_count(void);
const char *wfx_profile_kernel_name(int i);
int         wfx_profile_get(wfx_ctx *ctx, int i, uint64_t *launches, double *total_ms);

#ifdef __cplusplus
}
#endif
#endif /* WEFAX_HIP_H */
