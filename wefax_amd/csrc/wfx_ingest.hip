// Streaming ingest of an oversampled int16 capture (BASELINE configs[3]: 1.536 MS/s IQ; round 5): the time-domain counterpart of
// wefax.py:360-394 (stereo merge + scipy FFT resample) for the first one or two stages of the front end's chain,
//
//   y1[i] = sum_j c1[j] * x[32 i + j]          int16 samples (IQ pairs merged as (int16)(I + Q), wefax.py:360-373), integer-exact
//   y2[k] = sum_j c2[j] * y1[M2 k + j]         float64, the stage behind it (M2 = 3 for 1.536 MS/s -> 48 kHz -> 16 kHz), optional
//
// in ONE kernel that reads every input byte once and never writes y1 to memory.  It replaces the tile kernel of
// wfx_polyphase.hip (`decimate_kernel<..., MODE 1>` + a second `MODE 2` launch) on that path; the results are bit-identical to
// that pair: stage 1 is an integer sum (no rounding, any order), stage 2 keeps the tile kernel's canonical order (three polyphase
// FMA chains in ascending tap order, summed row 0 + row 1 + row 2).
//
// Why another kernel: the tile kernel kept its window in LDS in POLYPHASE order (row = sample index mod 32) so that one window
// read fed many taps, and paid for it with a 2-byte scatter of every input sample (four ds_write_b16 per IQ chunk) and tiles
// that re-staged their halo; its time did not move with the tap count (profiles/r03_v4/ingest_probe.txt: 32 taps as slow as 256)
// -- the staging, not the arithmetic, kept it at 4.5-5.5 TB/s.  Here a workgroup STREAMS through a long contiguous run:
//
//   * samples lie in LDS in their natural order as two BYTE PLANES: a sample s = 256 hi + (lo ^ 0x80) + 128 with hi, lo signed bytes;
//     one row = the 32 frames of one decimation step = 16 dwords: per 16 samples four dwords of high bytes, then four of low ones.
//     A 16-byte IQ chunk is merged (two v_perm + one v_pk_add_u16 per pair), split (two v_perm, one v_xor) and stored with one
//     ds_write2_b32;
//   * the 256-tap window sums run on the matrix cores, every sample read from LDS ONCE per byte plane: v_mfma_i32_16x16x64_i8 with
//     the TAPS as the A operand -- a fixed-point tap is four balanced signed bytes t = q0 + 2^8 q1 + 2^16 q2 + 2^24 q3; row 4 c + q
//     of A = digit plane q of taps 64 c .. 64 c + 63 (the whole filter: 16 rows x 64 = four registers per lane, parked in 1 KiB of LDS between the iterations)
//     -- and 16 consecutive LDS rows R as the B operand's columns (lane (j, g): the 16 bytes at sample 16 g of the two-row window
//     that starts at row R0 + j, one ds_read_b128).  Column j of the product is then what rows R, R + 1 contribute, through tap
//     chunk c, to output R - 2 c: lane (g = c, j) holds that chunk's four digit sums in its four accumulator registers, folds
//     them (sum_q 2^(8q) (2^8 Dh[q] + Dl[q]), 64-bit) and adds the result to the output's cell of an LDS array with ds_add_u64 --
//     integer sums, any order, exact for every input (a digit sum is at most 64 x 128 x 128 = 2^20).  Two MFMAs, two window reads
//     and one atomic per 16 rows; 128 sum(taps) for the low plane's offset joins when the cell is converted to float64.
//     (Round 5 first ran this stage on v_dot2_i32_i16 with the taps as scalar operands, then as MFMAs over windows read once per
//     output: both read every sample 8 x from LDS, and with noisy data that -- not the arithmetic -- pulled the shader clock from
//     2.36 to 2.02 GHz and the whole kernel, loads included, with it: docs/history/EXPERIMENTS_rounds1-5.md §9.)
//   * an iteration handles 512 outputs = 64 KiB of IQ frames; the next block's 16 chunks per lane are requested right after this
//     block's registers were stored to LDS and stay in flight during the whole compute phase; the last 8 rows are carried over
//     as the next iteration's halo (128 dwords through registers), so nothing is read twice inside a run;
//   * stage 2 runs from a 1024-entry LDS ring of y1 one iteration behind stage 1 (<= 171 outputs per iteration).
//
// Roofline: HBM.  Algorithmic bytes per launch = 4 B per IQ frame + 8 B per output.
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <type_traits>

#include "wfx_internal.h"

namespace {

#ifndef WFX_IG_THREADS
#define WFX_IG_THREADS 256
#endif
constexpr int IG_THREADS = WFX_IG_THREADS;      // (build-time A/B switch; 128 threads x 256 outputs, six workgroups per CU, measured 3 % slower)
constexpr int IG_M = 32;                        // stage-1 factor = frames per LDS row
constexpr int IG_BLK = 2 * IG_THREADS;          // stage-1 outputs (= new LDS rows) per iteration: two per thread
constexpr int IG_HALO = 8;                      // rows of history: a window spans 8 rows (<= 256 taps)
constexpr int IG_ROWS = IG_BLK + IG_HALO;
constexpr int IG_RD = IG_M / 2;                 // dwords per row
constexpr int IG_YRING = 2 * IG_BLK, IG_YMIRROR = 128;      // (a power of two >= one iteration's outputs + a stage-2 window)
static_assert(IG_YRING >= IG_BLK + IG_YMIRROR && IG_THREADS >= IG_HALO * (IG_M / 2), "ring / carry sizes");
#ifndef WFX_IG_LOAD_AUX
#define WFX_IG_LOAD_AUX 2
#endif
constexpr int IG_LOAD_AUX = WFX_IG_LOAD_AUX;      // cache policy of the block loads: 2 = nt (every byte is read once; measured 1.5 % over the default, 0)
constexpr int IG_STORE_AUX = 2;                 // cache policy of stage 2's stores: nt (0.1-0.2 ms of 3.6 on the 60-minute stream over the default)
constexpr int IG_KC = 4;                        // chunks of 64 taps per window = row groups of the MFMA's A operand
constexpr int IG_TAB = 64 * 4;                  // tap table: [lane][4 dwords]
constexpr int IG_TILES = IG_BLK / 16 + 1;       // tiles of 16 rows per iteration: rows 0 .. IG_BLK + 6 start a two-row window somebody needs

__host__ __device__ constexpr int ig_row_off(int r) { return r * IG_RD + (r >> 1) * 4; }       // dword offset of row r
constexpr int IG_XS_BYTES = ig_row_off(IG_ROWS) * 4;
constexpr int IG_C2PAD = 120;                   // stage-2 taps in LDS, zero padded to whole trips of 12
constexpr int IG_YS_BYTES = (IG_YRING + IG_YMIRROR + IG_C2PAD) * 8;

typedef unsigned short ig_us2 __attribute__((ext_vector_type(2)));
typedef int ig_v4i __attribute__((ext_vector_type(4)));
// two IQ frames (I | Q << 16 each) -> one dword of two merged samples, (int16)(I + Q) with the int16 wrap of wefax.py:367
__device__ __forceinline__ unsigned ig_merge2(unsigned w0, unsigned w1)
{
    const unsigned lo = __builtin_amdgcn_perm(w1, w0, 0x05040100u);      // (I0, I1)
    const unsigned hi = __builtin_amdgcn_perm(w1, w0, 0x07060302u);      // (Q0, Q1)
    return __builtin_bit_cast(unsigned, (ig_us2)(__builtin_bit_cast(ig_us2, lo) + __builtin_bit_cast(ig_us2, hi)));
}
// four samples (two dwords of int16 pairs) -> their high bytes and their low bytes ^ 0x80, sample k in byte k
__device__ __forceinline__ void ig_planes(unsigned m0, unsigned m1, unsigned &hi, unsigned &lo)
{
    hi = __builtin_amdgcn_perm(m1, m0, 0x07050301u);
    lo = __builtin_amdgcn_perm(m1, m0, 0x06040200u) ^ 0x80808080u;
}

struct ig_params {
    const void *in;
    long long n_in;             // frames readable behind `in`; anything beyond reads as zero
    double sc;                  // 2^-shift (IQ: half of it)
    int nper, ntaps2;           // stage 2: taps (nper: the tile kernel's padded row length, unused here)
    double *out;
    long long n_out;            // outputs of the last stage
    long long run_out;          // outputs per workgroup run
    long long in_bs, out_bs;    // batch strides (bytes / elements)
    long long out0;             // first output of the launch's first run
    long long bias;             // 128 * sum of the fixed-point taps (the low plane is stored less 128)
    unsigned long long *dbg_clk;        // lab builds only (-DWFX_LAB; WFX_INGEST_CLK=1): every 64th run's (shader clocks, 100 MHz ticks) -> the effective shader clock
    int dbg_flags;              // lab builds only (-DWFX_LAB; WFX_INGEST_DBG=flags; results are WRONG unless 0): 1 no stash, 2 no stage 2, 4 no stage 1, 8 no barrier B; WFX_INGEST_DBG_LDS: extra LDS bytes
};

// frames [e0, e0 + FPC) as one 16-byte chunk; frames at or beyond n_in read as zero
template <int FB>
__device__ __forceinline__ uint4 ig_fetch_guarded(const unsigned char *in, long long e0, long long n_in)
{
    constexpr int FPC = 16 / FB;
    if (e0 + FPC <= n_in) return *(const uint4 *)(in + e0 * FB);
    unsigned w[4] = {0u, 0u, 0u, 0u};
    if (FB == 4) {
#pragma unroll
        for (int k = 0; k < 4; ++k)
            if (e0 + k < n_in) w[k] = ((const unsigned *)in)[e0 + k];
    } else {
#pragma unroll
        for (int k = 0; k < 8; ++k)
            if (e0 + k < n_in) w[k >> 1] |= (unsigned)((const unsigned short *)in)[e0 + k] << (16 * (k & 1));
    }
    return make_uint4(w[0], w[1], w[2], w[3]);
}

// GUARD: the run may reach beyond n_in (the last runs of a capture, launched on their own): every chunk is range-checked.  All other
// runs load without a test.
template <int IN, int M2, bool GUARD>
__global__ void __launch_bounds__(IG_THREADS, IG_THREADS == 256 ? 3 : 4) ingest_stream_kernel(const ig_params P, const int *__restrict__ tp, const double *__restrict__ c2tab)
{
    constexpr int FB = IN == WFX_IN_I16_STEREO ? 4 : 2;                   // bytes per frame
    constexpr int FPC = 16 / FB;                                          // frames per 16-byte chunk
    constexpr int CPR = IG_M / FPC;                                       // chunks per row: 8 (IQ), 4 (mono)
    constexpr int CPT = IG_BLK * CPR / IG_THREADS;                        // chunks per thread and iteration: 16 (IQ), 8 (mono)
    constexpr int HC = IG_HALO * CPR;                                     // chunks of the first halo
    extern __shared__ __align__(16) unsigned char ig_lds[];
    int *xs = (int *)ig_lds;
    unsigned long long *acc = (unsigned long long *)(ig_lds + IG_XS_BYTES);                // [IG_BLK]: the iteration's window sums
    int *ts = (int *)(ig_lds + IG_XS_BYTES + IG_BLK * 8);                                  // [IG_TAB]
    double *ys = (double *)(ig_lds + IG_XS_BYTES + IG_BLK * 8 + IG_TAB * 4);
    double *cs = ys + IG_YRING + IG_YMIRROR;
    const int t = threadIdx.x;
    const unsigned char *in = (const unsigned char *)P.in + (size_t)blockIdx.y * (size_t)P.in_bs;
    double *out = P.out + (size_t)blockIdx.y * (size_t)P.out_bs;

    const unsigned long long clk0 = WFX_LAB_FLAGS(P.dbg_clk != nullptr) ? (unsigned long long)clock64() : 0ull, wall0 = WFX_LAB_FLAGS(P.dbg_clk != nullptr) ? (unsigned long long)wall_clock64() : 0ull;
    // run = blockIdx.x: workgroups go round-robin to the 8 XCDs, so the ~768 resident ones stream through 768 neighbouring runs, every
    // XCD through every eighth one.  (Dealing each XCD ONE eighth of the capture -- its 96 workgroups on 96 neighbouring runs -- measured
    // 9 % slower, 4.00 against 3.68 ms on the 60-minute stream: docs/history/EXPERIMENTS_rounds1-5.md §9.)
    const long long run = blockIdx.x;
    const long long o0 = P.out0 + run * P.run_out;                          // first output of this run
    const long long ocnt = P.n_out - o0 < P.run_out ? P.n_out - o0 : P.run_out;
    const long long s0 = M2 ? o0 * M2 : o0;                               // first stage-1 output the run needs
    const long long cnt1 = M2 ? (ocnt - 1) * M2 + P.ntaps2 : ocnt;
    const int niter = (int)((cnt1 + IG_BLK - 1) / IG_BLK);
    const long long f0 = s0 * IG_M;                                       // its first frame

    // chunk c of a block of rows goes to row c / CPR; inside the row a group of 16 samples is 8 dwords, high bytes then low bytes.  A
    // thread's chunks of one iteration are IG_THREADS / CPR rows (an even number) apart: a constant LDS offset
    auto chunk_off = [](int cc) { return IN == WFX_IN_I16_STEREO ? 8 * (cc >> 2) + (cc & 3) : 8 * (cc >> 1) + 2 * (cc & 1); };
    auto put = [&](int *dst, const uint4 &v) {
        if (IN == WFX_IN_I16_STEREO) {
            unsigned hi, lo;
            ig_planes(ig_merge2(v.x, v.y), ig_merge2(v.z, v.w), hi, lo);
            dst[0] = (int)hi;
            dst[4] = (int)lo;
        } else {
            unsigned h0, l0, h1, l1;
            ig_planes(v.x, v.y, h0, l0);
            ig_planes(v.z, v.w, h1, l1);
            *(uint2 *)dst = make_uint2(h0, h1);
            *(uint2 *)(dst + 4) = make_uint2(l0, l1);
        }
    };
    constexpr int RPU = IG_THREADS / CPR;
    static_assert(RPU % 2 == 0 && IG_HALO % 2 == 0, "row padding is per pair of rows");
    int *const xput = xs + ig_row_off(IG_HALO + t / CPR) + chunk_off(t % CPR);
    uint4 v[CPT];
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void *)(in + f0 * FB), 0, 0x7fffffff, 0x00020000);      // (gfx9 raw buffer, 32-bit data)
    auto load_block = [&](int n) {
        const long long fb = f0 + ((long long)IG_HALO + (long long)n * IG_BLK) * IG_M;
        if (!GUARD) {
            // straight-line loads, all in flight at once.  Buffer loads: the run's base in a resource descriptor, the lane's 16 bytes as
            // the vector offset, block and chunk as the SCALAR offset -- no vector arithmetic for addresses (the flat form spent two
            // 64-bit adds per load).  (No range checking is relied on: the host sends runs that may overrun to the checked form.)
            typedef unsigned ig_v4u __attribute__((ext_vector_type(4)));
            const unsigned boff = (unsigned)((fb - f0) * FB);
#pragma unroll
            for (int u = 0; u < CPT; ++u) {
                const ig_v4u a = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (unsigned)t * 16u, (int)(boff + (unsigned)u * IG_THREADS * 16u), IG_LOAD_AUX);
                v[u] = make_uint4(a.x, a.y, a.z, a.w);
            }
        } else {
#pragma unroll
            for (int u = 0; u < CPT; ++u) v[u] = ig_fetch_guarded<FB>(in, fb + (long long)(u * IG_THREADS + t) * FPC, P.n_in);
        }
    };

    uint4 h0 = make_uint4(0u, 0u, 0u, 0u);
    if (t < HC) h0 = ig_fetch_guarded<FB>(in, f0 + (long long)t * FPC, P.n_in);
    load_block(0);
    int carry = 0;
    double pend = 0.0;                                                    // a stage-2 output on its way to memory, and its run-local index
    int pend_k = -1;
    long long kdone = 0;                                                  // stage-2 outputs of this run already written

    // the taps' digit planes, the A operand as the 64 lanes hold it: parked in LDS between the iterations' stage-1 phases (four registers
    // that stage 2 needs)
    if (t < 64) ((ig_v4i *)ts)[t] = ((const ig_v4i *)tp)[t];

    auto stage1 = [&](int n) {
        const int lane = t & 63, j = lane & 15, g = lane >> 4;
        const ig_v4i tapA = ((const ig_v4i *)ts)[lane];
        // lane (j, g) of a tile reads the 16 samples at 16 g of the two-row window that starts at row R0 + j: row R0 + j + (g >> 1), its
        // first or second half (8 dwords: four of high bytes, four of low bytes); tiles are 16 rows = 288 dwords apart
        const int *base = xs + ig_row_off(j + (g >> 1)) + 8 * (g & 1);
        const ig_v4i zero = {0, 0, 0, 0};
#pragma unroll 1
        for (int tile = t >> 6; tile < IG_TILES; tile += IG_THREADS / 64) {
            const int *p = base + ig_row_off(16) * tile;
            const ig_v4i sh = *(const ig_v4i *)p, sl = *(const ig_v4i *)(p + 4);
            const ig_v4i dh = __builtin_amdgcn_mfma_i32_16x16x64_i8(tapA, sh, zero, 0, 0, 0);
            const ig_v4i dl = __builtin_amdgcn_mfma_i32_16x16x64_i8(tapA, sl, zero, 0, 0, 0);
            // accumulator register q of lane (g, j) = row 4 g + q of the product, column j: digit plane q of tap chunk g against the
            // window at row 16 tile + j, a share of output 16 tile + j - 2 g
            long long part = 0;
#pragma unroll
            for (int q = 0; q < 4; ++q) part += (long long)(dh[q] * 256 + dl[q]) << (8 * q);
            const int o = 16 * tile + j - 2 * g;
            if (o >= 0 && o < IG_BLK) atomicAdd(acc + o, (unsigned long long)part);
        }
    };
    // block n's window sums -> y1 (float64), into stage 2's ring or, without a stage 2, to memory; the cells are cleared for the next block
    auto convert = [&](int n) {
        const ulonglong2 a = *(const ulonglong2 *)(acc + 2 * t);
        *(ulonglong2 *)(acc + 2 * t) = make_ulonglong2(0ull, 0ull);
        const long long tA = (long long)a.x + P.bias, tB = (long long)a.y + P.bias;
        const double yA = (double)tA * P.sc, yB = (double)tB * P.sc;     // |t| < 2^51 (checked by the host), sc a power of two: exact
        const long long i1 = (long long)n * IG_BLK + 2 * t;              // run-local index of output A
        if (M2 == 0) {
            double *o = out + o0 + i1;
            if (i1 + 1 < ocnt && ((uintptr_t)o & 15u) == 0) {
                *(double2 *)o = make_double2(yA, yB);
            } else {
                if (i1 < ocnt) o[0] = yA;
                if (i1 + 1 < ocnt) o[1] = yB;
            }
        } else {
            const int s = (int)(i1 & (IG_YRING - 1));
            *(double2 *)(ys + s) = make_double2(yA, yB);
            if (s < IG_YMIRROR) *(double2 *)(ys + IG_YRING + s) = make_double2(yA, yB);
        }
    };
    // Stage 2's outputs leave between the stash and the next block's requests, one iteration after they were computed, as non-temporal
    // stores.  They are what the kernel pays for beyond its loads: 0.46 GB written beside 22 GB read cost 0.25-0.4 ms of 3.45-3.55
    // (the same stores into 2 KiB per run: 0.05) -- small write bursts that turn the channels around under the read streams; where
    // they are issued inside the iteration, how many iterations' worth are issued together (1, 4, 8) and whether only whole 128-byte
    // lines are written made no difference, the cache policy did (docs/history/EXPERIMENTS_rounds1-5.md 9.1).
    auto flush = [&]() {
        if (M2 == 0) return;
        if (pend_k >= 0 && !WFX_LAB_FLAGS(P.dbg_flags & 16)) {                          // (16: nothing is stored)
            typedef unsigned ig_v2u __attribute__((ext_vector_type(2)));
            const __amdgpu_buffer_rsrc_t orsrc = __builtin_amdgcn_make_buffer_rsrc((void *)(out + o0), 0, 0x7fffffff, 0x00020000);
            __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(ig_v2u, pend), orsrc, (unsigned)pend_k * 8u, 0, IG_STORE_AUX);
        }
        pend_k = -1;
    };
    // stage 2 on everything iteration n made complete: outputs k with M2 k + ntaps2 <= (n + 1) * IG_BLK.  The tile kernel's canonical
    // order -- one FMA chain per polyphase row r = i mod M2 in ascending tap order, then row 0 + row 1 (+ row 2) -- is walked tap by tap:
    // consecutive taps, consecutive ring entries, M2 independent chains; GRP taps are fetched (scalar loads, LDS reads) before
    // their FMAs so that one wait serves GRP of them
    auto stage2 = [&](int n) {
        if (M2 == 0) return;
        constexpr int MM = M2 ? M2 : 1;
        constexpr int GRP = 6;
        static_assert(GRP % MM == 0 && IG_C2PAD % (2 * GRP) == 0, "a group of taps covers whole rounds of the chains");
        const long long avail = (long long)(n + 1) * IG_BLK;
        long long kend = avail >= P.ntaps2 ? (avail - P.ntaps2) / MM + 1 : 0;
        if (kend > ocnt) kend = ocnt;
        const int ngrp2 = (P.ntaps2 + 2 * GRP - 1) / (2 * GRP);          // trips of two groups (the table is zero padded)
        for (long long k = kdone + t; k < kend; k += IG_THREADS) {
            const double *y = ys + (int)((k * MM) & (IG_YRING - 1));     // the window is contiguous: slots 0..127 are mirrored behind the ring
            double acc[MM];
#pragma unroll
            for (int r = 0; r < MM; ++r) acc[r] = 0.0;
            // taps (broadcast reads of the copy in LDS) and ring entries of the NEXT group are requested before this group's FMAs;
            // the two register sets swap roles by name
            double ca[GRP], wa[GRP], cb[GRP], wb[GRP];
            auto fetch = [&](double (&c)[GRP], double (&w)[GRP], int i) {
                if (GRP % 2 == 0) {
#pragma unroll
                    for (int q = 0; q + 1 < GRP; q += 2) {
                        const double2 cc = *(const double2 *)(cs + i + q);
                        c[q] = cc.x;
                        c[q + 1] = cc.y;
                    }
                } else {
#pragma unroll
                    for (int q = 0; q < GRP; ++q) c[q] = cs[i + q];
                }
#pragma unroll
                for (int q = 0; q < GRP; ++q) w[q] = y[i + q];
            };
            auto mac = [&](const double (&c)[GRP], const double (&w)[GRP]) {
#pragma unroll
                for (int q = 0; q < GRP; ++q) acc[q % MM] = fma(c[q], w[q], acc[q % MM]);
            };
            fetch(ca, wa, 0);
#pragma unroll 1
            for (int g = 0; g < ngrp2; ++g) {
                fetch(cb, wb, 2 * GRP * g + GRP);
                mac(ca, wa);
                if (g + 1 < ngrp2) fetch(ca, wa, 2 * GRP * (g + 1));
                mac(cb, wb);
            }
            double tot = 0.0;
#pragma unroll
            for (int r = 0; r < MM; ++r) tot += acc[r];
            // the output waits in a register for `flush`
            if (pend_k >= 0) flush();                                    // (a second output of this thread in one call: never with the run lengths the host picks)
            pend = tot;
            pend_k = (int)k;
        }
        kdone = kend;
    };

    if (M2) {
        // stage-2 taps into LDS, zero padded (a padded tap times a ring entry outside the window adds nothing -- the ring starts cleared,
        // so that entry is a finite number)
        for (int i = t; i < IG_C2PAD; i += IG_THREADS) cs[i] = i < P.ntaps2 ? c2tab[i] : 0.0;
        for (int i = t; i < IG_YRING + IG_YMIRROR; i += IG_THREADS) ys[i] = 0.0;
    }
    *(ulonglong2 *)(acc + 2 * t) = make_ulonglong2(0ull, 0ull);
    int *const cw = xs + ig_row_off(t / IG_RD) + (t % IG_RD);      // the carried rows: dword t of rows 0..7 <- dword t of rows IG_BLK ..
    if (t < HC) put(xs + ig_row_off(t / CPR) + chunk_off(t % CPR), h0);      // (the first iteration's come from memory)
    for (int n = 0; n < niter; ++n) {
        __syncthreads();                       // A: stage 1 of iteration n - 1 is done with the rows and its sums are complete; stage 2 is done with the ring
        if (n > 0 && t < IG_HALO * IG_RD) cw[0] = carry;
        if (!WFX_LAB_FLAGS(P.dbg_flags & 1)) {
#pragma unroll
            for (int u = 0; u < CPT; ++u) put(xput + u * ig_row_off(RPU), v[u]);
        } else {      // (keeps the loads alive)
            unsigned accx = 0;
#pragma unroll
            for (int u = 0; u < CPT; ++u) accx ^= v[u].x ^ v[u].y ^ v[u].z ^ v[u].w;
            if (accx == 0x9e3779b9u) xput[0] = (int)accx;
        }
        // (the fence keeps the scheduler from hoisting these loads above the stash: it would rename the registers and then wait for
        // BOTH blocks before the stash -- vmcnt counts in order --, i.e. expose the whole latency every iteration)
        __builtin_amdgcn_sched_barrier(0);
        flush();
        if (M2 == 0 && n > 0) convert(n - 1);   // (without a stage 2 the sums themselves go to memory: before the loads, for the same reason)
        __builtin_amdgcn_sched_barrier(0);
        if (n + 1 < niter) load_block(n + 1);   // in flight during the whole compute phase
        __builtin_amdgcn_sched_barrier(0);
        if (M2 != 0 && n > 0) convert(n - 1);
        if (!WFX_LAB_FLAGS(P.dbg_flags & 8)) __syncthreads();                       // B: the rows are in place, block n - 1 is in the ring
        if (!WFX_LAB_FLAGS(P.dbg_flags & 4)) stage1(n);
        if (n > 0 && !WFX_LAB_FLAGS(P.dbg_flags & 2)) stage2(n - 1);
        if (t < IG_HALO * IG_RD) carry = cw[ig_row_off(IG_BLK)];
    }
    __syncthreads();
    convert(niter - 1);
    if (M2 && !WFX_LAB_FLAGS(P.dbg_flags & 2)) {
        __syncthreads();
        stage2(niter - 1);
    }
    flush();
    if (WFX_LAB_FLAGS(P.dbg_clk != nullptr) && t == 0 && (blockIdx.x & 63) == 0 && blockIdx.y == 0 && (blockIdx.x >> 6) < 2048) {
        P.dbg_clk[2 * (blockIdx.x >> 6)] = (unsigned long long)clock64() - clk0;
        P.dbg_clk[2 * (blockIdx.x >> 6) + 1] = (unsigned long long)wall_clock64() - wall0;
    }
}

// plain read of a buffer in the ingest's own access shape (16-byte loads, 16 in flight per lane, contiguous 64 KiB blocks dealt round-robin):
// what this GPU delivers to a kernel that does nothing with the bytes -- the ceiling the ingest's rate is put beside (bench.py)
__global__ void __launch_bounds__(256) read_rate_kernel(const uint4 *__restrict__ v, unsigned long long nblocks, unsigned *__restrict__ sink)
{
    const int t = threadIdx.x;
    unsigned acc = 0;
    for (unsigned long long b = blockIdx.x; b < nblocks; b += gridDim.x) {
        const uint4 *p = v + b * 4096ull + t;
        uint4 a[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) a[u] = p[u * 256];
#pragma unroll
        for (int u = 0; u < 16; ++u) acc += a[u].x ^ a[u].y ^ a[u].z ^ a[u].w;
    }
    if (acc == 0x12345678u) sink[0] = acc;
}

}   // namespace

int wfx_dev_read_rate(wfx_ctx *ctx, const void *dev, uint64_t bytes, int reps, double *gbs)
{
    if (((uintptr_t)dev & 15u) || bytes < (1ull << 20) || reps < 1) return wfx_fail(ctx, WFX_ERR_BAD_ARG, "read rate: a 16-byte aligned buffer of at least 1 MiB");
    const unsigned long long nblocks = bytes >> 16;
    WFX_TRY(wfx_reserve(ctx, ctx->b_tmp, 64));
    hipEvent_t e0 = nullptr, e1 = nullptr;
    WFX_HIP(ctx, hipEventCreate(&e0));
    WFX_HIP(ctx, hipEventCreate(&e1));
    const unsigned grid = (unsigned)std::min<unsigned long long>(nblocks, 4096ull);
    hipLaunchKernelGGL(read_rate_kernel, dim3(grid), dim3(256), 0, ctx->stream, (const uint4 *)dev, nblocks, (unsigned *)ctx->b_tmp.p);
    double best = 0.0;
    for (int r = 0; r < reps; ++r) {
        (void)hipEventRecord(e0, ctx->stream);
        hipLaunchKernelGGL(read_rate_kernel, dim3(grid), dim3(256), 0, ctx->stream, (const uint4 *)dev, nblocks, (unsigned *)ctx->b_tmp.p);
        (void)hipEventRecord(e1, ctx->stream);
        hipError_t e = hipEventSynchronize(e1);
        float ms = 0.0f;
        if (e == hipSuccess) e = hipEventElapsedTime(&ms, e0, e1);
        if (e != hipSuccess) {
            (void)hipEventDestroy(e0);
            (void)hipEventDestroy(e1);
            return wfx_fail_hip(ctx, e, "read rate");
        }
        const double g = (double)(nblocks << 16) / (ms * 1e-3) / 1e9;
        if (g > best) best = g;
    }
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    if (gbs) *gbs = best;
    return 0;
}

// y1[i] = sum_j coef1[j] x[32 i + j] on the grid 2^-fix_shift (exact), and -- factor2 > 0 -- out[k] = sum_j coef2[j] y1[factor2 k + j]
// in float64 behind it; factor2 == 0: out = y1.  x = the int16 frames behind `in` (16-byte aligned), zero beyond n_in.
// *handled = 0 (and nothing enqueued) when the shapes are not the ones this kernel is built for: the caller runs the tile kernels.
int wfx_dev_ingest_stream(wfx_ctx *ctx, const void *in, int in_kind, uint64_t n_in, int factor, const double *coef1, int ntaps1, int fix_shift,
                          int factor2, const double *coef2, int ntaps2, double *out, uint64_t n_out, int nbatch, uint64_t in_stride,
                          uint64_t out_stride, int *handled)
{
    *handled = 0;
    if (getenv("WFX_INGEST_TILE")) return 0;                               // A/B switch: the tile kernels of rounds 1-4
    if (factor != IG_M || (in_kind != WFX_IN_I16_MONO && in_kind != WFX_IN_I16_STEREO)) return 0;
    if (ntaps1 < 1 || ntaps1 > IG_KC * 64 || fix_shift < 8 || fix_shift > 40) return 0;
    if (factor2 != 0 && factor2 != 2 && factor2 != 3) return 0;
    if (nbatch < 1 || nbatch > 65535 || n_out == 0) return 0;
    const int fb = in_kind == WFX_IN_I16_STEREO ? 4 : 2;
    if (((uintptr_t)in & 15u) || (nbatch > 1 && ((in_stride * fb) & 15u))) return 0;
    int nper = 0;
    if (factor2) {
        if (ntaps2 < factor2 || ntaps2 > IG_C2PAD) return 0;
        nper = 4 * (((ntaps2 + factor2 - 1) / factor2 + 3) / 4);          // the tile kernel's row length (canonical order)
    }
    // fixed-point taps as four balanced signed bytes v = q0 + 2^8 q1 + 2^16 q2 + 2^24 q3; the table is the MFMA's A operand as the lanes
    // hold it: lane (row i = lane & 15, group g = lane >> 4), byte b of its 16 = digit plane i & 3 of tap 64 (i >> 2) + 16 g + b
    std::vector<int32_t> fix(IG_KC * 64, 0);
    long long tsum = 0, tabs = 0;
    for (int j = 0; j < ntaps1; ++j) {
        const double v = nearbyint(ldexp(coef1[j], fix_shift));
        if (!(fabs(v) < (double)(1 << 30))) return 0;
        fix[j] = (int32_t)v;
        tsum += fix[j];
        tabs += llabs((long long)fix[j]);
    }
    if (tabs >= (1ll << 35)) return 0;                                     // |sum| <= 32768 sum |tap| < 2^50: the float64 result is exact
    std::vector<int32_t> tab(IG_TAB, 0);
    for (int j = 0; j < IG_KC * 64; ++j) {
        int32_t v = fix[j];
        for (int q = 0; q < 4; ++q) {
            const int32_t d = ((v + 128) & 255) - 128;
            v = (v - d) >> 8;
            const int c = j / 64, g = (j % 64) / 16, bb = j % 16;
            uint32_t &w = (uint32_t &)tab[(size_t)((16 * g + 4 * c + q) * 4 + bb / 4)];
            w |= (uint32_t)(uint8_t)(int8_t)d << (8 * (bb % 4));
        }
        if (v != 0) return 0;
    }
    const int *dtab = (const int *)wfx_coef_device(ctx, (const float *)tab.data(), tab.size());
    if (!dtab) return WFX_ERR_HIP;
    const double *dc2 = nullptr;
    if (factor2) {
        std::vector<double> c2((size_t)ntaps2 + 16, 0.0);
        for (int j = 0; j < ntaps2; ++j) c2[(size_t)j] = coef2[j];
        dc2 = (const double *)wfx_coef_device(ctx, (const float *)c2.data(), c2.size() * 2);
        if (!dc2) return WFX_ERR_HIP;
    }
    // run length: `ni` iterations of 512 stage-1 outputs per workgroup.  Long runs amortise the halo (8 rows of frames, and with
    // stage 2 the ntaps2 - factor2 stage-1 outputs two neighbouring runs both compute); short captures still fill the GPU
    const long long n1_total = factor2 ? ((long long)n_out - 1) * factor2 + ntaps2 : (long long)n_out;
    long long ni = 16 * 512 / IG_BLK;      // 8192 stage-1 outputs = 1 MiB of IQ frames per run
    if (const char *e = getenv("WFX_INGEST_NI")) ni = std::max(1, atoi(e));
    while (ni > 1 && n1_total * nbatch / (ni * IG_BLK) < 2048) ni >>= 1;
    long long run_out = ni * IG_BLK;
    if (factor2) {
        run_out = (ni * IG_BLK - (ntaps2 - factor2)) / factor2;
        // runs start on the 1 KiB grid of the input (a wave's 64 x 16 bytes then never straddle one more 128-byte line than they must):
        // an IQ run of run_out outputs starts run_out * factor2 * 128 bytes behind the previous one
        if (run_out >= 64) run_out &= ~(long long)31;
        if (run_out < 1) return 0;
    }
    const long long runs = ((long long)n_out + run_out - 1) / run_out;
    if (runs > 0x7fffffffll) return 0;
    ig_params P;
    P.in = in;
    P.n_in = (long long)n_in;
    P.sc = ldexp(in_kind == WFX_IN_I16_STEREO ? 0.5 : 1.0, -fix_shift);
    P.nper = nper;
    P.ntaps2 = ntaps2;
    P.out = out;
    P.n_out = (long long)n_out;
    P.run_out = run_out;
    P.in_bs = nbatch > 1 ? (long long)in_stride * fb : 0;
    P.out_bs = nbatch > 1 ? (long long)out_stride : 0;
    P.out0 = 0;
    P.bias = 128 * tsum;
    P.dbg_flags = 0;
    if (const char *e = WFX_LAB_ENV("WFX_INGEST_DBG")) P.dbg_flags = atoi(e);
    P.dbg_clk = nullptr;
    static unsigned long long *clk_buf = nullptr;
    if (WFX_LAB_ENV("WFX_INGEST_CLK")) {
        if (!clk_buf) (void)hipMalloc((void **)&clk_buf, 2 * 2048 * sizeof(unsigned long long));
        if (clk_buf) (void)hipMemsetAsync(clk_buf, 0, 2 * 2048 * sizeof(unsigned long long), ctx->stream);
        P.dbg_clk = clk_buf;
    }
    // runs whose last block of frames ends inside the capture load without range checks; the few behind them are launched on their own
    // with the checked form of the kernel
    const long long blk_frames = (long long)IG_BLK * IG_M;
    const long long run_y1 = factor2 ? run_out * factor2 : run_out;       // stage-1 outputs between the starts of two runs
    const long long run_iters = factor2 ? (run_y1 - factor2 + ntaps2 + IG_BLK - 1) / IG_BLK : ni;
    const long long run_frames = (IG_HALO + run_iters * IG_BLK) * (long long)IG_M;
    long long full = 0;                                                    // run r is unchecked iff r * run_y1 * 32 + run_frames <= n_in
    if ((long long)n_in >= run_frames) full = std::min(runs, ((long long)n_in - run_frames) / (run_y1 * IG_M) + 1);
    (void)blk_frames;
    typedef void (*kern_t)(const ig_params, const int *, const double *);
    auto pick = [&](bool guard) -> kern_t {
        if (in_kind == WFX_IN_I16_STEREO) {
            if (guard) return factor2 == 0 ? ingest_stream_kernel<WFX_IN_I16_STEREO, 0, true> : factor2 == 2 ? ingest_stream_kernel<WFX_IN_I16_STEREO, 2, true> : ingest_stream_kernel<WFX_IN_I16_STEREO, 3, true>;
            return factor2 == 0 ? ingest_stream_kernel<WFX_IN_I16_STEREO, 0, false> : factor2 == 2 ? ingest_stream_kernel<WFX_IN_I16_STEREO, 2, false> : ingest_stream_kernel<WFX_IN_I16_STEREO, 3, false>;
        }
        if (guard) return factor2 == 0 ? ingest_stream_kernel<WFX_IN_I16_MONO, 0, true> : factor2 == 2 ? ingest_stream_kernel<WFX_IN_I16_MONO, 2, true> : ingest_stream_kernel<WFX_IN_I16_MONO, 3, true>;
        return factor2 == 0 ? ingest_stream_kernel<WFX_IN_I16_MONO, 0, false> : factor2 == 2 ? ingest_stream_kernel<WFX_IN_I16_MONO, 2, false> : ingest_stream_kernel<WFX_IN_I16_MONO, 3, false>;
    };
    size_t lds = (size_t)IG_XS_BYTES + IG_BLK * 8 + IG_TAB * 4 + (factor2 ? (size_t)IG_YS_BYTES : 0);
    if (const char *e = WFX_LAB_ENV("WFX_INGEST_DBG_LDS")) {
        lds += (size_t)atoi(e);
        (void)hipFuncSetAttribute((const void *)pick(false), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        (void)hipFuncSetAttribute((const void *)pick(true), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    }
    wfx_prof_begin(ctx, K_POLYPHASE_IN);                                  // (one record: the checked tail is part of the same pass)
    if (full > 0) hipLaunchKernelGGL(pick(false), dim3((unsigned)full, (unsigned)nbatch), dim3(IG_THREADS), lds, ctx->stream, P, dtab, dc2);
    if (full < runs) {
        // what is left (less than two runs) goes to the range-checked form in runs of ONE iteration each: side by side on as many CUs
        // instead of one workgroup walking a whole run alone while the GPU waits for it (that cost 5 % of the pass)
        P.out0 = full * run_out;
        P.run_out = factor2 ? (IG_BLK - (ntaps2 - factor2)) / factor2 : IG_BLK;
        if (factor2 && P.run_out >= 64) P.run_out &= ~(long long)31;
        if (P.run_out < 1) P.run_out = run_out;
        const long long tail_runs = ((long long)n_out - P.out0 + P.run_out - 1) / P.run_out;
        hipLaunchKernelGGL(pick(true), dim3((unsigned)tail_runs, (unsigned)nbatch), dim3(IG_THREADS), lds, ctx->stream, P, dtab, dc2);
    }
    wfx_prof_end(ctx);
    if (P.dbg_clk && full > 0) {
        std::vector<unsigned long long> hc(2 * 2048);
        (void)hipStreamSynchronize(ctx->stream);
        (void)hipMemcpy(hc.data(), clk_buf, hc.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost);
        double sc = 0.0, sw = 0.0, lo = 1e30, hi = 0.0;
        int cnt = 0;
        for (int i = 0; i < 2048; ++i)
            if (hc[2 * i + 1]) {
                const double mhz = (double)hc[2 * i] / ((double)hc[2 * i + 1] / 100.0);
                sc += (double)hc[2 * i];
                sw += (double)hc[2 * i + 1];
                lo = std::min(lo, mhz);
                hi = std::max(hi, mhz);
                ++cnt;
            }
        if (cnt) fprintf(stderr, "[wfx ingest] shader clock over %d sampled runs: mean %.0f MHz (min %.0f, max %.0f); mean run %.1f us\n", cnt, sc / (sw / 100.0), lo, hi, sw / 100.0 / cnt);
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return wfx_fail_hip(ctx, e, "launch ingest_stream_kernel");
    *handled = 1;
    return 0;
}

// GB/s (input bytes per second) at which the streaming ingest itself -- the fused / 32 -> / 3 kernel, outputs into a scratch allocation --
// works through `bytes` of device memory taken as an IQ capture (`out` != nullptr: outputs there, bytes / 48 of them).  Its time does not depend on the data, but it does depend on WHERE the
// allocation lies: ~770 workgroups each walking a megabyte of their own, with small write bursts in between, ran 3.43-3.56 ms on some
// 22 GB allocations of a process and 3.9-4.1 ms on others, the same ones in every process of the box, while the dense sweep of
// wfx_dev_read_rate showed 6.2-6.4 TB/s on all of them (docs/history/EXPERIMENTS_rounds1-5.md 9.2).  Callers that keep a capture buffer for long can time a few
// allocations with this and keep the best (wefax_amd/_native.py: Context.dev_malloc_placed).
int wfx_dev_stream_rate(wfx_ctx *ctx, const void *dev, uint64_t bytes, double *out, int reps, double *gbs)
{
    if (((uintptr_t)dev & 15u) || bytes < (64ull << 20) || reps < 1) return wfx_fail(ctx, WFX_ERR_BAD_ARG, "stream rate: a 16-byte aligned buffer of at least 64 MiB");
    const uint64_t frames = bytes / 4;
    const int ntaps1 = 253, ntaps2 = 119;
    const uint64_t n1 = frames / IG_M - IG_HALO, n2 = (n1 - ntaps2) / 3 + 1;
    std::vector<double> c1(ntaps1, ldexp(1.0, -20)), c2(ntaps2, 1.0 / ntaps2);
    double *scratch = nullptr;                                             // (`out`: bytes / 48 are written there instead)
    if (!out) WFX_HIP(ctx, hipMalloc((void **)&scratch, n2 * sizeof(double)));
    hipEvent_t e0 = nullptr, e1 = nullptr;
    hipError_t he = hipEventCreate(&e0);
    if (he == hipSuccess) he = hipEventCreate(&e1);
    double best = 0.0;
    int rc = he == hipSuccess ? 0 : wfx_fail_hip(ctx, he, "stream rate");
    for (int r = -2; r < reps && rc == 0; ++r) {                           // (two launches to bring the clocks up)
        int handled = 0;
        (void)hipEventRecord(e0, ctx->stream);
        rc = wfx_dev_ingest_stream(ctx, dev, WFX_IN_I16_STEREO, frames, IG_M, c1.data(), ntaps1, 30, 3, c2.data(), ntaps2, out ? out : scratch, n2, 1, 0, 0, &handled);
        (void)hipEventRecord(e1, ctx->stream);
        if (rc == 0 && !handled) rc = wfx_fail(ctx, WFX_ERR_BAD_ARG, "stream rate: the ingest kernel declined the probe");
        if (rc) break;
        hipError_t e = hipEventSynchronize(e1);
        float ms = 0.0f;
        if (e == hipSuccess) e = hipEventElapsedTime(&ms, e0, e1);
        if (e != hipSuccess) {
            rc = wfx_fail_hip(ctx, e, "stream rate");
            break;
        }
        if (r >= 0) best = std::max(best, (double)(frames * 4) / (ms * 1e-3) / 1e9);
    }
    (void)hipStreamSynchronize(ctx->stream);
    if (e0) (void)hipEventDestroy(e0);
    if (e1) (void)hipEventDestroy(e1);
    if (scratch) (void)hipFree(scratch);
    if (rc == 0 && gbs) *gbs = best;
    return rc;
}
