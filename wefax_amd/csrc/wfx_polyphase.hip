// Halo-local sample-rate front end for oversampled captures (BASELINE configs[3]: 1.536 MS/s int16 IQ):
// the time-domain counterpart of wefax.py:375-394 (scipy FFT resample, a global operator) down to a hand-over rate above
// 11 025 Hz, built from ONE stencil,
//
//   decimate_fir64     y[i] = sum_j c[j] * x[first + i*M + j]
//
// chained by the host (wefax_amd/polyphase.py designs the filters and keeps the index bookkeeping: 1.536 MS/s --/32--> 48 kHz
// --/3--> 16 kHz; the exact FFT resampler of the decode path takes the last step).  It reads every input byte once (16-byte
// loads), keeps the window in LDS and is HBM-bound.  Two arithmetic forms, neither with a rounding error of its own where that is
// possible: int16 input and a power-of-two factor -- the ingest -- as an INTEGER dot product with taps on a 2^-30 grid (MODE 1);
// everything else float64 taps and sums in one canonical order per output (MODE 2).  The result never depends on how a capture
// is cut into slices.  The stereo / IQ merge of wefax.py:360-373 ((int16)(L+R) wrapped, /2) is fused into the load.  (The fp32
// forms of rounds 1-2 -- MODE 0, the rational x147/160 stage -- were removed in round 4.)
#include <algorithm>

#include "wfx_internal.h"

namespace {

constexpr int PP_THREADS = 256;
constexpr int PP_LDS_BYTES = 64 * 1024;

// the same 4 samples as float64 (MODE 2)
__device__ __forceinline__ void pp_load4d(const short *p, double *w)
{
    const int *q = (const int *)p;
    const int a = q[0], b = q[1];
    w[0] = (double)(short)a;
    w[1] = (double)(a >> 16);
    w[2] = (double)(short)b;
    w[3] = (double)(b >> 16);
}
__device__ __forceinline__ void pp_load4d(const double *p, double *w)
{
    w[0] = p[0]; w[1] = p[1]; w[2] = p[2]; w[3] = p[3];
}
typedef short pp_s2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ int pp_dot2(int w, int c, int acc)      // acc + w.lo * c.lo + w.hi * c.hi, int16 x int16 -> int32, exact
{
    return __builtin_amdgcn_sdot2(__builtin_bit_cast(pp_s2, w), __builtin_bit_cast(pp_s2, c), acc, false);
}

// ---- element access -----------------------------------------------------------
// IN 0: int16 mono, 1: int16 pairs merged with int16 wrap, 2: float64 (between the stages)
template <int IN> struct pp_in;
template <> struct pp_in<WFX_IN_I16_MONO> {
    typedef short store_t;                    // LDS representation
    static constexpr int PER16 = 8;           // elements per 16-byte chunk
    static constexpr int BYTES = 2;
    static constexpr float SCALE = 1.0f;
    __device__ static void chunk(const uint4 &v, store_t *e)
    {
        const unsigned w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            e[2 * k] = (short)(w[k] & 0xffffu);
            e[2 * k + 1] = (short)(w[k] >> 16);
        }
    }
    __device__ static store_t one(const void *p, long long i) { return ((const short *)p)[i]; }
};
template <> struct pp_in<WFX_IN_I16_STEREO> {
    typedef short store_t;
    static constexpr int PER16 = 4;
    static constexpr int BYTES = 4;
    static constexpr float SCALE = 0.5f;      // (L+R)/2: the wrapped sum is kept as int16, the /2 goes into the scale
    __device__ static void chunk(const uint4 &v, store_t *e)
    {
        const unsigned w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int k = 0; k < 4; ++k) e[k] = (short)((w[k] & 0xffffu) + (w[k] >> 16));
    }
    __device__ static store_t one(const void *p, long long i)
    {
        const unsigned w = ((const unsigned *)p)[i];
        return (short)((w & 0xffffu) + (w >> 16));
    }
};
template <> struct pp_in<WFX_IN_F64_MONO> {          // float64 between the stages of the exact chain (MODE 2 only)
    typedef double store_t;
    static constexpr int PER16 = 2;
    static constexpr int BYTES = 8;
    static constexpr float SCALE = 1.0f;
    __device__ static void chunk(const uint4 &v, store_t *e)
    {
        e[0] = __longlong_as_double((long long)(((unsigned long long)v.y << 32) | v.x));
        e[1] = __longlong_as_double((long long)(((unsigned long long)v.w << 32) | v.z));
    }
    __device__ static store_t one(const void *p, long long i) { return ((const double *)p)[i]; }
};

// one 16-byte chunk starting at element e0 of the aligned view; elements outside the caller's array read as zero
template <int IN>
__device__ __forceinline__ uint4 pp_fetch(const void *in, const unsigned char *base, long long e0, int misalign, long long n_in)
{
    typedef pp_in<IN> A;
    if (e0 >= misalign && e0 + A::PER16 <= n_in + misalign) return *(const uint4 *)(base + e0 * A::BYTES);
    const long long ci = e0 - misalign;
    unsigned w[4];
    if (A::BYTES == 8) {
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const bool ok = ci + k >= 0 && ci + k < n_in;
            w[2 * k] = ok ? ((const unsigned *)in)[2 * (ci + k)] : 0u;
            w[2 * k + 1] = ok ? ((const unsigned *)in)[2 * (ci + k) + 1] : 0u;
        }
        return make_uint4(w[0], w[1], w[2], w[3]);
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        if (A::BYTES == 4) {
            w[k] = (ci + k >= 0 && ci + k < n_in) ? ((const unsigned *)in)[ci + k] : 0u;
        } else {
            const long long c0 = ci + 2 * k, c1 = c0 + 1;
            const unsigned lo = (c0 >= 0 && c0 < n_in) ? ((const unsigned short *)in)[c0] : 0u;
            const unsigned hi = (c1 >= 0 && c1 < n_in) ? ((const unsigned short *)in)[c1] : 0u;
            w[k] = lo | (hi << 16);
        }
    }
    return make_uint4(w[0], w[1], w[2], w[3]);
}

// ---- integer decimation ---------------------------------------------------------
// LDS holds the tile's input window in polyphase order: sample w (window-relative) sits at
// row w % M, column w / M.  A thread owns 4 consecutive outputs and a group of rows; for one
// row it slides a register window along the columns: one 4-column LDS read feeds 16 FMAs, and
// the tap c[r + M*jq] is the same for every thread (scalar loads from the polyphase-ordered
// table).  The global loads of a tile are issued in batches of PP_BATCH 16-byte chunks per
// thread before any of them is consumed (bytes in flight, not occupancy, hide the HBM latency).
// When a tile has fewer than 256 output quads the rows are split over `rs` thread groups whose
// partial sums are added in group order through LDS: the order of additions per output is
// fixed by (M, rs) alone, never by where a slice starts.
// (The build-time A/B switches of rounds 1-3 are settled -- DESIGN.md 3.6 has what each measured; these are the winners.)
constexpr int PP_TB_MAX = 1024;        // largest tile (outputs)
constexpr int PP_EXACT_WAVES = 4;      // integer-exact ingest: compiled for this many waves per SIMD (128 VGPRs)
constexpr int PP_F64_WAVES = 1;        // float64 stages: waves per SIMD the kernel is compiled for
constexpr int PP_NB = 9;           // 16-byte chunks in flight per thread (decimate): 36 VGPRs

// MODE 1: EXACT -- int16 samples times taps on a fixed-point grid
// 2^-shift, split into a high and a low int16 part, summed with v_dot2_i32_i16 (two multiply-adds per instruction, the
// window pairs straight from LDS, no conversions; the shift travels in bits 8.. of `flush_rows`); the sum of a tile's products is an integer that fits 64 bits, so the
// float64 output is THE value of the FIR with those taps: no rounding anywhere, nothing depends on the tiling.  MODE 2:
// float64 taps and sums in one canonical order (the stages behind the ingest: a few per cent of the samples).
constexpr int PP_FIX_LB = 12;
// float64 rows in LDS: a lane reads 4 consecutive doubles at a stride of 4 doubles (32 bytes), i.e. every other 16-byte granule
// of a 256-byte bank row -- lanes i and i + 8 would meet in the same banks (8-way conflicts on every window read: the /3 stage
// behind the ingest ran LDS-bound at 1.05 ms for 57.6 M outputs).  Two doubles of padding after every 32 shift alternate
// 256-byte rows by one granule: 16 consecutive lanes then cover all 16 granules of a bank row (4-way for 64 lanes = 1 KB
// through a 256-byte port: the least there is).
template <typename S> __device__ __forceinline__ int pp_swz(int col) { return sizeof(S) == 8 ? col + 2 * (col >> 5) : col; }
static int pp_swz_stride(int cols, int esz) { return esz == 8 ? cols + 2 * (cols / 32 + 1) : cols; }
template <int IN, typename OUT, bool ALIGNED, int Q4T, int MODE>
__global__ void __launch_bounds__(PP_THREADS, MODE == 1 ? PP_EXACT_WAVES : PP_F64_WAVES)
decimate_kernel(const void *__restrict__ in, long long n_in, long long first, int M, int log2m, const float *__restrict__ cp, int q4_arg,
                OUT *__restrict__ out, long long n_out, int log2tb, int row_stride, int misalign, int flush_rows, long long in_bs, long long out_bs)
{
    // blockIdx.y: member of a batch of equally shaped jobs (the segments of a rank's columns layout, wfx_shard.hip): its input
    // starts in_bs BYTES (a multiple of 16: the same alignment for every member), its output out_bs elements further on
    in = (const unsigned char *)in + (size_t)blockIdx.y * (size_t)in_bs;
    out += (size_t)blockIdx.y * (size_t)out_bs;
    // log2m < 0: M is not a power of two (/3 at the end of a chain) -- rows are never split over thread groups then and the
    // window position of a chunk is found by a division
    const int q4 = Q4T ? Q4T : q4_arg;        // Q4T > 0: the tap loop has a compile-time trip count and unrolls (short filters)
    typedef pp_in<IN> A;
    typedef typename A::store_t S;
    extern __shared__ __align__(16) unsigned char lds_raw[];
    S *xs = (S *)lds_raw;
    const int t = threadIdx.x, tb = 1 << log2tb;
    const int log2q = log2tb - 2;                               // quads per tile
    const int log2qw = log2q < 6 ? 6 : log2q;                   // a row group is made of whole waves: its index is scalar
    int rs_log2 = 8 - log2qw;                                   // row groups: 256 threads / quads
    if (rs_log2 > log2m) rs_log2 = log2m;
    if (rs_log2 < 0) rs_log2 = 0;                               // (also when M is not a power of two)
    const int g = t & ((1 << log2qw) - 1);
    const int part = __builtin_amdgcn_readfirstlane(t >> 6) >> (log2qw - 6);
    const int rows_per = M >> rs_log2;
    float *psum = (float *)(lds_raw + (((size_t)row_stride * M * sizeof(S) + 15) & ~(size_t)15));     // [rs - 1][quads][4]
    // the taps are read from the table in memory with scalar loads (the row index is uniform) -- as LDS broadcast reads, 16 bytes per
    // lane, they cost the LDS pipe as much as the window reads -- and the next tile's first chunks are requested before the tap loop
    constexpr bool PF = true;
    const long long ntiles = (n_out + tb - 1) >> log2tb;
    // 16-byte aligned view of the input: element index e of the caller's array is element e + misalign of `base`
    const unsigned char *base = (const unsigned char *)in - (size_t)misalign * A::BYTES;
    struct geom {
        long long o0, src0, a0;
        int cnt, win, nchunks, wbase;
    };
    auto geom_of = [&](long long tile) {
        geom G;
        G.o0 = tile << log2tb;
        G.cnt = (int)((n_out - G.o0 < tb) ? (n_out - G.o0) : tb);
        G.src0 = first + G.o0 * M;                                      // caller index of window sample 0
        G.win = (((G.cnt + 3) & ~3) + 4 * q4) * M;                      // samples the register windows touch
        G.a0 = (G.src0 + misalign) & ~(long long)(A::PER16 - 1);        // aligned start, in elements of `base`
        G.nchunks = (int)((G.src0 + misalign + G.win - G.a0 + A::PER16 - 1) / A::PER16);     // <= PP_NB * 256 (host)
        G.wbase = (int)(G.a0 - misalign - G.src0);                      // window index of chunk 0's first element (<= 0)
        return G;
    };
    uint4 v[PP_NB];                                                     // one tile's input, in flight or waiting for its LDS slot
    auto fetch = [&](const geom &G, int cb) {
        // uniform test: the whole window lies inside the caller's array -> straight-line vector loads, all in flight at once
        if (G.a0 >= misalign && G.a0 + (long long)G.nchunks * A::PER16 <= n_in + misalign) {
#pragma unroll
            for (int u = 0; u < PP_NB; ++u)
                if (cb + u * PP_THREADS < G.nchunks) {
                    int c = cb + u * PP_THREADS + t;
                    c = c < G.nchunks ? c : G.nchunks - 1;
                    v[u] = *(const uint4 *)(base + (G.a0 + (long long)c * A::PER16) * A::BYTES);
                }
        } else {
#pragma unroll
            for (int u = 0; u < PP_NB; ++u) {
                const int c = cb + u * PP_THREADS + t;
                if (c < G.nchunks) v[u] = pp_fetch<IN>(in, base, G.a0 + (long long)c * A::PER16, misalign, n_in);
            }
        }
    };
    auto stash = [&](const geom &G, int cb) {
#pragma unroll
        for (int u = 0; u < PP_NB; ++u) {
            const int c = cb + u * PP_THREADS + t;
            if (c < G.nchunks) {
                S e[A::PER16];
                A::chunk(v[u], e);
                const int w0 = G.wbase + c * A::PER16;
                int r, col;
                if (log2m >= 0) {
                    r = w0 & (M - 1);
                    col = w0 >> log2m;
                } else {                                        // floor division: w0 may be negative in the first chunk
                    col = (w0 >= 0 ? w0 : w0 - (M - 1)) / M;
                    r = w0 - col * M;
                }
                if (ALIGNED) {      // the host aligned the window to the 16-byte grid and M >= PER16: one column, consecutive rows
                    const int addr = r * row_stride + pp_swz<S>(col);
#pragma unroll
                    for (int k = 0; k < A::PER16; ++k) xs[addr + k * row_stride] = e[k];
                } else {
#pragma unroll
                    for (int k = 0; k < A::PER16; ++k) {
                        if (w0 + k >= 0 && w0 + k < G.win) xs[r * row_stride + pp_swz<S>(col)] = e[k];
                        ++r;
                        if (r == M) {
                            r = 0;
                            ++col;
                        }
                    }
                }
            }
        }
    };
    if (PF && (long long)blockIdx.x < ntiles) fetch(geom_of(blockIdx.x), 0);
    for (long long tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const geom Gn = geom_of(tile);
        const long long o0 = Gn.o0;
        const int cnt = Gn.cnt;
        __syncthreads();                       // the previous tile's compute phase is done with the LDS window
        // (PF: the tile's first batch of chunks was requested before the previous tile's compute phase)
        if (!PF) fetch(Gn, 0);
        stash(Gn, 0);
        for (int cb = PP_NB * PP_THREADS; cb < Gn.nchunks; cb += PP_NB * PP_THREADS) {
            fetch(Gn, cb);
            stash(Gn, cb);
        }
        __syncthreads();
        if (PF && tile + gridDim.x < ntiles) fetch(geom_of(tile + gridDim.x), 0);      // in flight during the compute phase
        const bool active = (4 * g < cnt) && (part < (1 << rs_log2));
        if constexpr (MODE == 1) {
            // exact: hi parts in one int32 per output over the whole tile, lo parts flushed into an int64 every flush_rows rows
            // (the host checked both bounds against the worst-case sample, |s| = 32768)
            long long tl0 = 0, tl1 = 0, tl2 = 0, tl3 = 0;
            if (active) {
                const int r0 = part * rows_per;
                int ah0 = 0, ah1 = 0, ah2 = 0, ah3 = 0, al0 = 0, al1 = 0, al2 = 0, al3 = 0;
                const int *ci = (const int *)cp;
                constexpr int NP = 2 * (Q4T ? Q4T : 1);
                if constexpr (Q4T > 0) {
                    // compile-time tap count: the NEXT row's window pairs (LDS) and taps (scalar loads) are requested before this
                    // row's 16 q4 + 4 dot products -- one row's load latency no longer sits in front of every row's arithmetic
                    constexpr int NW = NP + 2, NC = 4 * NP + 4;
                    int w[NW], wn[NW], c[NC], cn[NC];
                    {
                        const int *row = (const int *)(xs + r0 * row_stride + 4 * g);
                        const int *ct = ci + r0 * NC;
#pragma unroll
                        for (int p = 0; p < NW; ++p) w[p] = row[p];
#pragma unroll
                        for (int p = 0; p < NC; ++p) c[p] = ct[p];
                    }
                    for (int r = r0; r < r0 + rows_per; ++r) {
                        const int rn = r + 1 < r0 + rows_per ? r + 1 : r;                  // (the last row fetches itself again: no branch)
                        const int *rowx = (const int *)(xs + rn * row_stride + 4 * g);
                        const int *ctx_ = ci + rn * NC;
#pragma unroll
                        for (int p = 0; p < NW; ++p) wn[p] = rowx[p];
#pragma unroll
                        for (int p = 0; p < NC; ++p) cn[p] = ctx_[p];
#pragma unroll
                        for (int p = 0; p < NP; ++p) {
                            const int eh = c[p], el = c[NP + p], dh = c[2 * NP + p], dl = c[3 * NP + 2 + p];
                            ah0 = pp_dot2(w[p], eh, ah0); al0 = pp_dot2(w[p], el, al0);
                            ah2 = pp_dot2(w[p + 1], eh, ah2); al2 = pp_dot2(w[p + 1], el, al2);
                            ah1 = pp_dot2(w[p], dh, ah1); al1 = pp_dot2(w[p], dl, al1);
                            ah3 = pp_dot2(w[p + 1], dh, ah3); al3 = pp_dot2(w[p + 1], dl, al3);
                        }
                        {
                            const int dh = c[3 * NP], dl = c[4 * NP + 2];
                            ah1 = pp_dot2(w[NP], dh, ah1); al1 = pp_dot2(w[NP], dl, al1);
                            ah3 = pp_dot2(w[NP + 1], dh, ah3); al3 = pp_dot2(w[NP + 1], dl, al3);
                        }
                        if (((r + 1) & ((flush_rows & 255) - 1)) == 0) {          // both halves leave their int32 here (taps on grids up to 2^-30)
                            tl0 += al0 + ((long long)ah0 << PP_FIX_LB); tl1 += al1 + ((long long)ah1 << PP_FIX_LB);
                            tl2 += al2 + ((long long)ah2 << PP_FIX_LB); tl3 += al3 + ((long long)ah3 << PP_FIX_LB);
                            al0 = al1 = al2 = al3 = 0;
                            ah0 = ah1 = ah2 = ah3 = 0;
                        }
#pragma unroll
                        for (int p = 0; p < NW; ++p) w[p] = wn[p];
#pragma unroll
                        for (int p = 0; p < NC; ++p) c[p] = cn[p];
                    }
                } else
                for (int r = r0; r < r0 + rows_per; ++r) {
                    const int *row = (const int *)(xs + r * row_stride + 4 * g);      // dword p = samples (w[2p], w[2p+1])
                    const int *c = ci + r * (8 * q4 + 4);     // E_hi[2 q4], E_lo[2 q4], D_hi[2 q4 + 2], D_lo[2 q4 + 2]
                    const int np = Q4T ? NP : 2 * q4;
                    const int *ehi = c, *elo = c + np, *dhi = c + 2 * np, *dlo = c + 3 * np + 2;
                    int w0 = row[0];
#pragma unroll(Q4T ? NP : 1)
                    for (int p = 0; p < np; ++p) {
                        const int w1 = row[p + 1];
                        const int eh = ehi[p], el = elo[p], dh = dhi[p], dl = dlo[p];
                        ah0 = pp_dot2(w0, eh, ah0); al0 = pp_dot2(w0, el, al0);      // output 0: sum_p W[p] . (c[2p], c[2p+1])
                        ah2 = pp_dot2(w1, eh, ah2); al2 = pp_dot2(w1, el, al2);      // output 2: the window one pair later
                        ah1 = pp_dot2(w0, dh, ah1); al1 = pp_dot2(w0, dl, al1);      // output 1: sum_p W[p] . (c[2p-1], c[2p])
                        ah3 = pp_dot2(w1, dh, ah3); al3 = pp_dot2(w1, dl, al3);
                        w0 = w1;
                    }
                    {       // the odd outputs' last pair (c[4 q4 - 1], 0)
                        const int w1 = row[np + 1];
                        const int dh = dhi[np], dl = dlo[np];
                        ah1 = pp_dot2(w0, dh, ah1); al1 = pp_dot2(w0, dl, al1);
                        ah3 = pp_dot2(w1, dh, ah3); al3 = pp_dot2(w1, dl, al3);
                    }
                    if (((r + 1) & ((flush_rows & 255) - 1)) == 0) {
                        tl0 += al0 + ((long long)ah0 << PP_FIX_LB); tl1 += al1 + ((long long)ah1 << PP_FIX_LB);
                        tl2 += al2 + ((long long)ah2 << PP_FIX_LB); tl3 += al3 + ((long long)ah3 << PP_FIX_LB);
                        al0 = al1 = al2 = al3 = 0;
                        ah0 = ah1 = ah2 = ah3 = 0;
                    }
                }
                tl0 += al0 + ((long long)ah0 << PP_FIX_LB); tl1 += al1 + ((long long)ah1 << PP_FIX_LB);
                tl2 += al2 + ((long long)ah2 << PP_FIX_LB); tl3 += al3 + ((long long)ah3 << PP_FIX_LB);
            }
            if (rs_log2 > 0) {
                long long *ps = (long long *)psum;
                if (active && part > 0) {
                    long long *q = ps + ((size_t)((part - 1) << log2qw) + g) * 4;
                    q[0] = tl0; q[1] = tl1; q[2] = tl2; q[3] = tl3;
                }
                __syncthreads();
                if (active && part == 0)
                    for (int q = 1; q < (1 << rs_log2); ++q) {
                        const long long *o = ps + ((size_t)((q - 1) << log2qw) + g) * 4;
                        tl0 += o[0]; tl1 += o[1]; tl2 += o[2]; tl3 += o[3];
                    }
            }
            if (active && part == 0) {
                const long long a[4] = {tl0, tl1, tl2, tl3};
                const double sc = (double)A::SCALE / (double)(1ll << (flush_rows >> 8));   // a power of two: exact
                if (4 * g + 3 < cnt && ((o0 + 4 * g) & 1) == 0 && sizeof(OUT) == 8) {
                    double2 *o2 = (double2 *)(out + o0 + 4 * g);
                    o2[0] = make_double2((double)a[0] * sc, (double)a[1] * sc);
                    o2[1] = make_double2((double)a[2] * sc, (double)a[3] * sc);
                } else {
#pragma unroll
                    for (int k = 0; k < 4; ++k)
                        if (4 * g + k < cnt) out[o0 + 4 * g + k] = (OUT)((double)a[k] * sc);
                }
            }
        } else if constexpr (MODE == 2) {
            double tot0 = 0, tot1 = 0, tot2 = 0, tot3 = 0;
            if (active) {
                const int r0 = part * rows_per;
                const double *cd = (const double *)cp;            // taps by scalar loads from the table in memory (the row index is uniform)
                for (int r = r0; r < r0 + rows_per; ++r) {
                    const S *rowb = xs + r * row_stride;
                    const double *c = cd + r * 4 * q4;
                    double acc0 = 0, acc1 = 0, acc2 = 0, acc3 = 0;
                    double wa[4], wb[4], wn[4];
                    pp_load4d(rowb + pp_swz<S>(4 * g), wa);
                    pp_load4d(rowb + pp_swz<S>(4 * g + 4), wb);
                    double c0 = c[0], c1 = c[1], c2 = c[2], c3 = c[3];
                    // one group of 4 taps on the window (A = columns 0..3, B = 4..7); the columns and taps of the NEXT group are
                    // requested first.  The three window arrays rotate by NAME over three consecutive groups (no register moves:
                    // eight v_mov_b64 per group cost as much as half its FMAs).
                    auto group = [&](const double (&A)[4], const double (&B)[4], double (&N)[4], int j) {
                        const double u0 = c0, u1 = c1, u2 = c2, u3 = c3;
                        if (j + 1 < q4) {
                            pp_load4d(rowb + pp_swz<S>(4 * g + 4 * j + 8), N);
                            c0 = c[4 * j + 4]; c1 = c[4 * j + 5]; c2 = c[4 * j + 6]; c3 = c[4 * j + 7];
                        }
                        acc0 = fma(u0, A[0], acc0); acc1 = fma(u0, A[1], acc1); acc2 = fma(u0, A[2], acc2); acc3 = fma(u0, A[3], acc3);
                        acc0 = fma(u1, A[1], acc0); acc1 = fma(u1, A[2], acc1); acc2 = fma(u1, A[3], acc2); acc3 = fma(u1, B[0], acc3);
                        acc0 = fma(u2, A[2], acc0); acc1 = fma(u2, A[3], acc1); acc2 = fma(u2, B[0], acc2); acc3 = fma(u2, B[1], acc3);
                        acc0 = fma(u3, A[3], acc0); acc1 = fma(u3, B[0], acc1); acc2 = fma(u3, B[1], acc2); acc3 = fma(u3, B[2], acc3);
                    };
                    int j = 0;
                    for (; j + 3 <= q4; j += 3) {
                        group(wa, wb, wn, j);
                        group(wb, wn, wa, j + 1);
                        group(wn, wa, wb, j + 2);
                    }
                    if (q4 - j >= 1) group(wa, wb, wn, j);
                    if (q4 - j >= 2) group(wb, wn, wa, j + 1);
                    tot0 += acc0; tot1 += acc1; tot2 += acc2; tot3 += acc3;
                }
            }
            if (rs_log2 > 0) {
                double *ps = (double *)psum;
                if (active && part > 0) {
                    double *q = ps + ((size_t)((part - 1) << log2qw) + g) * 4;
                    q[0] = tot0; q[1] = tot1; q[2] = tot2; q[3] = tot3;
                }
                __syncthreads();
                if (active && part == 0)
                    for (int q = 1; q < (1 << rs_log2); ++q) {
                        const double *o = ps + ((size_t)((q - 1) << log2qw) + g) * 4;
                        tot0 += o[0]; tot1 += o[1]; tot2 += o[2]; tot3 += o[3];
                    }
            }
            if (active && part == 0) {
                const double a[4] = {tot0, tot1, tot2, tot3};
#pragma unroll
                for (int k = 0; k < 4; ++k)
                    if (4 * g + k < cnt) out[o0 + 4 * g + k] = (OUT)(a[k] * (double)A::SCALE);
            }
        }
    }
}

int ilog2_exact(int m)
{
    int l = 0;
    while ((1 << l) < m) ++l;
    return ((1 << l) == m) ? l : -1;
}

// MODE 1 (exact, int16 in, aligned) and MODE 2 (float64 arithmetic, any input kind): float64 out
template <int IN, int MODE>
int launch_decimate64(wfx_ctx *ctx, const void *in, long long n_in, long long first, int M, const void *cp, int q4, double *out, long long n_out,
                      bool aligned, int flush_rows, int nbatch = 1, long long in_bs = 0, long long out_bs = 0)
{
    typedef pp_in<IN> A;
    const int log2m = ilog2_exact(M);
    const int esz = (int)sizeof(typename A::store_t);
    const int pad = esz == 2 ? 6 : 4;
    int tb = log2m < 0 ? 1024 : PP_TB_MAX;
    while (tb > 64 && (size_t)pp_swz_stride(tb + 4 * q4 + pad, esz) * M * esz > (size_t)PP_LDS_BYTES) tb >>= 1;
    if (log2m < 0 && tb < 1024) return wfx_fail(ctx, WFX_ERR_BAD_ARG, "decimate: %d taps x factor %d do not fit the LDS window", 4 * q4 * M, M);
    const int row_stride = pp_swz_stride(tb + 4 * q4 + pad, esz);
    const size_t lds_x = ((size_t)row_stride * M * esz + 15) & ~(size_t)15;
    // partial sums of the row groups, laid out as the kernel does: rs - 1 groups of (quads rounded up to whole waves) x 4
    const int log2tb = ilog2_exact(tb), log2q = log2tb - 2, log2qw = log2q < 6 ? 6 : log2q;
    int rs_log2 = 8 - log2qw;
    if (rs_log2 > log2m) rs_log2 = log2m;
    if (rs_log2 < 0) rs_log2 = 0;
    const size_t lds = lds_x + (size_t)((((1 << rs_log2) - 1) << log2qw) * 4) * 8;
    if (lds_x > (size_t)PP_LDS_BYTES || lds > 150 * 1024) return wfx_fail(ctx, WFX_ERR_BAD_ARG, "decimate: %d taps x factor %d do not fit the LDS window", 4 * q4 * M, M);
    const int misalign = (int)(((uintptr_t)in & 15u) / A::BYTES);
    const long long ntiles = (n_out + tb - 1) / tb;
    unsigned grid = (unsigned)(ntiles < 4096 ? ntiles : 4096);
    if (nbatch > 1 && (long long)grid * nbatch > 8192) grid = (unsigned)std::max<long long>(1, 8192 / nbatch);      // (tiles are walked grid-stride)
    void (*kern)(const void *, long long, long long, int, int, const float *, int, double *, long long, int, int, int, int, long long, long long);
    if (MODE == 1)
        kern = q4 == 2 ? decimate_kernel<IN, double, true, 2, MODE> : q4 == 3 ? decimate_kernel<IN, double, true, 3, MODE> : decimate_kernel<IN, double, true, 0, MODE>;
    else
        kern = aligned ? decimate_kernel<IN, double, true, 0, MODE> : decimate_kernel<IN, double, false, 0, MODE>;
    if (lds > 48 * 1024) WFX_HIP(ctx, hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    wfx_prof_begin(ctx, (IN == WFX_IN_F64_MONO ? K_POLYPHASE : K_POLYPHASE_IN));
    hipLaunchKernelGGL(kern, dim3(grid, nbatch), dim3(PP_THREADS), lds, ctx->stream, in, n_in, first, M, log2m, (const float *)cp, q4, out, n_out,
                       ilog2_exact(tb), row_stride, misalign, flush_rows, in_bs, out_bs);
    wfx_prof_end(ctx);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return wfx_fail_hip(ctx, e, "launch decimate_kernel (float64 out)");
    return 0;
}

}   // namespace

// The same stencil with float64 taps and a float64 result that carries NO rounding of the sum where that is possible:
//   int16 input (mono / IQ) and a power-of-two factor >= the samples per 16 bytes (the ingest of an oversampled capture):
//     taps rounded to the grid 2^-fix_shift, split hi * 2^12 + lo, integer dot products (MODE 1).  The output IS
//     sum_j round(c[j] 2^shift) x[..] / 2^shift -- it depends on neither tiling nor slicing; *exact_out = 1.
//   anything else (float64 between stages, other factors): float64 FMAs in one canonical order per output (MODE 2).
// fix_shift 0, or taps the exact form cannot take at that shift (a tap >= 2^23 grid steps, sums that could overflow 32 bits
// for the worst-case input): MODE 2.
int wfx_dev_decimate_fir64(wfx_ctx *ctx, const void *in, int in_kind, uint64_t n_in, int64_t first, int M, const double *coef, int ntaps,
                           double *out, uint64_t n_out, int fix_shift, int *exact_out, int nbatch, uint64_t in_stride, uint64_t out_stride)
{
    if (exact_out) *exact_out = 0;
    if (nbatch < 1 || nbatch > 65535) return wfx_fail(ctx, WFX_ERR_BAD_ARG, "decimate: batch of %d", nbatch);
    if (M < 1 || M > 64) return wfx_fail(ctx, WFX_ERR_BAD_ARG, "decimate: factor %d is not in 1..64", M);
    if (ntaps < 1 || ntaps > 4096) return wfx_fail(ctx, WFX_ERR_BAD_ARG, "decimate: %d taps", ntaps);
    if (in_kind != WFX_IN_I16_MONO && in_kind != WFX_IN_I16_STEREO && in_kind != WFX_IN_F64_MONO)
        return wfx_fail(ctx, WFX_ERR_BAD_ARG, "decimate (float64): input kind %d", in_kind);
    if (n_out == 0) return 0;
    const int per16 = in_kind == WFX_IN_I16_MONO ? 8 : (in_kind == WFX_IN_I16_STEREO ? 4 : 2);
    const int ebytes = in_kind == WFX_IN_I16_MONO ? 2 : (in_kind == WFX_IN_I16_STEREO ? 4 : 8);
    if ((uintptr_t)in % ebytes) return wfx_fail(ctx, WFX_ERR_BAD_ARG, "decimate: misaligned input pointer");
    const int misalign = (int)(((uintptr_t)in & 15u) / ebytes);
    const long long in_bs = nbatch > 1 ? (long long)in_stride * ebytes : 0, out_bs = nbatch > 1 ? (long long)out_stride : 0;
    if (in_bs % 16) return wfx_fail(ctx, WFX_ERR_BAD_ARG, "decimate: the members of a batch must start a multiple of 16 bytes apart (%lld)", in_bs);
    const bool aligned = M >= per16 && ilog2_exact(M) >= 0;
    // the streaming kernel (wfx_ingest.hip) takes the ingest when the first window sits on the 16-byte grid (no tap shift needed)
    if (aligned && in_kind != WFX_IN_F64_MONO && fix_shift >= 8 && first >= 0 && (uint64_t)first < n_in && (first + misalign) % per16 == 0 &&
        !WFX_LAB_ENV("WFX_FE_NO_EXACT")) {
        int handled = 0;
        WFX_TRY(wfx_dev_ingest_stream(ctx, (const unsigned char *)in + (size_t)first * ebytes, in_kind, n_in - (uint64_t)first, M, coef, ntaps, fix_shift,
                                      0, nullptr, 0, out, n_out, nbatch, in_stride, out_stride, &handled));
        if (handled) {
            if (exact_out) *exact_out = 1;
            return 0;
        }
    }
    int d = 0;
    if (aligned) {
        d = (int)(((first + misalign) % per16 + per16) % per16);
        first -= d;
    }
    const int nt = ntaps + d;
    const int per_row = (nt + M - 1) / M;
    const int q4 = (per_row + 3) / 4;
    const long long ni = (long long)n_in, no = (long long)n_out;
    if (aligned && in_kind != WFX_IN_F64_MONO && fix_shift >= 8 && fix_shift <= 40 && !WFX_LAB_ENV("WFX_FE_NO_EXACT")) {
        // fixed-point taps; row r: E_hi[2 q4] E_lo[2 q4] D_hi[2 q4 + 2] D_lo[2 q4 + 2] as int16 pairs (E[p] = (c[2p], c[2p+1]) of the
        // row's taps c[k] = coef[r + M k - d], D[p] = (c[2p-1], c[2p]))
        const size_t rowlen = (size_t)8 * q4 + 4;
        std::vector<int32_t> fix((size_t)M * 4 * q4, 0);
        bool ok = true;
        for (int j = 0; j < ntaps && ok; ++j) {
            const double v = nearbyint(ldexp(coef[j], fix_shift));
            if (!(fabs(v) < (double)((1 << 27) - (1 << PP_FIX_LB)))) ok = false;        // (the high half is an int16)
            else fix[(size_t)((j + d) % M) * 4 * q4 + (size_t)((j + d) / M)] = (int32_t)v;
        }
        // worst-case |sample| = 32768: the hi parts and the lo parts of flush_rows rows each go into one int32 per output
        const int half = 1 << (PP_FIX_LB - 1);
        auto hi_of = [&](int32_t v) { return (v + half) >> PP_FIX_LB; };
        auto lo_of = [&](int32_t v) { return v - (hi_of(v) << PP_FIX_LB); };
        int flush_rows = 0;
        if (ok) {
            // rows are split over thread groups only in powers of two: a flush interval that divides every group's row count
            for (int fr = M; fr >= 1 && !flush_rows; fr >>= 1) {
                bool fits = true;
                for (int r0 = 0; r0 < M && fits; r0 += fr) {
                    long long sl = 0, sh = 0;
                    for (int r = r0; r < r0 + fr; ++r)
                        for (int k = 0; k < 4 * q4; ++k) {
                            sl += llabs((long long)lo_of(fix[(size_t)r * 4 * q4 + k]));
                            sh += llabs((long long)hi_of(fix[(size_t)r * 4 * q4 + k]));
                        }
                    if (sl * 32768 >= (1ll << 31) || sh * 32768 >= (1ll << 31)) fits = false;
                }
                if (fits) flush_rows = fr;
            }
            if (!flush_rows) ok = false;
        }
        if (ok) {
            std::vector<int32_t> tab((size_t)M * rowlen, 0);
            auto pack = [](int a, int b) { return (int32_t)(((uint32_t)(uint16_t)(int16_t)a) | ((uint32_t)(uint16_t)(int16_t)b << 16)); };
            for (int r = 0; r < M; ++r) {
                const int32_t *c = &fix[(size_t)r * 4 * q4];
                int32_t *row = &tab[(size_t)r * rowlen];
                const int np = 2 * q4;
                for (int p = 0; p < np; ++p) {
                    row[p] = pack(hi_of(c[2 * p]), hi_of(c[2 * p + 1]));
                    row[np + p] = pack(lo_of(c[2 * p]), lo_of(c[2 * p + 1]));
                }
                for (int p = 0; p <= np; ++p) {
                    const int32_t a = p > 0 ? c[2 * p - 1] : 0, b = p < np ? c[2 * p] : 0;
                    row[2 * np + p] = pack(hi_of(a), hi_of(b));
                    row[3 * np + 2 + p] = pack(lo_of(a), lo_of(b));
                }
            }
            const float *dtab = wfx_coef_device(ctx, (const float *)tab.data(), tab.size());
            if (!dtab) return WFX_ERR_HIP;
            if (exact_out) *exact_out = 1;
            const int fr = flush_rows | (fix_shift << 8);
            return in_kind == WFX_IN_I16_MONO ? launch_decimate64<WFX_IN_I16_MONO, 1>(ctx, in, ni, first, M, dtab, q4, out, no, true, fr, nbatch, in_bs, out_bs)
                                              : launch_decimate64<WFX_IN_I16_STEREO, 1>(ctx, in, ni, first, M, dtab, q4, out, no, true, fr, nbatch, in_bs, out_bs);
        }
    }
    std::vector<double> cp((size_t)M * 4 * q4, 0.0);
    for (int j = 0; j < ntaps; ++j) cp[(size_t)((j + d) % M) * 4 * q4 + (size_t)((j + d) / M)] = coef[j];
    const float *dcoef = wfx_coef_device(ctx, (const float *)cp.data(), cp.size() * 2);
    if (!dcoef) return WFX_ERR_HIP;
    switch (in_kind) {
    case WFX_IN_I16_MONO: return launch_decimate64<WFX_IN_I16_MONO, 2>(ctx, in, ni, first, M, dcoef, q4, out, no, aligned, 0, nbatch, in_bs, out_bs);
    case WFX_IN_I16_STEREO: return launch_decimate64<WFX_IN_I16_STEREO, 2>(ctx, in, ni, first, M, dcoef, q4, out, no, aligned, 0, nbatch, in_bs, out_bs);
    default: return launch_decimate64<WFX_IN_F64_MONO, 2>(ctx, in, ni, first, M, dcoef, q4, out, no, aligned, 0, nbatch, in_bs, out_bs);
    }
}
