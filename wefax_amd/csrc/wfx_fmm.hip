// a7 without a transform over the whole capture (round 5): the imaginary part of scipy.signal.hilbert (wefax.py:174) as
//
//     H[n] = (2/N) sum over m with n - m odd of cot(pi (n - m) / N) x[m]                        (N even)
//
// split per leaf of <= 64 samples into a NEAR field (the leaf and its two neighbours, summed directly from a table of the kernel at
// odd lags) and a FAR field carried by a one-dimensional fast multipole method on 16 Chebyshev nodes per box (tools/farfield_model.py
// is the NumPy statement of the same arithmetic and its gate: 1e-14 relative against scipy on BASELINE configs[1]).  Targets of one
// parity hear sources of the other parity only, so every box carries two weight vectors.  The capture is read twice (once per
// kernel below) instead of six transform passes; nothing global is exchanged between boxes but 16 numbers per box and parity --
// the form that shards with KB-sized exchanges.
//
//   fmm_up_leaf     a workgroup = 128 consecutive leaves: P2M (Chebyshev moments of every leaf and parity, reduced through LDS,
//                   turned into nodal weights) and seven levels of M2M; every level's weights go to memory
//   fmm_top         ONE workgroup: M2M up to level 2, then M2L + L2L down to the level of the workgroups' subtree roots
//   fmm_down_leaf   the same 128 leaves: seven levels of L2L + M2L (neighbours' weights from memory), nodal values -> Chebyshev
//                   coefficients, then per leaf: far field by the three-term recurrence, near field from a window of three leaves
//                   in LDS, |x + iH| (or H itself) out
//
// All float64.  Roofline: the two leaf kernels are f64-FMA-bound (~175 FMAs per sample: near field 82, P2M 32, L2P 32, M2L 28).
#include <algorithm>
#include <cmath>
#include <cstdlib>

#include "wfx_internal.h"
#include "wfx_notch.h"

namespace {

constexpr int FP = 16;                      // Chebyshev nodes per box
constexpr int FW = 6;                       // levels inside a leaf workgroup's subtree
constexpr int FLV = 1 << FW;                // leaves per leaf workgroup
constexpr int FTH = 512;                    // its threads: 8 waves x 8 leaves
constexpr int FNEAR = 256;                  // near-field table: odd lags -255 .. 255
constexpr int FXW = FLV * 64 + 2 * 64;      // capacity of a leaf workgroup's sample window (its leaves and one more on either side)
constexpr int FTD = 6;                      // deepest tier of the kernels between the leaf workgroups and the top
constexpr int FHB = 3;                      // boxes beyond either end of a subtree that its interaction lists reach

struct fmm_geom {
    long long n;
    int L;                                  // leaves = 2^L, each (k n / 2^L, (k+1) n / 2^L]: <= 64 samples
    double scale;                           // 2 / n
    double du;                              // a sample's step in a leaf's box coordinate: 2^(L+1) / n
};

// global index of box b of level lev (levels 2 ..): both parities of a box lie side by side, 2 x 16 doubles
__host__ __device__ inline size_t fmm_box(int lev, long long b) { return (size_t)(((1ll << lev) - 4) + b); }

__device__ __forceinline__ long long fmm_leaf_first(const fmm_geom &g, long long k)       // first sample of leaf k (k may be nleaf: = n)
{
    return (k * g.n + ((1ll << g.L) - 1)) >> g.L;
}

struct fmm_tabs {
    const double *At;       // [2][16 j][16 i]: A_c[i][j] = S_j(parent)(u_i(child c)), stored j-major (lanes run over i)
    const double *Aj;       // [2][16 i][16 j]: the same, i-major (lanes run over j: M2M)
    const double *Cw;       // [16 k][16 j]: W_j = sum_k Cw[k][j] mu_k
    const double *Ca;       // [16 j][16 k]: a_k = sum_j Ca[j][k] L_j
    const double *G;        // [level][r - 2][2][16][16]: [..][0][j][i] = G_r[i][j], [..][1][j][i] = G_r[j][i]; G_r[i][j] = cot(pi (r + (c_i - c_j) / 2) / 2^level)
    const double *gnear;    // [FNEAR]: cot(pi d / n), d = 2 q - 255
};

// L2L + M2L of one level into `dst` ([box][parity][16]) for the nb boxes b0 .. b0 + nb of a level with nbl boxes.  `lsrc`: the parents' values;
// `wst`: weights of boxes b0 - 3 .. b0 + nb + 3 (wrapped round the circle by whoever loaded them), [slot][parity][16]; `Gs`: the level's
// M2L matrices [r - 2][2][16][16]; `At`: [2][16 j][16 i].  Lanes run over the target node i.
template <int NT>
__device__ __forceinline__ void fmm_down_level(const double *lsrc, double *dst, const double *wst, const double *Gs, const double *At, long long b0, int nb,
                                               long long nbl, int t)
{
    for (int it = t; it < nb * 2 * FP; it += NT) {
        const int i = it & 15, hh = (it >> 4) & 1, bb = it >> 5;
        const long long tb = b0 + bb;
        double v = 0.0;
        if (lsrc) {
            const double *lp = lsrc + ((bb >> 1) * 2 + hh) * FP;
            const double *am = At + (tb & 1) * FP * FP;
#pragma unroll
            for (int j = 0; j < FP; ++j) v = fma(am[j * FP + i], lp[j], v);
        }
        // interaction list: the children of the parent's neighbours that do not touch the box (raw offset r = target - source)
        const int par = (int)(tb & 1);
#pragma unroll
        for (int s = 0; s < 3; ++s) {
            int off = s == 0 ? (par ? -3 : -2) : (s == 1 ? (par ? -2 : 2) : (par ? 2 : 3));
            if (nbl == 4) {                                      // level 2: only the box opposite
                if (s > 0) break;
                off = 2;
            }
            const int r = -off;
            const double *w = wst + ((bb + off + FHB) * 2 + (1 - hh)) * FP;
            const double *gm = Gs + ((size_t)((r > 0 ? r : -r) - 2) * 2 + (r > 0 ? 0 : 1)) * FP * FP;
            double acc = 0.0;
#pragma unroll
            for (int j = 0; j < FP; ++j) acc = fma(gm[j * FP + i], w[j], acc);
            v += r > 0 ? acc : -acc;
        }
        dst[it] = v;
    }
}

// ---- the leaf phase on the matrix cores (round 6) ----------------------------------------------------------------------------------------
// The near field of 16 leaves at once is ONE Toeplitz product: with sources and targets counted from the target leaf's first sample
// (tau = 2 (16 I + i) + e, sigma = sigma0 + 2 (4 K + k)) the tap cot(pi (tau - sigma) / n) does not depend on the leaf, so
//     D[i][leaf] += A_{I,K}[i][k] * B_K[k][leaf],   A = taps (a 16 x 4 block of the Toeplitz matrix),  B = the leaves' samples
// is v_mfma_f64_16x16x4_f64 with the leaves on the 16 columns.  A wave takes (16 leaves, target parity e): two row blocks I (a leaf
// has <= 32 samples of a parity) x ~21 source blocks K; what the window of three leaves does not hold for a given leaf (sizes differ
// by one) is masked to zero in B.  Operand traffic: two ds_read_b64 per lane and 1024 multiply-adds (the VALU form: two per one).
// The far field (L2P) is evaluated by the lane that holds the target's near sum: accumulator register r of lane (g, leaf) is row
// g + 4 r, i.e. 8 targets of ONE leaf and parity per lane, whose 16 Chebyshev coefficients wait in registers.
typedef double fmm_d4 __attribute__((ext_vector_type(4)));
// workgroup barrier that orders LDS traffic only: __syncthreads() also drains every global load and store in flight (s_waitcnt vmcnt(0)) --
// the matrices requested two levels ahead, the weights a level has just sent to memory -- and a tree level then lasts one memory latency
__device__ __forceinline__ void fmm_lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
constexpr int UP3W = 6;                     // fmm_up_leaf3 on an int16 capture: waves per SIMD the register budget is set for
constexpr int FXP0 = 8, FXP1 = 24;          // unread-but-addressable doubles before / behind the sample window (masked operand reads)

// lab build (tools/build_variant.sh ... -DWFX_FMM_STAMPS): cycle stamps of the leaf kernels' phases, printed by the launcher
#ifdef WFX_FMM_STAMPS
__device__ unsigned long long fmm_stamp_buf[3][4096 * 16];
#define FSTAMP(kern, i)                                                                                  \
    do {                                                                                                 \
        if (threadIdx.x == 0 && blockIdx.x < 4096) fmm_stamp_buf[kern][blockIdx.x * 16 + (i)] = wall_clock64(); \
    } while (0)
#else
#define FSTAMP(kern, i)
#endif

constexpr int FS = 18;                      // LDS stride of a box's 16 numbers per parity (16 would put the 16 columns of an operand read into one bank pair)
constexpr int FWALL = ((2 << FW) - 2 + FW * 2 * FHB) * 2 * FS;      // every level's weights of a leaf workgroup's subtree, halos included (doubles)
static_assert(FWALL >= FXW + FXP0 + FXP1, "the sample window takes the place of the tree phase's weights");

// the M2L matrices of one level for the boxes of parity class c as A operands: offsets (-2, +2, +3) for even boxes, (-3, -2, +2) for odd ones;
// r = -offset selects G_|r| (r > 0) or -G_|r| transposed (r < 0).  Every matrix of the tables is stored [K index][row], so a lane's A operand
// of K-step ks is table[64 ks + lane].
__device__ __forceinline__ void fmm_ga_load(const double *__restrict__ Gl, int c, int lane, double (&ga)[3][4])
{
#pragma unroll
    for (int s = 0; s < 3; ++s) {
        const int off = s == 0 ? (c ? -3 : -2) : (s == 1 ? (c ? -2 : 2) : (c ? 2 : 3));
        const int r = -off;
        const double *gm = Gl + ((size_t)((r > 0 ? r : -r) - 2) * 2 + (r > 0 ? 0 : 1)) * FP * FP + lane;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) ga[s][ks] = r > 0 ? gm[64 * ks] : -gm[64 * ks];
    }
}

// ---- one level of the downward pass on the matrix cores, in two halves.  The 16 columns of a product are 16 boxes of ONE parity class
// c = wave & 1 (same interaction offsets, same L2L matrix; d is a template parameter: a run-time level turns the callers' register sets into
// an indexed array in scratch memory).  The M2L sums of ALL levels of a subtree depend on the weights alone, not on one another --
// only the L2L product walks down the levels.  Per level the first form ran sixteen DEPENDENT products (4 L2L + 12 M2L) behind a barrier and
// the kernel waited through six such chains (17 of the fused leaf kernel's 25 us per workgroup); now a wave first runs the M2L chains of all
// its tasks (<= 4, independent of one another: they pipeline) and keeps the sums in registers, then the levels are four products each.
// Tasks: d >= 5: (class, 16 boxes, weight parity) -- level 6 on all eight waves, level 5 on waves 0 .. 3; d <= 4: columns = (box, parity), waves
// 4 + 2 (d & 1) + class: every wave has at most three tasks, and all their matrices are requested before the first barrier.
template <int d>
__device__ __forceinline__ bool fmm_level_task(int wave, int col, int &q, int &h, bool &store)
{
    constexpr int nb = 1 << d, half = nb >> 1;
    if constexpr (d >= 5) {
        constexpr int ng = half >> 4;                   // groups of 16 boxes per class: 1 or 2
        if (wave >= 4 * ng) return false;
        const int rest = wave >> 1;
        q = 16 * (rest & (ng - 1)) + col;
        h = rest / ng;
        store = true;
        return true;
    } else {
        if ((wave >> 1) != 2 + (d & 1)) return false;   // waves 4, 5: levels 4 and 2; waves 6, 7: levels 3 and 1 (waves 0 .. 3 carry level 5)
        const int cc = col & (nb - 1);                  // (columns beyond the level's boxes repeat earlier ones and are not stored)
        q = cc & (half - 1);
        h = cc >> (d - 1);
        store = col < nb;
        return true;
    }
}

template <int d>
__device__ __forceinline__ fmm_d4 fmm_m2l_level(const double *wl, const double (&ga)[3][4], int wave, int lane)
{
    fmm_d4 acc = {0.0, 0.0, 0.0, 0.0};
    int q = 0, h = 0;
    bool store = false;
    if (!fmm_level_task<d>(wave, lane & 15, q, h, store)) return acc;
    const int c = wave & 1, kq = lane >> 4;
    const int b = 2 * q + c;
#pragma unroll
    for (int s = 0; s < 3; ++s) {
        const int off = s == 0 ? (c ? -3 : -2) : (s == 1 ? (c ? -2 : 2) : (c ? 2 : 3));
        const double *bp = wl + ((b + off + FHB) * 2 + (1 - h)) * FS + kq;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(ga[s][ks], bp[4 * ks], acc, 0, 0, 0);
    }
    return acc;
}

// acc = the level's M2L sums (fmm_m2l_level); + L2L from the parents in `src` ([box][h][FS]); -> dst (conv: as Chebyshev coefficients)
template <int d>
__device__ __forceinline__ void fmm_l2l_level(const double *src, double *dst, fmm_d4 acc, const double (&atr)[4], int wave, int lane, const double *__restrict__ conv)
{
    int q = 0, h = 0;
    bool store = false;
    if (!fmm_level_task<d>(wave, lane & 15, q, h, store)) return;
    const int c = wave & 1, kq = lane >> 4;
    const int b = 2 * q + c;
    const double *bp = src + (q * 2 + h) * FS + kq;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(atr[ks], bp[4 * ks], acc, 0, 0, 0);
    if (conv) {
        fmm_d4 cf = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) cf = __builtin_amdgcn_mfma_f64_16x16x4f64(conv[64 * ks + lane], acc[ks], cf, 0, 0, 0);
        acc = cf;
    }
    if (store) {
        double *o = dst + (b * 2 + h) * FS + kq;        // accumulator register r of lane (kq, column) is row kq + 4 r
#pragma unroll
        for (int r = 0; r < 4; ++r) o[4 * r] = acc[r];
    }
}

#define FMM_WALL(D_) (un + (((1 << (D_)) - 2) + 2 * FHB * ((D_) - 1)) * 2 * FS)
#define FMM_GLV(D_) (T.G + (size_t)(lv0 + (D_) - 2) * 4 * FP * FP)
#define FMM_L2L_STEP(D_, ACC, NLEV, CONVP)                                                                              \
    if ((D_) <= (NLEV)) {                                                                                               \
        if ((D_) > 1) fmm_lds_barrier();                                                                                \
        FSTAMP(1, (D_));                                                                                                \
        fmm_l2l_level<(D_)>(((D_) & 1) ? la : lb, ((D_) & 1) ? lb : la, ACC, atr, wave, lane, CONVP);                   \
    }
// the levels lv0 + 1 .. lv0 + NLEV (NLEV <= 6 at run time) of a subtree whose weights wait in `un` and whose root's values wait in `la`
#define FMM_DOWN_CHAIN(NLEV, CONV6)                                                                                     \
    {                                                                                                                   \
        fmm_d4 m6 = {0.0, 0.0, 0.0, 0.0}, m5 = m6, ms0 = m6, ms1 = m6;                                                  \
        {                                                                                                               \
            double gA[3][4], gB[3][4], gC[3][4];                                                                        \
            if (6 <= (NLEV)) fmm_ga_load(FMM_GLV(6), wave & 1, lane, gA);                                               \
            if (wave < 4) {                                                                                             \
                if (5 <= (NLEV)) fmm_ga_load(FMM_GLV(5), wave & 1, lane, gB);                                           \
            } else if (wave < 6) {                                                                                      \
                if (4 <= (NLEV)) fmm_ga_load(FMM_GLV(4), wave & 1, lane, gB);                                           \
                if (2 <= (NLEV)) fmm_ga_load(FMM_GLV(2), wave & 1, lane, gC);                                           \
            } else {                                                                                                    \
                if (3 <= (NLEV)) fmm_ga_load(FMM_GLV(3), wave & 1, lane, gB);                                           \
                if (1 <= (NLEV)) fmm_ga_load(FMM_GLV(1), wave & 1, lane, gC);                                           \
            }                                                                                                           \
            fmm_lds_barrier();                  /* the weights and the root's values are in LDS */                      \
            FSTAMP(1, 0);                                                                                               \
            if (6 <= (NLEV)) m6 = fmm_m2l_level<6>(FMM_WALL(6), gA, wave, lane);                                        \
            if (wave < 4) {                                                                                             \
                if (5 <= (NLEV)) m5 = fmm_m2l_level<5>(FMM_WALL(5), gB, wave, lane);                                    \
            } else if (wave < 6) {                                                                                      \
                if (4 <= (NLEV)) ms0 = fmm_m2l_level<4>(FMM_WALL(4), gB, wave, lane);                                   \
                if (2 <= (NLEV)) ms1 = fmm_m2l_level<2>(FMM_WALL(2), gC, wave, lane);                                   \
            } else {                                                                                                    \
                if (3 <= (NLEV)) ms0 = fmm_m2l_level<3>(FMM_WALL(3), gB, wave, lane);                                   \
                if (1 <= (NLEV)) ms1 = fmm_m2l_level<1>(FMM_WALL(1), gC, wave, lane);                                   \
            }                                                                                                           \
        }                                                                                                               \
        FMM_L2L_STEP(1, ms1, NLEV, nullptr)                                                                             \
        FMM_L2L_STEP(2, ms1, NLEV, nullptr)                                                                             \
        FMM_L2L_STEP(3, ms0, NLEV, nullptr)                                                                             \
        FMM_L2L_STEP(4, ms0, NLEV, nullptr)                                                                             \
        FMM_L2L_STEP(5, m5, NLEV, nullptr)                                                                              \
        FMM_L2L_STEP(6, m6, NLEV, CONV6)                                                                                \
    }

// The weights of levels ltop + 1 .. ltop + D of a workgroup's subtree (its boxes and three more on either side, round the circle) into LDS
// ([level][slot][parity][FS], level d at box offset (2^d - 2) + 6 (d - 1)), as ONE batch of independent 16-byte requests per lane: one
// memory latency for the whole tree phase (a loop per level waits for memory once per trip).
__device__ __forceinline__ void fmm_load_walls(double *un, const double *__restrict__ Wg, int ltop, int D, int t, long long blk)
{
    static_assert(FW == 6 && FHB == 3 && FTD <= 6, "the slot table below");
    constexpr int NV2MAX = ((2 << FW) - 2 + FW * 2 * FHB) * 2 * (FP / 2);      // pairs of doubles
    constexpr int NQ = (NV2MAX + FTH - 1) / FTH;
    const int nv2 = ((2 << D) - 2 + D * 2 * FHB) * 2 * (FP / 2);
    double2 wv[NQ];
    int wdst[NQ];
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        const int i2 = t + q * FTH;                     // pair j2 of vector v = (slot, parity)
        const int v = i2 >> 3, j2 = i2 & 7, bs = v >> 1, hh = v & 1;
        const int d = 1 + (bs >= 8) + (bs >= 18) + (bs >= 32) + (bs >= 54) + (bs >= 92);
        const int base = d == 1 ? 0 : (d == 2 ? 8 : (d == 3 ? 18 : (d == 4 ? 32 : (d == 5 ? 54 : 92))));
        const int lev = ltop + d;
        const long long sb = ((blk << d) - FHB + (bs - base) + (1ll << lev)) & ((1ll << lev) - 1);
        wdst[q] = i2 < nv2 ? v * FS + 2 * j2 : -1;
        wv[q] = i2 < nv2 ? *(const double2 *)(Wg + (fmm_box(lev, 0) + sb) * 2 * FP + hh * FP + 2 * j2) : make_double2(0.0, 0.0);
    }
#pragma unroll
    for (int q = 0; q < NQ; ++q)
        if (wdst[q] >= 0) *(double2 *)(un + wdst[q]) = wv[q];
}

// ---- downward pass inside a leaf subtree: six levels of L2L + M2L, out come the leaves' far fields as Chebyshev coefficients ----------------
__global__ void __launch_bounds__(FTH, 4) fmm_tree_leaf(const fmm_geom g, const fmm_tabs T, const double *__restrict__ Wg, const double *__restrict__ Lg,
                                                        double *__restrict__ Cg, int wg0)
{
    const long long blk = (long long)blockIdx.x + wg0;           // (a launch covers a chunk of the leaf workgroups: fmm_run)
    extern __shared__ __align__(16) double fl[];
    double *la = fl;                                    // [64][2][FS]
    double *lb = la + FLV * 2 * FS;                     // [32][2][FS]
    double *un = lb + (FLV / 2) * 2 * FS;               // the weights of all six levels
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int ltop = g.L - FW;
    FSTAMP(1, 0);
    fmm_load_walls(un, Wg, ltop, FW, t, blk);
    double atr[4];                                      // the L2L matrix of the wave's box class (wave & 1) as A operands
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) atr[ks] = T.At[(wave & 1) * FP * FP + 64 * ks + lane];
    if (t < 2 * FP) la[(t >> 4) * FS + (t & 15)] = Lg[fmm_box(ltop, blk) * 2 * FP + t];
    {
        const int lv0 = ltop;
        FMM_DOWN_CHAIN(FW, T.Ca)
    }
    double *src = la;                                   // (level 6 ends in the large buffer)
    fmm_lds_barrier();                                    // (src == la: the 64 leaves' Chebyshev coefficients)
    FSTAMP(1, 7);
    double *o = Cg + (size_t)blk * FLV * 2 * FP;
#pragma unroll
    for (int q = 0; q < (FLV * 2 * FP / 2) / FTH; ++q) {
        const int i2 = t + q * FTH;
        *(double2 *)(o + 2 * i2) = *(const double2 *)(src + (i2 >> 3) * FS + 2 * (i2 & 7));
    }
    FSTAMP(1, 8);
}

// ---- tree levels + leaves in ONE kernel (the coefficients never leave LDS): near field + far field -> H -> |x + iH| -> 5-tap median + level-0 histogram (OUT 0: H, 1: the envelope, 2: median + histogram) ----
// OUT 2: the envelope takes the samples' place in LDS once every near field has read them; medians of positions that need a neighbour
// workgroup's envelope (its first and last two) are left to fmm_edge_median, which finds the four envelope values at either end of every
// workgroup in `edge`.  scipy.signal.medfilt pads with ZEROS at the capture's ends: those four medians are complete here.
template <int OUT>
__global__ void __launch_bounds__(FTH, 4) fmm_tree_leaf_env(const double *__restrict__ x, const fmm_geom g, const fmm_tabs T, const double *__restrict__ Wg,
                                                            const double *__restrict__ Lg, double *__restrict__ out, int smax, int xcap, double *__restrict__ edge, unsigned *__restrict__ l0hist, int wg0, int xwrap)
{
    // xwrap 0 (a rank of a sharded decode): x is addressed by the UNWRAPPED sample index -- the leaf before the rank's first (for rank 0: the
    // capture's last) and the one behind its last lie in front of / behind its own samples in memory
    const long long blk = (long long)blockIdx.x + wg0;
    extern __shared__ __align__(16) double fl[];
    // ---- phase 1 (fmm_tree_leaf's): six levels of L2L + M2L; the 64 leaves' coefficients end in `la` -------------------------------------------
    double *la = fl;                                    // [64][2][FS]
    double *lb = la + FLV * 2 * FS;                     // [32][2][FS]
    double *un = lb + (FLV / 2) * 2 * FS;               // the weights of all six levels
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    FSTAMP(2, 0);
    {
        const int ltop = g.L - FW;
        fmm_load_walls(un, Wg, ltop, FW, t, blk);
        double atr[4];
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) atr[ks] = T.At[(wave & 1) * FP * FP + 64 * ks + lane];
        if (t < 2 * FP) la[(t >> 4) * FS + (t & 15)] = Lg[fmm_box(ltop, blk) * 2 * FP + t];
        const int lv0 = ltop;
        FMM_DOWN_CHAIN(FW, T.Ca)
    }
    FSTAMP(1, 7);
    // ---- phase 2 (fmm_leaf_env's): the coefficients stay where they are; the near-field table and the sample window take the place of the
    // weights.  A workgroup WAITS in phase 1 and ISSUES in phase 2: the two workgroups of a CU drift apart and fill each other's gaps ----
    double *ca = la;
    double *gn = lb;                                    // [FNEAR]
    double *xw = gn + FNEAR + FXP0;                     // the sample window (xcap doubles + pads)
    unsigned *h0 = (unsigned *)ca;
    static_assert((FLV / 2) * 2 * FS + FWALL >= FNEAR + FXP0 + FXW + FXP1, "the window fits where the weights were");
    // (measured and not kept: the window requested at the kernel's start and held in registers through phase 1 -- 18 more registers spill at
    // two workgroups per CU and the requests compete with the weights': phase 1 16.6 against 14.6 us)
    const long long leaf0 = blk * FLV, nleaf = 1ll << g.L;
    const long long km = leaf0 == 0 ? nleaf - 1 : leaf0 - 1;
    const long long w0 = fmm_leaf_first(g, km) - (leaf0 == 0 ? g.n : 0);                       // may be negative
    const long long kl = leaf0 + FLV == nleaf ? 0 : leaf0 + FLV;                              // the leaf behind the last one
    const long long w1 = (leaf0 + FLV == nleaf ? g.n : 0) + fmm_leaf_first(g, kl + 1);
    const int wlen = (int)(w1 - w0);                                                            // <= FXW
    constexpr int XPT = (FXW + FTH - 1) / FTH;
    {
        double xr[XPT];
#pragma unroll
        for (int q = 0; q < XPT; ++q) {
            const int idx = t + q * FTH;
            long long m = w0 + idx;
            if (xwrap) m = m < 0 ? m + g.n : (m >= g.n ? m - g.n : m);
            xr[q] = idx < wlen ? x[m] : 0.0;
        }
        fmm_lds_barrier();                              // every wave has read the last level's operands: `lb` and `un` are free
        for (int i = t; i < FNEAR; i += FTH) gn[i] = T.gnear[i];
#pragma unroll
        for (int q = 0; q < XPT; ++q) {
            const int idx = t + q * FTH;
            if (idx < xcap) xw[idx] = xr[q];
        }
    }
    fmm_lds_barrier();
    FSTAMP(2, 1);
    // ---- wave = (16 leaves G, target parity e relative to the leaf's first sample) ---------------------------------------------------------
    const int n16 = lane & 15, gq = lane >> 4;
    const int SMe = (smax + 1) & ~1;
    const int NKB = ((2 * smax + SMe) / 2 + 3) / 4;
    const int G = wave >> 1, e = wave & 1;
    const int lk = 16 * G + n16;
    const long long k = leaf0 + lk;
    const long long a = fmm_leaf_first(g, k), b = fmm_leaf_first(g, k + 1), am = fmm_leaf_first(g, k - 1), ap2 = fmm_leaf_first(g, k + 2);
    const int s0 = (int)(b - a), slo = -(int)(a - am), shi = (int)(ap2 - a);      // sources of the near field: slo <= sigma < shi
    const int sigma0 = -SMe + (1 - e);
    const double *bp = xw + ((int)(a - w0) + sigma0 + 2 * gq);                   // B: lane (k = gq, column n16) reads sample sigma0 + 2 (4 K + gq) of ITS leaf
    int sig = sigma0 + 2 * gq;
    const double *ap = gn + (n16 - gq + e + (SMe + 254) / 2);                     // A: lane (row n16, k = gq) reads the tap of lag 2 (16 I + row - 4 K - k + e) + SMe - 1
    fmm_d4 acc0 = {0.0, 0.0, 0.0, 0.0}, acc1 = {0.0, 0.0, 0.0, 0.0};
    // (operands of block K + 1 are requested before block K's products are issued; the window's pads make the last, unused request legal)
    double bn = bp[0], a0n = ap[0], a1n = ap[16];
    for (int K = 0; K < NKB; ++K) {
        const bool in = (sig >= slo) & (sig < shi);
        const double bv = in ? bn : 0.0, a0 = a0n, a1 = a1n;
        sig += 8;
        bn = bp[8 * (K + 1)];
        a0n = ap[-4 * (K + 1)];
        a1n = ap[16 - 4 * (K + 1)];
        acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, bv, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, bv, acc1, 0, 0, 0);
    }
    FSTAMP(2, 2);
    // far field + output: register r of block I is target tau = 2 (16 I + gq + 4 r) + e of leaf lk.  Its place in the box: u = ua + tau du
    // (one division per lane, not per target)
    const int ht = (int)((a + e) & 1);
    const double ua = 2.0 * ((double)((a << g.L) - k * g.n) / (double)g.n) - 1.0;
    double ac[FP];
    {
        const double *cp = ca + (lk * 2 + ht) * FS;
#pragma unroll
        for (int kk = 0; kk < FP; ++kk) ac[kk] = cp[kk];
    }
    double *xp = xw + (int)(a - w0);
    double *op = out + a;
    double res[8];
#pragma unroll
    for (int I = 0; I < 2; ++I) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int tau = 2 * (16 * I + gq + 4 * r) + e;
            const bool valid = tau < s0;
            const double u = valid ? fma((double)tau, g.du, ua) : 0.0;
            double t0 = 1.0, t1 = u, far = fma(ac[1], u, ac[0]);
            const double u2 = 2.0 * u;
#pragma unroll
            for (int kk = 2; kk < FP; ++kk) {
                const double t2 = fma(u2, t1, -t0);
                far = fma(ac[kk], t2, far);
                t0 = t1;
                t1 = t2;
            }
            const double near = I == 0 ? acc0[r] : acc1[r];
            const double H = g.scale * (far + near);
            if (OUT == 0) {
                if (valid) op[tau] = H;
            } else {
                const double xv = xp[valid ? tau : 0];
                const double ev = sqrt(fma(xv, xv, H * H));
                if (OUT == 1) {
                    if (valid) op[tau] = ev;
                } else {
                    res[4 * I + r] = ev;
                }
            }
        }
    }
    FSTAMP(2, 3);
    if (OUT != 2) return;
    fmm_lds_barrier();                                    // every near field has read its samples: the envelope takes their place
    static_assert(FLV * 2 * FS * 8 >= WFX_SEL_BINS * 4, "the histogram fits where the coefficients were");
    for (int i = t; i < WFX_SEL_BINS; i += FTH) h0[i] = 0;
#pragma unroll
    for (int I = 0; I < 2; ++I)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int tau = 2 * (16 * I + gq + 4 * r) + e;
            if (tau < s0) xp[tau] = res[4 * I + r];
        }
    fmm_lds_barrier();
    FSTAMP(2, 4);
    {
        const long long wa = fmm_leaf_first(g, leaf0), wb = fmm_leaf_first(g, leaf0 + FLV);   // the workgroup's own samples [wa, wb)
        const int own = (int)(wb - wa);
        const double *ew = xw + (int)(wa - w0);
        const bool first = wa == 0, last = wb == g.n;
        for (int i0 = 0; i0 < own; i0 += FTH) {
            const int i = i0 + t;
            // medians that are complete here: all but the first and last two -- and those too where zeros stand beyond the capture's end
            const bool valid = i < own && (i >= 2 || first) && (i < own - 2 || last);
            double m = 0.0;
            if (valid) {
                const double e0 = i >= 2 ? ew[i - 2] : 0.0, e1 = i >= 1 ? ew[i - 1] : 0.0;
                const double e3 = i + 1 < own ? ew[i + 1] : 0.0, e4 = i + 2 < own ? ew[i + 2] : 0.0;
                m = wfx_median5(e0, e1, ew[i], e3, e4);
                out[wa + i] = m;
            }
            if (l0hist) wfx_sel_count(h0, (unsigned)(wfx_f64_key(m) >> 53), valid, lane);
        }
        if (t < 8) edge[(size_t)blk * 8 + t] = t < 4 ? ew[t] : ew[own - 8 + t];
    }
    if (l0hist) {
        fmm_lds_barrier();
        for (int i = t; i < WFX_SEL_BINS; i += FTH)
            if (h0[i]) atomicAdd(&l0hist[i], h0[i]);
    }
    FSTAMP(2, 5);
}


// medians across the seam of workgroups b and b + 1 of fmm_leaf_env<2> (positions end - 2, end - 1, end, end + 1): one thread per seam
__global__ void __launch_bounds__(256) fmm_edge_median(const fmm_geom g, const double *__restrict__ edge, double *__restrict__ out, unsigned *__restrict__ l0hist,
                                                       int b_lo, int b_hi, int own_lo, int own_hi)
{
    // seams b in [b_lo, b_hi): between workgroups b and b + 1.  A rank of a sharded decode writes (and counts) only the medians of workgroups
    // it owns, [own_lo, own_hi): a seam with a neighbour rank is done half by either side (the neighbour's four values arrived in `edge`).
    // (the histogram goes through LDS with one atomic per wave and digit: 8 000 global atomics on a handful of bins took 69 us)
    __shared__ unsigned h0[WFX_SEL_BINS];
    const int t = threadIdx.x;
    const int b = b_lo + blockIdx.x * 256 + t;
    const bool act = b < b_hi;
    const bool wl = act && b >= own_lo && b < own_hi, wr = act && b + 1 >= own_lo && b + 1 < own_hi;
    if (l0hist)
        for (int i = t; i < WFX_SEL_BINS; i += 256) h0[i] = 0;
    __syncthreads();
    double m0 = 0.0, m1 = 0.0, m2 = 0.0, m3 = 0.0;
    if (act) {
        const long long pos = fmm_leaf_first(g, (long long)(b + 1) * FLV);         // first sample of workgroup b + 1
        const double *l = edge + (size_t)b * 8 + 4, *f = edge + (size_t)(b + 1) * 8;
        m0 = wfx_median5(l[0], l[1], l[2], l[3], f[0]);
        m1 = wfx_median5(l[1], l[2], l[3], f[0], f[1]);
        m2 = wfx_median5(l[2], l[3], f[0], f[1], f[2]);
        m3 = wfx_median5(l[3], f[0], f[1], f[2], f[3]);
        if (wl) {
            out[pos - 2] = m0;
            out[pos - 1] = m1;
        }
        if (wr) {
            out[pos] = m2;
            out[pos + 1] = m3;
        }
    }
    if (l0hist) {
        wfx_sel_count(h0, (unsigned)(wfx_f64_key(m0) >> 53), wl, t & 63);
        wfx_sel_count(h0, (unsigned)(wfx_f64_key(m1) >> 53), wl, t & 63);
        wfx_sel_count(h0, (unsigned)(wfx_f64_key(m2) >> 53), wr, t & 63);
        wfx_sel_count(h0, (unsigned)(wfx_f64_key(m3) >> 53), wr, t & 63);
        __syncthreads();
        for (int i = t; i < WFX_SEL_BINS; i += 256)
            if (h0[i]) atomicAdd(&l0hist[i], h0[i]);
    }
}

// ---- P2M + M2M on the matrix cores: a workgroup = 64 consecutive leaves ------------------------------------------------------------------------
// P2M: wave = (16 leaves G, sample parity e relative to the leaf's first sample); lane (column = leaf, quarter gq) sums the Chebyshev moments
// of its quarter of the leaf's samples of that parity (samples j = gq + 4 r: 31 multiply-adds each).  The sum over the four quarters and
// the step from moments to nodal weights are ONE product per moment k: A[row j][K index = quarter] = Cw[k][j] for all four quarters,
// B[quarter][leaf] = the lane's own register mu_k -- 16 products, no cross-lane traffic.  M2M: columns = (parent, parity), two matrices.
__device__ __forceinline__ void fmm_up_level_mfma(const double *src, double *dst, double *__restrict__ gout, const double (&ajr)[2][4], int d, int wave, int lane)
{
    const int ncol = 2 << d;                            // (parent, parity) pairs of the level
    if (16 * wave >= ncol) return;
    const int col = lane & 15, kq = lane >> 4;
    const int colg = 16 * wave + col, cc = colg & (ncol - 1);
    const bool valid = colg < ncol;
    const int p = cc >> 1, h = cc & 1;
    fmm_d4 acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int c = 0; c < 2; ++c) {
        const double *bp = src + ((2 * p + c) * 2 + h) * FS + kq;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(ajr[c][ks], bp[4 * ks], acc, 0, 0, 0);
    }
    if (valid) {
        double *o = dst + cc * FS + kq, *go = gout + cc * FP + kq;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            o[4 * r] = acc[r];
            go[4 * r] = acc[r];
        }
    }
}

__global__ void __launch_bounds__(FTH, 2) fmm_up_leaf2(const double *__restrict__ x, const fmm_geom g, const fmm_tabs T, double *__restrict__ Wg)
{
    extern __shared__ __align__(16) double fl[];
    double *wb0 = fl;                                   // [64][2][FS]
    double *wb1 = wb0 + FLV * 2 * FS;                   // [32][2][FS]
    double *xw = wb1 + (FLV / 2) * 2 * FS;              // the workgroup's samples: <= 64 x 64
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const long long leaf0 = (long long)blockIdx.x * FLV;
    const long long w0 = fmm_leaf_first(g, leaf0), w1 = fmm_leaf_first(g, leaf0 + FLV);
    const int wlen = (int)(w1 - w0);
    FSTAMP(0, 0);
    constexpr int XPT = (FLV * 64) / FTH;
#pragma unroll
    for (int q = 0; q < XPT; ++q) {
        const int idx = t + q * FTH;
        xw[idx] = idx < wlen ? x[w0 + idx] : 0.0;
    }
    double ajr[2][4];
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) ajr[c][ks] = T.Aj[c * FP * FP + 64 * ks + lane];
    const int n16 = lane & 15, gq = lane >> 4;
    const int G = wave >> 1, e = wave & 1;
    const int lk = 16 * G + n16;
    const long long k = leaf0 + lk;
    const long long a = fmm_leaf_first(g, k), b = fmm_leaf_first(g, k + 1);
    const int s0 = (int)(b - a);
    const int ht = (int)((a + e) & 1);
    const double ua = 2.0 * ((double)((a << g.L) - k * g.n) / (double)g.n) - 1.0;
    double cwa[FP];                                     // A operands: Cw[k][row], the same for the four quarters
#pragma unroll
    for (int kk = 0; kk < FP; ++kk) cwa[kk] = T.Cw[kk * FP + n16];
    fmm_lds_barrier();
    FSTAMP(0, 1);
    double mu[FP];
#pragma unroll
    for (int kk = 0; kk < FP; ++kk) mu[kk] = 0.0;
    const double *xp = xw + (int)(a - w0);
#pragma unroll
    for (int r = 0; r < 8; ++r) {
        const int tau = 2 * (gq + 4 * r) + e;
        const bool valid = tau < s0;
        const double xv = valid ? xp[valid ? tau : 0] : 0.0;
        const double u = valid ? fma((double)tau, g.du, ua) : 0.0;
        const double u2 = 2.0 * u;
        mu[0] += xv;
        mu[1] = fma(u, xv, mu[1]);
        double t0 = 1.0, t1 = u;
#pragma unroll
        for (int kk = 2; kk < FP; ++kk) {
            const double t2 = fma(u2, t1, -t0);
            mu[kk] = fma(t2, xv, mu[kk]);
            t0 = t1;
            t1 = t2;
        }
    }
    FSTAMP(0, 2);
    {
        fmm_d4 acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int kk = 0; kk < FP; ++kk) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(cwa[kk], mu[kk], acc, 0, 0, 0);
        double *o = wb0 + (lk * 2 + ht) * FS + gq, *go = Wg + (fmm_box(g.L, k) * 2 + ht) * FP + gq;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            o[4 * r] = acc[r];
            go[4 * r] = acc[r];
        }
    }
    double *src = wb0, *dst = wb1;
#pragma unroll 1
    for (int d = FW - 1; d >= 0; --d) {
        fmm_lds_barrier();
        FSTAMP(0, 3 + (FW - 1 - d));
        fmm_up_level_mfma(src, dst, Wg + fmm_box(g.L - FW + d, (long long)blockIdx.x << d) * 2 * FP, ajr, d, wave, lane);
        double *tmp = src;
        src = dst;
        dst = tmp;
    }
}

// ---- a6 + P2M + M2M in one kernel (round 6): the notch / slope filter (wefax.py:63-72) applied to the workgroup's own 64 leaves -----------
// The raw capture is read ONCE (int16: 2 bytes per sample instead of a float64 round trip through memory): window of the leaves' samples and
// 24 more on either side -> LDS -> the 49-tap form of filtfilt (notch_kernel's arithmetic, wfx_stages.hip: eight outputs per lane from a
// 56-sample register window, ascending taps) -> the filtered samples in LDS for P2M and in `y` for the leaf kernel.  The first and last
// 64 samples of the capture come from the exact recurrences (odd extension, lfilter_zi), run by the first / last workgroup.
template <typename TIN> struct up3_win;
template <> struct up3_win<short> {
    static __device__ __forceinline__ void load8(const short *w, double (&v)[8])
    {
        const uint4 q = *(const uint4 *)w;             // (8 t shorts from a 16-byte aligned base)
        const unsigned d[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            v[2 * i] = (double)(short)(d[i] & 0xffffu);
            v[2 * i + 1] = (double)((int)d[i] >> 16);
        }
    }
};
template <> struct up3_win<double> {
    static __device__ __forceinline__ void load8(const double *w, double (&v)[8])
    {
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = w[i];
    }
};

template <typename TIN, int MINW>
__global__ void __launch_bounds__(FTH, MINW) fmm_up_leaf3(const TIN *__restrict__ raw, const notch_coef c, double *__restrict__ y, const fmm_geom g, const fmm_tabs T,
                                                           double *__restrict__ Wg, int xc, wfx_dev_scalars *__restrict__ clear, int wg0)
{
    extern __shared__ __align__(16) double fl[];
    double *wb0 = fl;                                   // [64][2][FS]; before P2M: the raw window (int16 capture) and the edge routine's scratch
    double *xw = wb0 + FLV * 2 * FS;                    // the workgroup's filtered samples (xc = 64 x the largest leaf); after P2M: wb1
    double *wb1 = xw;                                   // [32][2][FS]
    TIN *rw = sizeof(TIN) == 2 ? (TIN *)wb0 : (TIN *)(xw + xc);        // raw[wa - 24 .. wb + 24) (a float64 capture has its own region)
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const long long blk = (long long)blockIdx.x + wg0;               // (a rank of a sharded decode launches its own workgroups only)
    const long long leaf0 = blk * FLV;
    const long long wa = fmm_leaf_first(g, leaf0), wb = fmm_leaf_first(g, leaf0 + FLV);
    const int own = (int)(wb - wa);
    FSTAMP(0, 0);
    if (clear && blockIdx.x == 0)        // the decode's device scalars start from zero (this is the first kernel of the path that sees them)
        for (int i = t; i < (int)(sizeof(wfx_dev_scalars) / 8); i += FTH) ((unsigned long long *)clear)[i] = 0ull;
    {
        constexpr int NQ = (FLV * 64 + 2 * NOTCH_K + FTH - 1) / FTH;
        TIN pre[NQ];
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const long long src = wa - NOTCH_K + t + q * FTH;
            pre[q] = raw[src < 0 ? 0 : (src >= g.n ? g.n - 1 : src)];
        }
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const int j = t + q * FTH;
            const long long src = wa - NOTCH_K + j;
            if (j < own + 2 * NOTCH_K + 8) rw[j] = (src >= 0 && src < g.n) ? pre[q] : (TIN)0;
        }
    }
    fmm_lds_barrier();
    FSTAMP(0, 1);
    if (8 * t < own) {
        // window element i feeds output u with tap |i - u - K|: ascending i for every output, as notch_kernel sums
        double acc[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) acc[u] = 0.0;
#pragma unroll
        for (int blk = 0; blk < 7; ++blk) {
            double wv[8];
            up3_win<TIN>::load8(rw + 8 * t + 8 * blk, wv);
#pragma unroll
            for (int ii = 0; ii < 8; ++ii) {
                const int i = 8 * blk + ii;
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int k = i - u - NOTCH_K;
                    if (k >= -NOTCH_K && k <= NOTCH_K) acc[u] = fma(c.g[k < 0 ? -k : k], wv[ii], acc[u]);
                }
            }
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) xw[8 * t + u] = acc[u];
    }
    fmm_lds_barrier();
    // the capture's first / last 64 samples: exact forward / backward recurrences on the odd extension (notch_kernel's edge workgroup)
    if (wa < NOTCH_EDGE || wb > g.n - NOTCH_EDGE) {
        constexpr int EL = NOTCH_EDGE + NOTCH_SETTLE, LEN = NOTCH_PAD + EL;     // 127, 136
        double *e = (double *)wb0;                                            // (the raw window is no longer needed)
        const bool left = wa < NOTCH_EDGE;                                     // (a capture of >= 32 768 samples: never both in one workgroup)
        for (int i = t; i < LEN; i += FTH)
            e[i] = left ? (i < NOTCH_PAD ? notch_left<TIN>(c, raw, NOTCH_PAD - i) : (double)raw[i - NOTCH_PAD])
                        : (i < EL ? (double)raw[g.n - EL + i] : notch_right<TIN>(c, raw, (uint64_t)g.n, i - EL + 1));
        fmm_lds_barrier();
        if (t == 0) {
            // left: exact forward start, backward started SETTLE samples to the right with a zero state
            // right: forward started SETTLE samples early with a zero state, exact backward start
            double z0 = left ? c.zi[0] * e[0] : 0.0, z1 = left ? c.zi[1] * e[0] : 0.0;
            for (int i0 = 0; i0 < LEN; i0 += 8) {
                double v[8];
#pragma unroll
                for (int k = 0; k < 8; ++k) v[k] = e[i0 + k];
#pragma unroll
                for (int k = 0; k < 8; ++k) v[k] = biquad_step(c, v[k], z0, z1);
#pragma unroll
                for (int k = 0; k < 8; ++k) e[i0 + k] = v[k];
            }
            z0 = left ? 0.0 : c.zi[0] * e[LEN - 1];
            z1 = left ? 0.0 : c.zi[1] * e[LEN - 1];
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
        fmm_lds_barrier();
        if (t < NOTCH_EDGE) {
            if (left)
                xw[t] = e[NOTCH_PAD + t];
            else
                xw[(int)(g.n - NOTCH_EDGE - wa) + t] = e[EL - NOTCH_EDGE + t];
        }
        fmm_lds_barrier();
    }
    FSTAMP(0, 2);
    for (int i = t; i < own; i += FTH) y[wa + i] = xw[i];
    // ---- P2M: wave = (16 leaves G, sample parity e relative to the leaf's first sample); lane (column = leaf, quarter gq) -----------------------
    const int n16 = lane & 15, gq = lane >> 4;
    const int G = wave >> 1, e = wave & 1;
    const int lk = 16 * G + n16;
    const long long k = leaf0 + lk;
    const long long a = fmm_leaf_first(g, k), b = fmm_leaf_first(g, k + 1);
    const int s0 = (int)(b - a);
    const int ht = (int)((a + e) & 1);
    const double ua = 2.0 * ((double)((a << g.L) - k * g.n) / (double)g.n) - 1.0;
    double mu[FP];
#pragma unroll
    for (int kk = 0; kk < FP; ++kk) mu[kk] = 0.0;
    const double *xp = xw + (int)(a - wa);
#pragma unroll
    for (int r = 0; r < 8; ++r) {
        const int tau = 2 * (gq + 4 * r) + e;
        const bool valid = tau < s0;
        const double xv = valid ? xp[valid ? tau : 0] : 0.0;
        const double u = valid ? fma((double)tau, g.du, ua) : 0.0;
        const double u2 = 2.0 * u;
        mu[0] += xv;
        mu[1] = fma(u, xv, mu[1]);
        double t0 = 1.0, t1 = u;
#pragma unroll
        for (int kk = 2; kk < FP; ++kk) {
            const double t2 = fma(u2, t1, -t0);
            mu[kk] = fma(t2, xv, mu[kk]);
            t0 = t1;
            t1 = t2;
        }
    }
    FSTAMP(0, 3);
    {
        // sum over the four quarters and moments -> nodal weights: one product per moment, A[row j][quarter] = Cw[k][j]
        fmm_d4 acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int kk = 0; kk < FP; ++kk) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(T.Cw[kk * FP + n16], mu[kk], acc, 0, 0, 0);
        double *o = wb0 + (lk * 2 + ht) * FS + gq, *go = Wg + (fmm_box(g.L, k) * 2 + ht) * FP + gq;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            o[4 * r] = acc[r];
            go[4 * r] = acc[r];
        }
    }
    double ajr[2][4];                                   // (requested here, not at the top: the filter and P2M phases need the registers)
#pragma unroll
    for (int cc = 0; cc < 2; ++cc)
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) ajr[cc][ks] = T.Aj[cc * FP * FP + 64 * ks + lane];
    double *src = wb0, *dst = wb1;
#pragma unroll 1
    for (int d = FW - 1; d >= 0; --d) {
        fmm_lds_barrier();
        FSTAMP(0, 4 + (FW - 1 - d));
        fmm_up_level_mfma(src, dst, Wg + fmm_box(g.L - FW + d, blk << d) * 2 * FP, ajr, d, wave, lane);
        double *tmp = src;
        src = dst;
        dst = tmp;
    }
}

// ---- the tiers and the top of the tree on the matrix cores (512 threads; the same level routines as the leaf kernels) --------------------
__global__ void __launch_bounds__(FTH) fmm_up_tier2(const fmm_geom g, const fmm_tabs T, double *__restrict__ Wg, int a, int D, int box0)
{
    const long long blk = (long long)blockIdx.x + box0;
    __shared__ __align__(16) double b0[(1 << FTD) * 2 * FS], b1[(1 << (FTD - 1)) * 2 * FS];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    double ajr[2][4];
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) ajr[c][ks] = T.Aj[c * FP * FP + 64 * ks + lane];
    const double *ch = Wg + fmm_box(a + D, blk << D) * 2 * FP;
    constexpr int NQ = ((1 << FTD) * 2 * FP / 2) / FTH;        // pairs per thread: 2
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        const int i2 = t + q * FTH;
        if (i2 < ((2 * FP / 2) << D)) *(double2 *)(b0 + (i2 >> 3) * FS + 2 * (i2 & 7)) = *(const double2 *)(ch + 2 * i2);
    }
    double *src = b0, *dst = b1;
#pragma unroll 1
    for (int d = D - 1; d >= 0; --d) {
        fmm_lds_barrier();
        fmm_up_level_mfma(src, dst, Wg + fmm_box(a + d, blk << d) * 2 * FP, ajr, d, wave, lane);
        double *tmp = src;
        src = dst;
        dst = tmp;
    }
}

__global__ void __launch_bounds__(FTH) fmm_down_tier2(const fmm_geom g, const fmm_tabs T, const double *__restrict__ Wg, double *__restrict__ Lg, int a, int D, int box0)
{
    const long long blk = (long long)blockIdx.x + box0;
    __shared__ __align__(16) double la[(1 << FTD) * 2 * FS], lb[(1 << (FTD - 1)) * 2 * FS], un[FWALL];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    fmm_load_walls(un, Wg, a, D, t, blk);
    double atr[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) atr[ks] = T.At[(wave & 1) * FP * FP + 64 * ks + lane];
    if (t < 2 * FP) la[(t >> 4) * FS + (t & 15)] = Lg[fmm_box(a, blk) * 2 * FP + t];
    {
        const int lv0 = a;
        FMM_DOWN_CHAIN(D, nullptr)
    }
    double *src = (D & 1) ? lb : la;                    // (level d writes lb for odd d)
    fmm_lds_barrier();
    double *o = Lg + fmm_box(a + D, blk << D) * 2 * FP;
    for (int i2 = t; i2 < ((2 * FP / 2) << D); i2 += FTH) {
        const double *p = src + (i2 >> 3) * FS + 2 * (i2 & 7);
        *(double2 *)(o + 2 * i2) = make_double2(p[0], p[1]);
    }
}

// the top: levels 2 .. atop (<= 5), one workgroup.  M2M up to level 2 with every level's weights kept in LDS (halo slots filled round the
// circle), level 2 by the scalar routine (its interaction list is the one box opposite), levels 3 .. atop on the matrix cores.
__global__ void __launch_bounds__(FTH) fmm_top2(const fmm_geom g, const fmm_tabs T, double *__restrict__ Wg, double *__restrict__ Lg, int atop)
{
    constexpr int WLV = (32 + 2 * FHB) * 2 * FS;        // one level's weights, halo slots included
    __shared__ __align__(16) double wl[4][WLV];         // levels 2 .. 5
    __shared__ __align__(16) double la[64 * 2 * FS], lb[64 * 2 * FS];
    __shared__ double w2[(4 + 2 * FHB) * 2 * FP], l2[4 * 2 * FP], Ats[2 * FP * FP], Gs[4 * FP * FP];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    double ajr[2][4], atr[4];
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) ajr[c][ks] = T.Aj[c * FP * FP + 64 * ks + lane];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) atr[ks] = T.At[(wave & 1) * FP * FP + 64 * ks + lane];
    for (int i = t; i < 2 * FP * FP; i += FTH) Ats[i] = T.At[i];
    for (int i = t; i < 4 * FP * FP; i += FTH) Gs[i] = T.G[i];                     // level 2
    const int ntop = 1 << atop;
    for (int i = t; i < ntop * 2 * FP; i += FTH) wl[atop - 2][(FHB * 2 + (i >> 4)) * FS + (i & 15)] = Wg[fmm_box(atop, 0) * 2 * FP + i];
    for (int lev = atop - 1; lev >= 2; --lev) {
        fmm_lds_barrier();
        fmm_up_level_mfma(&wl[lev + 1 - 2][FHB * 2 * FS], &wl[lev - 2][FHB * 2 * FS], Wg + fmm_box(lev, 0) * 2 * FP, ajr, lev, wave, lane);
    }
    fmm_lds_barrier();
    // halo slots of every level (three boxes before the first and behind the last, round the circle); level 2 also in the scalar routine's layout
    for (int lev = 2; lev <= atop; ++lev) {
        const int nb = 1 << lev;
        double *w = wl[lev - 2];
        for (int i = t; i < 2 * FHB * 2 * FP; i += FTH) {
            const int v = i >> 4, side = v / (FHB * 2), vv = v % (FHB * 2);                // side 0: slots 0..2, side 1: slots nb+3 .. nb+5
            const int slot = side ? nb + FHB + (vv >> 1) : (vv >> 1);
            const int sb = (slot - FHB + 4 * nb) & (nb - 1);
            w[(slot * 2 + (vv & 1)) * FS + (i & 15)] = w[((sb + FHB) * 2 + (vv & 1)) * FS + (i & 15)];
        }
    }
    fmm_lds_barrier();
    // the M2L sums of levels 3 .. 5 need the weights alone: they run here, beside level 2's scalar routine, and wait in registers
    fmm_d4 m3 = {0.0, 0.0, 0.0, 0.0}, m4 = m3, m5 = m3;
    {
        double g3[3][4], g4[3][4], g5[3][4];
        if (3 <= atop && (wave >> 1) == 3) fmm_ga_load(T.G + (size_t)(3 - 2) * 4 * FP * FP, wave & 1, lane, g3);
        if (4 <= atop && (wave >> 1) == 2) fmm_ga_load(T.G + (size_t)(4 - 2) * 4 * FP * FP, wave & 1, lane, g4);
        if (5 <= atop && wave < 4) fmm_ga_load(T.G + (size_t)(5 - 2) * 4 * FP * FP, wave & 1, lane, g5);
        if (3 <= atop && (wave >> 1) == 3) m3 = fmm_m2l_level<3>(wl[1], g3, wave, lane);
        if (4 <= atop && (wave >> 1) == 2) m4 = fmm_m2l_level<4>(wl[2], g4, wave, lane);
        if (5 <= atop && wave < 4) m5 = fmm_m2l_level<5>(wl[3], g5, wave, lane);
    }
    for (int i = t; i < (4 + 2 * FHB) * 2 * FP; i += FTH) w2[i] = wl[0][(i >> 4) * FS + (i & 15)];
    fmm_lds_barrier();
    fmm_down_level<FTH>(nullptr, l2, w2, Gs, Ats, 0, 4, 4, t);
    fmm_lds_barrier();
    for (int i = t; i < 4 * 2 * FP; i += FTH) la[(i >> 4) * FS + (i & 15)] = l2[i];
    double *src = la;
    if (3 <= atop) {
        fmm_lds_barrier();
        fmm_l2l_level<3>(la, lb, m3, atr, wave, lane, nullptr);
        src = lb;
    }
    if (4 <= atop) {
        fmm_lds_barrier();
        fmm_l2l_level<4>(lb, la, m4, atr, wave, lane, nullptr);
        src = la;
    }
    if (5 <= atop) {
        fmm_lds_barrier();
        fmm_l2l_level<5>(la, lb, m5, atr, wave, lane, nullptr);
        src = lb;
    }
    fmm_lds_barrier();
    for (int i = t; i < ntop * 2 * FP; i += FTH) Lg[fmm_box(atop, 0) * 2 * FP + i] = src[(i >> 4) * FS + (i & 15)];
}

// ---- host: tables -------------------------------------------------------------------------------------------------------------------
struct fmm_static {
    std::vector<double> At, Aj, Cw, Ca;
};

static double cheb_node(int i) { return cos((2 * i + 1) * M_PI / (2 * FP)); }
static double cheb_S(int j, double u)       // S_j(u) = 1/p + 2/p sum_k T_k(c_j) T_k(u)
{
    const double tj = acos(cheb_node(j)), tu = acos(std::min(1.0, std::max(-1.0, u)));
    double s = 1.0 / FP;
    for (int k = 1; k < FP; ++k) s += (2.0 / FP) * cos(k * tj) * cos(k * tu);
    return s;
}

static const fmm_static &fmm_static_tables()
{
    static fmm_static S;
    if (S.At.empty()) {
        S.At.resize(2 * FP * FP);
        S.Aj.resize(2 * FP * FP);
        S.Cw.resize(FP * FP);
        S.Ca.resize(FP * FP);
        for (int c = 0; c < 2; ++c)
            for (int i = 0; i < FP; ++i)
                for (int j = 0; j < FP; ++j) {
                    const double v = cheb_S(j, (cheb_node(i) + (c ? 1.0 : -1.0)) / 2);
                    S.At[(c * FP + j) * FP + i] = v;
                    S.Aj[(c * FP + i) * FP + j] = v;
                }
        for (int k = 0; k < FP; ++k)
            for (int j = 0; j < FP; ++j) {
                const double v = k == 0 ? 1.0 / FP : (2.0 / FP) * cos(k * acos(cheb_node(j)));
                S.Cw[k * FP + j] = v;       // W_j = sum_k Cw[k][j] mu_k
                S.Ca[j * FP + k] = v;       // a_k = sum_j Ca[j][k] L_j
            }
    }
    return S;
}

// cot(pi z / 2^lev) for z = r + delta: near z = 2^lev / 2 (the box opposite on level 2) through the tangent, accurate where the value is small
static double cot_unit(double z, int lev)
{
    const double period = (double)(1ll << lev);
    double y = z - period * nearbyint(z / period);                   // (-period/2, period/2]
    if (fabs(fabs(y) - period / 2) < period / 8) {
        const double e = (y > 0 ? y - period / 2 : y + period / 2);  // cot(pi/2 + pi e / period) = -tan(pi e / period)
        return -tan(M_PI * e / period);
    }
    return 1.0 / tan(M_PI * y / period);
}

}   // namespace

// ---- host -------------------------------------------------------------------------------------------------------------------------------
// Everything a run needs for a capture of n samples: geometry, device tables (cached per n), work arrays (whole-capture size: a rank of a
// sharded decode uses its own part and the boxes it receives), tiers, LDS sizes.  handled = 0: a length this form does not take (odd, short,
// too long).
// ---- scipy.signal.resample (wefax.py:160-161) as a multipole sum: DESIGN.md 8.1, the gated model is tools/resample_farfield_model.py ------
// Downsampling n0 -> num (num even, num < n0): y[k] = (-1)^k / n0 (C - sum_n x_n s_n cot(pi (k / num - n / n0))) + [k n0 = n num] x_n num / n0,
// s_n = sin(pi num n / n0), C = sum_n x_n cos(pi num n / n0).  The sources are the n0 input samples at n / n0, the targets the num outputs at
// k / num, both cut into the SAME 2^L dyadic arcs (<= 64 sources each); the tree between them is the Hilbert transform's (unit cotangent
// kernel, both parity slots used as two halves of one source set), the near field three source leaves per target leaf on the vector pipe
// (no two pairs share a lag: nothing for the matrix cores here).
struct rs_geom {
    fmm_geom s;                             // sources: n = n0
    fmm_geom t;                             // targets: n = num (same L)
    double kappa;                           // pi / (n0 num): the angle of one unit of m = k n0 - n num
    double inv_n0, ratio;                   // 1 / n0, num / n0
    unsigned pk, pn;                        // num / gcd, n0 / gcd: target j pk lies on source j pn
    int cmin, cmax;                         // sources in three consecutive leaves: at least 3 floor(n0 / 2^L), at most 3 ceil
};

// sin and cos of pi t for |t| <= 1/2 by their Taylor series in x = pi t (|x| <= 1.571: x^23 / 23! = 1.3e-18, x^24 / 24! = 8e-20), the sine as
// x (1 + x^2 P(x^2)) so that it keeps its RELATIVE accuracy where it is small; a quarter of the library routine's instructions (no range
// reduction, no special cases)
__device__ __forceinline__ void rs_sincos_half(double t, double *s, double *c)
{
    const double x = 3.14159265358979323846 * t, z = x * x;
    double ps = -1.0 / 51090942171709440000.0;                     // -1 / 21!
    ps = fma(ps, z, 1.0 / 121645100408832000.0);                   // 1 / 19!
    ps = fma(ps, z, -1.0 / 355687428096000.0);                     // -1 / 17!
    ps = fma(ps, z, 1.0 / 1307674368000.0);                        // 1 / 15!
    ps = fma(ps, z, -1.0 / 6227020800.0);                          // -1 / 13!
    ps = fma(ps, z, 1.0 / 39916800.0);                             // 1 / 11!
    ps = fma(ps, z, -1.0 / 362880.0);                              // -1 / 9!
    ps = fma(ps, z, 1.0 / 5040.0);                                 // 1 / 7!
    ps = fma(ps, z, -1.0 / 120.0);                                 // -1 / 5!
    ps = fma(ps, z, 1.0 / 6.0);                                    // 1 / 3!
    *s = fma(-x * z, ps, x);
    double pc = 1.0 / 1124000727777607680000.0;                    // 1 / 22!
    pc = fma(pc, z, -1.0 / 2432902008176640000.0);                 // -1 / 20!
    pc = fma(pc, z, 1.0 / 6402373705728000.0);                     // 1 / 18!
    pc = fma(pc, z, -1.0 / 20922789888000.0);                      // -1 / 16!
    pc = fma(pc, z, 1.0 / 87178291200.0);                          // 1 / 14!
    pc = fma(pc, z, -1.0 / 479001600.0);                           // -1 / 12!
    pc = fma(pc, z, 1.0 / 3628800.0);                              // 1 / 10!
    pc = fma(pc, z, -1.0 / 40320.0);                               // -1 / 8!
    pc = fma(pc, z, 1.0 / 720.0);                                  // 1 / 6!
    pc = fma(pc, z, -1.0 / 24.0);                                  // -1 / 4!
    pc = fma(pc, z, 0.5);
    *c = fma(-z, pc, 1.0);
}

// sin and cos of pi num n / n0 with the angle reduced in integers to the nearest multiple of pi: s_n multiplies cotangents of 1e9 exactly
// where it is small, so it needs RELATIVE accuracy there (an exact zero at the coincident pairs).  r = num n mod 2 n0.  (ONE routine for
// every place that needs a weight: a sample's weight must not depend on which kernel computed it)
__device__ __forceinline__ void rs_sin_cos_r(long long r, long long n0, double inv_n0, double *s, double *c)
{
    const int j = (2 * r >= n0) + (2 * r >= 3 * n0);
    double sv, cv;
    rs_sincos_half((double)(r - j * n0) * inv_n0, &sv, &cv);
    *s = (j & 1) ? -sv : sv;
    *c = (j & 1) ? -cv : cv;
}

// P2M + M2M of one workgroup of 64 source leaves (fmm_up_leaf2 with the weights x_n s_n, which it also leaves in `w` for the near field) and
// the workgroup's part of C
__global__ void __launch_bounds__(FTH, 2) rs_up_leaf(const double *__restrict__ x, const rs_geom rg, const fmm_tabs T, double *__restrict__ Wg, double *__restrict__ w,
                                                     double *__restrict__ cpart, int wg0, long long x_index0)
{
    const fmm_geom g = rg.s;
    extern __shared__ __align__(16) double fl[];
    double *wb0 = fl;                                   // [64][2][FS]
    double *wb1 = wb0 + FLV * 2 * FS;                   // [32][2][FS]
    double *xw = wb1 + (FLV / 2) * 2 * FS;              // the workgroup's weighted samples: <= 64 x 64
    __shared__ double cred[FTH / 64];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const long long blk = (long long)blockIdx.x + wg0;
    const long long leaf0 = blk * FLV;
    const long long w0 = fmm_leaf_first(g, leaf0), w1 = fmm_leaf_first(g, leaf0 + FLV);
    const int wlen = (int)(w1 - w0);
    constexpr int XPT = (FLV * 64) / FTH;
    double cp = 0.0;
    // the angle pi num n / n0 of sample n, kept as the integer r = num n mod 2 n0 and stepped by (num FTH) mod 2 n0 from one of the thread's
    // samples to the next; reduced to the nearest multiple of n0 by comparisons
    const long long two_n0 = 2 * g.n;
    long long r = (rg.t.n * (w0 + t)) % two_n0;
    const long long rstep = (rg.t.n * FTH) % two_n0;
#pragma unroll
    for (int q = 0; q < XPT; ++q) {
        const int idx = t + q * FTH;
        double wv = 0.0;
        if (idx < wlen) {
            const double xv = x[w0 + idx - x_index0];
            double sv, cv;
            rs_sin_cos_r(r, g.n, rg.inv_n0, &sv, &cv);
            wv = xv * sv;
            cp = fma(xv, cv, cp);
            w[w0 + idx] = wv;
        }
        r += rstep;
        r = r >= two_n0 ? r - two_n0 : r;
        xw[idx] = wv;
    }
    // the workgroup's part of C: a fixed order of additions (lanes by halving, then the eight waves one after the other)
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) cp += __shfl_xor(cp, o, 64);
    if (lane == 0) cred[wave] = cp;
    double ajr[2][4];
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) ajr[c][ks] = T.Aj[c * FP * FP + 64 * ks + lane];
    const int n16 = lane & 15, gq = lane >> 4;
    const int G = wave >> 1, e = wave & 1;
    const int lk = 16 * G + n16;
    const long long k = leaf0 + lk;
    const long long a = fmm_leaf_first(g, k), b = fmm_leaf_first(g, k + 1);
    const int s0 = (int)(b - a);
    const int ht = (int)((a + e) & 1);
    const double ua = 2.0 * ((double)((a << g.L) - k * g.n) / (double)g.n) - 1.0;
    double cwa[FP];
#pragma unroll
    for (int kk = 0; kk < FP; ++kk) cwa[kk] = T.Cw[kk * FP + n16];
    __syncthreads();
    if (t == 0) {
        double c8 = 0.0;
        for (int i = 0; i < FTH / 64; ++i) c8 += cred[i];
        cpart[blk] = c8;
    }
    double mu[FP];
#pragma unroll
    for (int kk = 0; kk < FP; ++kk) mu[kk] = 0.0;
    const double *xp = xw + (int)(a - w0);
#pragma unroll
    for (int r = 0; r < 8; ++r) {
        const int tau = 2 * (gq + 4 * r) + e;
        const bool valid = tau < s0;
        const double xv = valid ? xp[valid ? tau : 0] : 0.0;
        const double u = valid ? fma((double)tau, g.du, ua) : 0.0;
        const double u2 = 2.0 * u;
        mu[0] += xv;
        mu[1] = fma(u, xv, mu[1]);
        double t0 = 1.0, t1 = u;
#pragma unroll
        for (int kk = 2; kk < FP; ++kk) {
            const double t2 = fma(u2, t1, -t0);
            mu[kk] = fma(t2, xv, mu[kk]);
            t0 = t1;
            t1 = t2;
        }
    }
    {
        fmm_d4 acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int kk = 0; kk < FP; ++kk) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(cwa[kk], mu[kk], acc, 0, 0, 0);
        double *o = wb0 + (lk * 2 + ht) * FS + gq, *go = Wg + (fmm_box(g.L, k) * 2 + ht) * FP + gq;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            o[4 * r] = acc[r];
            go[4 * r] = acc[r];
        }
    }
    double *src = wb0, *dst = wb1;
#pragma unroll 1
    for (int d = FW - 1; d >= 0; --d) {
        fmm_lds_barrier();
        fmm_up_level_mfma(src, dst, Wg + fmm_box(g.L - FW + d, blk << d) * 2 * FP, ajr, d, wave, lane);
        double *tmp = src;
        src = dst;
        dst = tmp;
    }
}

// the groups' sums added in a fixed order: 256 runs of consecutive groups, then the runs by halving
__device__ __forceinline__ void rs_csum_groups(const double *__restrict__ gsum, int ngrp, double *part, double *__restrict__ csum)
{
    const int t = threadIdx.x;
    const int run = (ngrp + 255) / 256;
    double a = 0.0;
    for (int i = 0; i < run; ++i) {
        const int gi = t * run + i;
        if (gi < ngrp) a += gsum[gi];
    }
    part[t] = a;
    __syncthreads();
    for (int o = 128; o >= 1; o >>= 1) {
        if (t < o) part[t] += part[t + o];
        __syncthreads();
    }
    if (t == 0) *csum = part[0];
}

// C = the sum of the workgroups' parts in a fixed order: groups of `grp` consecutive parts (a power of two: the boxes of one level of the
// tree, so that ranks of a sharded decode own whole groups), then the groups one after the other
__global__ void __launch_bounds__(256) rs_csum(const double *__restrict__ cpart, int nparts, int grp, double *__restrict__ gsum, double *__restrict__ csum)
{
    __shared__ double part[256];
    const int t = threadIdx.x;
    const int ngrp = nparts / grp;                          // a power of two
    for (int gi = t; gi < ngrp; gi += 256) {
        double a = 0.0;
        for (int i = 0; i < grp; ++i) a += cpart[(size_t)gi * grp + i];
        gsum[gi] = a;
    }
    __syncthreads();
    rs_csum_groups(gsum, ngrp, part, csum);
}

// the first half alone, for the groups [g_lo, g_hi) of a rank
__global__ void __launch_bounds__(256) rs_csum_own(const double *__restrict__ cpart, int grp, double *__restrict__ gsum, int g_lo, int g_hi)
{
    const int gi = g_lo + (int)(blockIdx.x * 256 + threadIdx.x);
    if (gi >= g_hi) return;
    double a = 0.0;
    for (int i = 0; i < grp; ++i) a += cpart[(size_t)gi * grp + i];
    gsum[gi] = a;
}

// the second half alone (a sharded decode: the groups' sums have been gathered from all ranks)
__global__ void __launch_bounds__(256) rs_csum2(const double *__restrict__ gsum, int ngrp, double *__restrict__ csum)
{
    __shared__ double part[256];
    rs_csum_groups(gsum, ngrp, part, csum);
}

// near field + far field + the closing factors for one workgroup of 64 target leaves.  Its targets are taken 64 at a time, one per lane,
// whatever leaf they lie in (a wave's lanes span at most three leaves when a leaf holds >= 32 targets, more when it holds few): every lane
// walks ITS leaf's three source leaves in LDS (lanes of one leaf read the same address: a broadcast).
// Per pair: m = k n0 - n num (an exact integer in a double), x = kappa m, cot x = 1 / x - x / 3 - x^3 / 45 - 2 x^5 / 945 (|x| <= 2 pi / 2^L <=
// 0.0123: the next term is below 1e-17 of 1 / x).  Reciprocals four at a time from ONE v_rcp_f64 + Newton step (2e-15: tools/micro/rcp_f64.hip)
// of the product, 1 / x0 = x1 / (x0 x1) ...: 28 issue cycles a pair instead of 39.  The odd powers are sums over the leaf's window with
// per-leaf moments S_j = sum_i i^j w_i: -1/3 sum w x = -(kappa / 3) (m0 S0 - num S1), computed once per leaf; POLY (L < 13, where x^3 / 45
// still counts) evaluates them per pair instead.  A coincident pair has m = 0 AND an exactly zero weight: 1e-60 added to x (no change
// to any other x: |x| >= 7e-19) keeps its product a finite zero.
template <bool POLY>
__global__ void __launch_bounds__(FTH, 4) rs_leaf(const double *__restrict__ w, const double *__restrict__ x, const rs_geom rg, const double *__restrict__ Cg,
                                                  const double *__restrict__ csum, double *__restrict__ y, int swin, int wg0, int wrap, long long x_index0,
                                                  long long y_index0)
{
    const fmm_geom gs = rg.s, gt = rg.t;
    const long long blk = (long long)blockIdx.x + wg0;
    extern __shared__ __align__(16) double fl[];
    double *ca = fl;                                    // [64][FS]: Chebyshev coefficients of the leaves' far fields (both slots added)
    double *sm = ca + FLV * FS;                         // [64][2]: S0, S1 of the leaves' windows
    double *ww = sm + FLV * 2;                          // the weighted sources of leaves leaf0 - 1 .. leaf0 + 64 (+ 4 zeros)
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const long long leaf0 = blk * FLV, nleaf = 1ll << gs.L;
    const long long km = leaf0 == 0 ? nleaf - 1 : leaf0 - 1;
    const long long ws0 = fmm_leaf_first(gs, km) - (leaf0 == 0 ? gs.n : 0);                       // may be negative
    const long long kl = leaf0 + FLV == nleaf ? 0 : leaf0 + FLV;
    const long long ws1 = (leaf0 + FLV == nleaf ? gs.n : 0) + fmm_leaf_first(gs, kl + 1);
    const int wlen = (int)(ws1 - ws0);
    for (int idx = t; idx < swin; idx += FTH) {
        long long m = ws0 + idx;
        if (wrap) m = m < 0 ? m + gs.n : (m >= gs.n ? m - gs.n : m);
        ww[idx] = idx < wlen ? w[m] : 0.0;
    }
    {
        const double *cg = Cg + (size_t)blk * FLV * 2 * FP;
        for (int i = t; i < FLV * FP; i += FTH) {
            const int lf = i >> 4, kk = i & 15;
            ca[lf * FS + kk] = cg[(lf * 2) * FP + kk] + cg[(lf * 2 + 1) * FP + kk];
        }
    }
    fmm_lds_barrier();
    if (!POLY) {
        // the moments of every leaf's window, all 64 leaves at once: eight lanes a leaf, each over every eighth source, then three halvings
        // (a fixed order of additions)
        {
            const int lk = t >> 3, part = t & 7;
            const long long k = leaf0 + lk;
            const long long sa = k == 0 ? fmm_leaf_first(gs, nleaf - 1) - gs.n : fmm_leaf_first(gs, k - 1);
            const long long sb = k + 2 > nleaf ? fmm_leaf_first(gs, k + 2 - nleaf) + gs.n : fmm_leaf_first(gs, k + 2);
            const int cnt = (int)(sb - sa);
            const double *wp = ww + (int)(sa - ws0);
            double s0 = 0.0, s1 = 0.0;
            for (int i = part; i < cnt; i += 8) {
                const double wv = wp[i];
                s0 += wv;
                s1 = fma((double)i, wv, s1);
            }
#pragma unroll
            for (int o = 4; o >= 1; o >>= 1) {
                s0 += __shfl_xor(s0, o, 64);
                s1 += __shfl_xor(s1, o, 64);
            }
            if (part == 0) {
                sm[2 * lk] = s0;
                sm[2 * lk + 1] = s1;
            }
        }
        fmm_lds_barrier();
    }
    const double cs = *csum;
    const double numd = (double)gt.n;
    const long long T0 = fmm_leaf_first(gt, leaf0), T1 = fmm_leaf_first(gt, leaf0 + FLV);
    const int nchunk = (int)((T1 - T0 + 63) >> 6);
#pragma unroll 1
    for (int ch = wave; ch < nchunk; ch += FTH / 64) {
        const long long kraw = T0 + 64ll * ch + lane;
        const bool valid = kraw < T1;
        const long long kt = valid ? kraw : T1 - 1;
        // the target's leaf floor(kt 2^L / num): an estimate in floating point, put right against the leaves' first targets
        long long k = (long long)((double)kt * rg.t.du * 0.5);
        k = k > nleaf - 1 ? nleaf - 1 : k;
        k -= fmm_leaf_first(gt, k) > kt;
        k += fmm_leaf_first(gt, k + 1) <= kt;
        const int lk = (int)(k - leaf0);
        const long long sa = k == 0 ? fmm_leaf_first(gs, nleaf - 1) - gs.n : fmm_leaf_first(gs, k - 1);
        const long long sb = k + 2 > nleaf ? fmm_leaf_first(gs, k + 2 - nleaf) + gs.n : fmm_leaf_first(gs, k + 2);
        const int cnt = (int)(sb - sa);
        // (a leaf holds floor or ceil(n0 / 2^L) sources: three of them between rg.cmin and rg.cmax -- bounds every lane shares, no vote)
        const int cmin = rg.cmin, cmax = rg.cmax;
        const double *wp = ww + (int)(sa - ws0);
        const double m0 = (double)(kt * gs.n - sa * gt.n);
        double m = m0;
        double acc0 = 0.0, acc1 = 0.0;
        // eight pairs a trip, ONE reciprocal: sum_j w_j / x_j = N / (x_0 .. x_7) with the numerator built pairwise --
        // (w0 x1 + w1 x0) x2 x3 + (w2 x3 + w3 x2) x0 x1 for four, two fours combined the same way: 40 operations + one v_rcp_f64 per 8 pairs
        // (back-substituting every 1 / x_j from the product's reciprocal, four at a time: 46 + 2).  The product of eight stays between
        // 1e-111 (a coincident pair's 1e-60 and seven neighbours >= pi / N0 apart) and 1e-15: no underflow, no overflow of its reciprocal
        const int c4 = cmin & ~7;
        double accp = 0.0;
        for (int i = 0; i < c4; i += 8) {
            double xv[8], wv[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                wv[j] = wp[i + j];
                xv[j] = fma(m, rg.kappa, 1e-60);
                m -= numd;
            }
            const double p01 = xv[0] * xv[1], p23 = xv[2] * xv[3], p45 = xv[4] * xv[5], p67 = xv[6] * xv[7];
            const double pa = p01 * p23, pb = p45 * p67;
            const double n01 = fma(wv[1], xv[0], wv[0] * xv[1]), n23 = fma(wv[3], xv[2], wv[2] * xv[3]);
            const double n45 = fma(wv[5], xv[4], wv[4] * xv[5]), n67 = fma(wv[7], xv[6], wv[6] * xv[7]);
            const double na = fma(n23, p01, n01 * p23), nb = fma(n67, p45, n45 * p67);
            const double nn = fma(nb, pa, na * pb), pp = pa * pb;
            double R = __builtin_amdgcn_rcp(pp);
            const double e1 = fma(-pp, R, 1.0);
            R = fma(R, e1, R);
            acc0 = fma(nn, R, acc0);
            if (POLY) {
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const double q = xv[j] * xv[j];
                    accp = fma(wv[j], xv[j] * fma(q, fma(q, 2.0 / 945.0, 1.0 / 45.0), 1.0 / 3.0), accp);
                }
            }
        }
        if (POLY) acc0 -= accp;
        for (int i = c4; i < cmax; ++i) {                          // the last few, and what only some lanes' leaves hold
            const double wv = i < cnt ? wp[i] : 0.0;
            const double xx = fma(m, rg.kappa, 1e-60);
            m -= numd;
            double r = __builtin_amdgcn_rcp(xx);
            const double e1 = fma(-xx, r, 1.0);
            r = fma(r, e1, r);
            if (POLY) {
                const double q = xx * xx;
                r = fma(-xx, fma(q, fma(q, 2.0 / 945.0, 1.0 / 45.0), 1.0 / 3.0), r);
            }
            acc0 = fma(wv, r, acc0);
        }
        double acc = acc0 + acc1;
        if (!POLY) acc = fma(-rg.kappa * (1.0 / 3.0), fma(m0, sm[2 * lk], -numd * sm[2 * lk + 1]), acc);
        // far field: the target's place in its box
        const double u = fma((double)((kt << gt.L) - k * gt.n), rg.t.scale, -1.0);
        const double *cp = ca + lk * FS;
        double t0 = 1.0, t1 = u, far = fma(cp[1], u, cp[0]);
        const double u2 = 2.0 * u;
#pragma unroll
        for (int kk = 2; kk < FP; ++kk) {
            const double t2 = fma(u2, t1, -t0);
            far = fma(cp[kk], t2, far);
            t0 = t1;
            t1 = t2;
        }
        double yv = ((kt & 1) ? -rg.inv_n0 : rg.inv_n0) * (cs - (far + acc));
        // a target that coincides with a source: D(0) = (num + 1) / n0, of which the C term carries 1 / n0
        // (k n0 = n num  <=>  k = j num / g, n = j n0 / g with g = gcd(n0, num))
        const unsigned jq = (unsigned)kt / rg.pk;
        if (jq * rg.pk == (unsigned)kt) yv = fma(x[(long long)jq * rg.pn - x_index0], rg.ratio, yv);
        if (valid) y[kt - y_index0] = yv;
    }
}

struct fmm_plan {
    fmm_geom g;
    fmm_tabs T;
    int L = 0, smax = 0, xcap = 0, xc3 = 0, ntier = 0, tier_a[8], tier_d[8], atop = 0;
    unsigned nwg = 0;
    double *Wg = nullptr, *Lg = nullptr, *Cg = nullptr, *Eg = nullptr;
    size_t lds_tree = 0, lds_leaf = 0, lds_up2 = 0, lds_up3_i16 = 0, lds_up3_f64 = 0;
};

static bool fmm_levels(uint64_t n, int *L_out)
{
    if (n % 2 || n < 32768 || n > (1ull << 32)) return false;
    int L = 0;
    while (((double)n / (double)(1ull << L)) > 64.0) ++L;                 // leaf size in (32, 64]
    if (L < FW + 2 || L > 26) return false;                               // ((sample << L) stays inside 63 bits; at least level 2 above the leaf roots)
    *L_out = L;
    return true;
}

static int fmm_setup(wfx_ctx *ctx, uint64_t n, fmm_plan &P, int *handled)
{
    *handled = 0;
    int L = 0;
    if (!fmm_levels(n, &L)) return 0;
    // device tables, cached per n: the static ones, the M2L matrices of levels 2..L (unit kernel), the near table of THIS n
    const double *dt = nullptr;
    for (auto &e : ctx->fmm_tables)
        if (e.first == n) dt = e.second;
    size_t off_g = 4 * FP * FP + 2 * FP * FP, off_n = off_g + (size_t)(L - 1) * 4 * FP * FP;
    if (!dt) {
        const fmm_static &S = fmm_static_tables();
        std::vector<double> tab;
        tab.insert(tab.end(), S.At.begin(), S.At.end());
        tab.insert(tab.end(), S.Aj.begin(), S.Aj.end());
        tab.insert(tab.end(), S.Cw.begin(), S.Cw.end());
        tab.insert(tab.end(), S.Ca.begin(), S.Ca.end());
        for (int lev = 2; lev <= L; ++lev)
            for (int r = 2; r <= 3; ++r)
                for (int tr = 0; tr < 2; ++tr)
                    for (int j = 0; j < FP; ++j)
                        for (int i = 0; i < FP; ++i) {
                            const int ii = tr ? j : i, jj = tr ? i : j;      // tr = 0: [j][i] holds G[i][j]; tr = 1: [j][i] holds G[j][i]
                            tab.push_back(cot_unit(r + (cheb_node(ii) - cheb_node(jj)) / 2, lev));
                        }
        for (int q = 0; q < FNEAR; ++q) {
            const double d = 2.0 * q - (FNEAR - 1);
            tab.push_back(1.0 / tan(M_PI * d / (double)n));
        }
        void *dev = nullptr;
        WFX_HIP(ctx, hipMalloc(&dev, tab.size() * 8));
        WFX_HIP(ctx, hipMemcpyAsync(dev, tab.data(), tab.size() * 8, hipMemcpyHostToDevice, ctx->stream));
        WFX_HIP(ctx, hipStreamSynchronize(ctx->stream));
        if (ctx->fmm_tables.size() >= 16) {
            for (auto &e : ctx->fmm_tables) (void)hipFree((void *)e.second);
            ctx->fmm_tables.clear();
        }
        ctx->fmm_tables.push_back({n, (const double *)dev});
        dt = (const double *)dev;
    }
    P.T.At = dt;
    P.T.Aj = dt + 2 * FP * FP;
    P.T.Cw = dt + 4 * FP * FP;
    P.T.Ca = dt + 5 * FP * FP;
    P.T.G = dt + off_g;
    P.T.gnear = dt + off_n;
    P.g.n = (long long)n;
    P.g.L = L;
    P.g.scale = 2.0 / (double)n;
    P.g.du = ldexp(2.0, L) / (double)n;
    P.L = L;
    const size_t nbox = (size_t)1 << (L + 1);                             // all levels
    WFX_TRY(wfx_reserve(ctx, ctx->b_work, nbox * 2 * FP * 8));           // weights W
    P.nwg = 1u << (L - FW);
    const size_t lg_doubles = ((size_t)1 << (L - FW + 1)) * 2 * FP + 8, cg_doubles = ((size_t)1 << L) * 2 * FP;
    // local expansions down to the leaf workgroups' roots | the leaves' far fields as Chebyshev coefficients | eight envelope values per workgroup
    WFX_TRY(wfx_reserve(ctx, ctx->b_work2, (lg_doubles + cg_doubles + (size_t)P.nwg * 8) * 8 + 64));
    P.Wg = (double *)ctx->b_work.p;
    P.Lg = (double *)ctx->b_work2.p;
    P.Cg = P.Lg + lg_doubles;
    P.Eg = P.Cg + cg_doubles;
    P.smax = (int)((n + ((1ull << L) - 1)) >> L);                         // the largest leaf
    P.xcap = std::min(FXW, ((FLV + 2) * P.smax + 1) & ~1);                // the leaf kernel's sample window: 64 leaves and one more on either side
    P.xc3 = FLV * P.smax;                                                 // the fused notch kernel's
    P.lds_tree = (size_t)(FLV * 2 * FS + (FLV / 2) * 2 * FS + FWALL) * 8;
    P.lds_leaf = (size_t)(FLV * 2 * FS + FNEAR + FXP0 + P.xcap + FXP1) * 8;      // 49 776 bytes for leaves of <= 55 samples: three workgroups per CU
    P.lds_up2 = (size_t)(FLV * 2 * FS + (FLV / 2) * 2 * FS + FLV * 64) * 8;
    P.lds_up3_i16 = (size_t)(FLV * 2 * FS + P.xc3) * 8;
    P.lds_up3_f64 = P.lds_up3_i16 + (size_t)(P.xc3 + 2 * NOTCH_K + 8) * 8;
    const size_t lds_up3_max = (size_t)(FLV * 2 * FS + FLV * 64 + FLV * 64 + 2 * NOTCH_K + 8) * 8;
    static bool attr_done = false;
    if (!attr_done) {
        WFX_HIP(ctx, hipFuncSetAttribute((const void *)fmm_up_leaf2, hipFuncAttributeMaxDynamicSharedMemorySize, (int)P.lds_up2));
        WFX_HIP(ctx, hipFuncSetAttribute((const void *)fmm_up_leaf3<short, UP3W>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_up3_max));
        WFX_HIP(ctx, hipFuncSetAttribute((const void *)fmm_up_leaf3<double, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_up3_max));
        WFX_HIP(ctx, hipFuncSetAttribute((const void *)fmm_tree_leaf, hipFuncAttributeMaxDynamicSharedMemorySize, (int)P.lds_tree));
        WFX_HIP(ctx, hipFuncSetAttribute((const void *)fmm_tree_leaf_env<0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)P.lds_tree));
        WFX_HIP(ctx, hipFuncSetAttribute((const void *)fmm_tree_leaf_env<1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)P.lds_tree));
        WFX_HIP(ctx, hipFuncSetAttribute((const void *)fmm_tree_leaf_env<2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)P.lds_tree));
        attr_done = true;
    }
    // tiers between the leaf workgroups' roots (level L - 6) and the top (levels 2 .. atop <= 5): at most six levels each
    int cur = L - FW;
    P.ntier = 0;
    while (cur > 5) {
        const int D = std::min(FTD, cur - 2);
        P.tier_a[P.ntier] = cur - D;
        P.tier_d[P.ntier] = D;
        ++P.ntier;
        cur -= D;
    }
    P.atop = cur;
    *handled = 1;
    return 0;
}

// notch + P2M + M2M of the leaf workgroups [wg_lo, wg_hi) (x == nullptr: from the raw capture `raw`, index = sample number; else from the
// filtered capture x), all levels' weights to memory
static void fmm_launch_up(wfx_ctx *ctx, const fmm_plan &P, const double *x, const void *raw, int raw_kind, const notch_coef *nc, double *audio,
                          wfx_dev_scalars *clear, unsigned wg_lo, unsigned wg_hi)
{
    const unsigned cnt = wg_hi - wg_lo;
    if (!cnt) return;
    if (!x) {
        if (raw_kind == WFX_IN_I16_MONO)
            hipLaunchKernelGGL((fmm_up_leaf3<short, UP3W>), dim3(cnt), dim3(FTH), P.lds_up3_i16, ctx->stream, (const short *)raw, *nc, audio, P.g, P.T, P.Wg, P.xc3,
                               clear, (int)wg_lo);
        else
            hipLaunchKernelGGL((fmm_up_leaf3<double, 4>), dim3(cnt), dim3(FTH), P.lds_up3_f64, ctx->stream, (const double *)raw, *nc, audio, P.g, P.T, P.Wg, P.xc3,
                               clear, (int)wg_lo);
    } else {
        hipLaunchKernelGGL(fmm_up_leaf2, dim3(cnt), dim3(FTH), P.lds_up2, ctx->stream, x, P.g, P.T, P.Wg);       // (whole capture only)
    }
}

static void fmm_launch_tree_leaf(wfx_ctx *ctx, const fmm_plan &P, const double *x, double *out, int out_mode, unsigned *l0hist, unsigned wg_lo, unsigned wg_hi, int xwrap)
{
    const dim3 grid(wg_hi - wg_lo), block(FTH);
    if (out_mode == 2)
        hipLaunchKernelGGL(fmm_tree_leaf_env<2>, grid, block, P.lds_tree, ctx->stream, x, P.g, P.T, (const double *)P.Wg, (const double *)P.Lg, out, P.smax, P.xcap, P.Eg, l0hist,
                           (int)wg_lo, xwrap);
    else if (out_mode == 1)
        hipLaunchKernelGGL(fmm_tree_leaf_env<1>, grid, block, P.lds_tree, ctx->stream, x, P.g, P.T, (const double *)P.Wg, (const double *)P.Lg, out, P.smax, P.xcap, P.Eg,
                           (unsigned *)nullptr, (int)wg_lo, xwrap);
    else
        hipLaunchKernelGGL(fmm_tree_leaf_env<0>, grid, block, P.lds_tree, ctx->stream, x, P.g, P.T, (const double *)P.Wg, (const double *)P.Lg, out, P.smax, P.xcap, P.Eg,
                           (unsigned *)nullptr, (int)wg_lo, xwrap);
}

// x != nullptr: the filtered capture is given.  x == nullptr: `raw` (int16 or float64 mono) goes through the notch inside the first kernel, which
// leaves the filtered capture in `audio` (wefax.py:63-72 + 174-175 in four leaf-level passes over the capture's bytes)
static int fmm_run(wfx_ctx *ctx, const double *x, const void *raw, int raw_kind, const notch_coef *nc, double *audio, wfx_dev_scalars *clear, uint64_t n, double *out,
                   int out_mode, unsigned *l0hist, int *handled)
{
    fmm_plan P;
    WFX_TRY(fmm_setup(ctx, n, P, handled));
    if (!*handled) return 0;
    *handled = 0;
    const unsigned nwg = P.nwg;
    wfx_prof_begin(ctx, K_FMM_UP);
    fmm_launch_up(ctx, P, x, raw, raw_kind, nc, audio, clear, 0, nwg);
    if (!x) x = audio;
    wfx_prof_end(ctx);
    wfx_prof_begin(ctx, K_FMM_MID);
    for (int k = 0; k < P.ntier; ++k)
        hipLaunchKernelGGL(fmm_up_tier2, dim3(1u << P.tier_a[k]), dim3(FTH), 0, ctx->stream, P.g, P.T, P.Wg, P.tier_a[k], P.tier_d[k], 0);
    hipLaunchKernelGGL(fmm_top2, dim3(1), dim3(FTH), 0, ctx->stream, P.g, P.T, P.Wg, P.Lg, P.atop);
    for (int k = P.ntier - 1; k >= 0; --k)
        hipLaunchKernelGGL(fmm_down_tier2, dim3(1u << P.tier_a[k]), dim3(FTH), 0, ctx->stream, P.g, P.T, (const double *)P.Wg, P.Lg, P.tier_a[k], P.tier_d[k], 0);
    wfx_prof_end(ctx);
    // tree + leaves in one kernel: a workgroup waits through its six tree levels and issues through its near field; the two workgroups of a
    // CU drift apart and fill each other's gaps, and the coefficients never leave LDS
    wfx_prof_begin(ctx, K_FMM_LEAF);
    fmm_launch_tree_leaf(ctx, P, x, out, out_mode, l0hist, 0, nwg, 1);
    if (out_mode == 2 && nwg > 1)
        hipLaunchKernelGGL(fmm_edge_median, dim3((nwg + 254) / 256), dim3(256), 0, ctx->stream, P.g, (const double *)P.Eg, out, l0hist, 0, (int)nwg - 1, 0, (int)nwg);
    wfx_prof_end(ctx);
#ifdef WFX_FMM_STAMPS
    {
        static std::vector<unsigned long long> hb(3 * 4096 * 16);
        (void)hipStreamSynchronize(ctx->stream);
        (void)hipMemcpyFromSymbol(hb.data(), HIP_SYMBOL(fmm_stamp_buf), hb.size() * 8);
        for (int kern = 0; kern < 3; ++kern) {
            const unsigned nb = std::min(nwg, 4096u);
            unsigned long long t0 = ~0ull, t1 = 0;
            double sum[16] = {0};
            const int ns = kern == 0 ? (raw ? 10 : 9) : (kern == 1 ? 9 : (out_mode == 2 ? 6 : 4));
            for (unsigned b2 = 0; b2 < nb; ++b2) {
                const unsigned long long *p = &hb[(size_t)kern * 4096 * 16 + (size_t)b2 * 16];
                t0 = std::min(t0, p[0]);
                t1 = std::max(t1, p[ns - 1]);
                for (int i = 1; i < ns; ++i) sum[i] += (double)(p[i] - p[i - 1]);
            }
            fprintf(stderr, "fmm stamps kernel %d (%u workgroups, 100 MHz ticks -> us): span %.1f us; mean phase us:", kern, nb, (double)(t1 - t0) / 100.0);
            for (int i = 1; i < ns; ++i) fprintf(stderr, " %.2f", sum[i] / nb / 100.0);
            fprintf(stderr, "\n");
        }
    }
#endif
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return wfx_fail_hip(ctx, e, "launch fmm kernels");
    *handled = 1;
    return 0;
}

int wfx_dev_hilbert_fmm(wfx_ctx *ctx, const double *x, uint64_t n, double *out, int out_mode, unsigned *l0hist, int *handled)
{
    return fmm_run(ctx, x, nullptr, 0, nullptr, nullptr, nullptr, n, out, out_mode, l0hist, handled);
}

// a6 + a7 fused: `in` (int16 or float64 mono, n samples) -> notch filtfilt -> audio (float64) -> |audio + i H| -> 5-tap median -> env (+ level-0
// histogram of the select).  *handled = 0 and nothing enqueued when the length has no multipole form or the biquad is not the 49-tap case.
int wfx_dev_notch_hilbert_fmm(wfx_ctx *ctx, const void *in, int in_kind, uint64_t n, const double b[3], const double a[3], const double *ext18, double *audio,
                              double *env, unsigned *l0hist, wfx_dev_scalars *clear, int *handled)
{
    *handled = 0;
    if (in_kind != WFX_IN_I16_MONO && in_kind != WFX_IN_F64_MONO) return 0;
    if (n < NOTCH_SMALL || pow(biquad_pole_radius(a), NOTCH_K) > 1e-16) return 0;
    notch_coef c;
    notch_prepare(c, b, a, ext18);
    return fmm_run(ctx, nullptr, in, in_kind, &c, audio, clear, n, env, 2, l0hist, handled);
}

// ---- the resampler: host side ---------------------------------------------------------------------------------------------------------------
constexpr int RS_POLY_BELOW = 13;            // levels below which cot's x^3 term still counts (x <= 2 pi / 2^L)

struct rs_plan {
    fmm_plan P;                              // tables, levels, tiers and the tree's arrays (shared with the Hilbert transform: one after the other)
    rs_geom rg;
    int swin = 0, grp = 1;
    size_t lds_leaf = 0;
    double *w = nullptr, *cpart = nullptr, *gsum = nullptr, *csum = nullptr;
};

// handled = 0: no multipole form for these lengths (upsampling, odd counts, short captures: the transform route serves them)
static int rs_setup(wfx_ctx *ctx, uint64_t n0, uint64_t num, rs_plan &R, int *handled)
{
    *handled = 0;
    if (num >= n0 || (num & 1) || num < 2 || n0 >= (1ull << 32) || (double)n0 * (double)num >= 4.0e18) return 0;
    const uint64_t nkey = n0 + (n0 & 1);                                  // the tree's depth and tables depend on the level count only
    WFX_TRY(fmm_setup(ctx, nkey, R.P, handled));
    if (!*handled) return 0;
    *handled = 0;
    const int L = R.P.L;
    R.rg.s.n = (long long)n0;
    R.rg.s.L = L;
    R.rg.s.scale = 2.0 / (double)n0;
    R.rg.s.du = ldexp(2.0, L) / (double)n0;
    R.rg.t.n = (long long)num;
    R.rg.t.L = L;
    R.rg.t.scale = 2.0 / (double)num;
    R.rg.t.du = ldexp(2.0, L) / (double)num;
    R.rg.kappa = M_PI / ((double)n0 * (double)num);
    R.rg.inv_n0 = 1.0 / (double)n0;
    R.rg.ratio = (double)num / (double)n0;
    {
        uint64_t ga = n0, gb = num;
        while (gb) {
            const uint64_t tmp = ga % gb;
            ga = gb;
            gb = tmp;
        }
        R.rg.pk = (unsigned)(num / ga);
        R.rg.pn = (unsigned)(n0 / ga);
    }
    const int smax = (int)((n0 + ((1ull << L) - 1)) >> L);
    R.rg.cmin = 3 * (int)(n0 >> L);
    R.rg.cmax = 3 * smax;
    R.swin = (FLV + 2) * smax + 8;
    R.lds_leaf = (size_t)(FLV * FS + FLV * 2 + R.swin) * 8;
    wfx_fmm_shard_geo geo;
    wfx_fmm_shard_geometry(nkey, &geo);
    R.grp = 1 << (geo.ltop - geo.lg);
    const size_t nwg = R.P.nwg;
    WFX_TRY(wfx_reserve(ctx, ctx->b_envraw, ((size_t)n0 + 2 * nwg + 8) * 8));
    R.w = (double *)ctx->b_envraw.p;
    R.cpart = R.w + n0;
    R.gsum = R.cpart + nwg;
    R.csum = R.gsum + nwg;
    static bool attr_done = false;
    if (!attr_done) {
        WFX_HIP(ctx, hipFuncSetAttribute((const void *)rs_up_leaf, hipFuncAttributeMaxDynamicSharedMemorySize, (int)R.P.lds_up2));
        WFX_HIP(ctx, hipFuncSetAttribute((const void *)rs_leaf<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)((FLV * FS + FLV * 2 + (FLV + 2) * 64 + 8) * 8)));
        WFX_HIP(ctx, hipFuncSetAttribute((const void *)rs_leaf<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)((FLV * FS + FLV * 2 + (FLV + 2) * 64 + 8) * 8)));
        attr_done = true;
    }
    *handled = 1;
    return 0;
}

// scipy.signal.resample(x, num) of the n0 real samples x -> y (num samples), wefax.py:160-161, without a transform over the capture
int wfx_dev_resample_fmm(wfx_ctx *ctx, const double *x, uint64_t n0, uint64_t num, double *y, int *handled)
{
    rs_plan R;
    WFX_TRY(rs_setup(ctx, n0, num, R, handled));
    if (!*handled) return 0;
    *handled = 0;
    const fmm_plan &P = R.P;
    const unsigned nwg = P.nwg;
    wfx_prof_begin(ctx, K_RS_UP);
    hipLaunchKernelGGL(rs_up_leaf, dim3(nwg), dim3(FTH), P.lds_up2, ctx->stream, x, R.rg, P.T, P.Wg, R.w, R.cpart, 0, 0ll);
    hipLaunchKernelGGL(rs_csum, dim3(1), dim3(256), 0, ctx->stream, (const double *)R.cpart, (int)nwg, R.grp, R.gsum, R.csum);
    wfx_prof_end(ctx);
    wfx_prof_begin(ctx, K_FMM_MID);
    for (int k = 0; k < P.ntier; ++k)
        hipLaunchKernelGGL(fmm_up_tier2, dim3(1u << P.tier_a[k]), dim3(FTH), 0, ctx->stream, P.g, P.T, P.Wg, P.tier_a[k], P.tier_d[k], 0);
    hipLaunchKernelGGL(fmm_top2, dim3(1), dim3(FTH), 0, ctx->stream, P.g, P.T, P.Wg, P.Lg, P.atop);
    for (int k = P.ntier - 1; k >= 0; --k)
        hipLaunchKernelGGL(fmm_down_tier2, dim3(1u << P.tier_a[k]), dim3(FTH), 0, ctx->stream, P.g, P.T, (const double *)P.Wg, P.Lg, P.tier_a[k], P.tier_d[k], 0);
    wfx_prof_end(ctx);
    wfx_prof_begin(ctx, K_FMM_TREE);
    hipLaunchKernelGGL(fmm_tree_leaf, dim3(nwg), dim3(FTH), P.lds_tree, ctx->stream, P.g, P.T, (const double *)P.Wg, (const double *)P.Lg, P.Cg, 0);
    wfx_prof_end(ctx);
    wfx_prof_begin(ctx, K_RS_LEAF);
    if (P.L < RS_POLY_BELOW)
        hipLaunchKernelGGL(rs_leaf<true>, dim3(nwg), dim3(FTH), R.lds_leaf, ctx->stream, (const double *)R.w, x, R.rg, (const double *)P.Cg, (const double *)R.csum, y, R.swin,
                           0, 1, 0ll, 0ll);
    else
        hipLaunchKernelGGL(rs_leaf<false>, dim3(nwg), dim3(FTH), R.lds_leaf, ctx->stream, (const double *)R.w, x, R.rg, (const double *)P.Cg, (const double *)R.csum, y, R.swin,
                           0, 1, 0ll, 0ll);
    wfx_prof_end(ctx);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return wfx_fail_hip(ctx, e, "launch resampler kernels");
    *handled = 1;
    return 0;
}

// ---- one capture over several GPUs: every rank runs the leaf-level kernels on its own workgroups, the top of the tree is computed by all ----
// (wfx_shard.hip, plan 3).  The tree's geometry depends on n only, never on the world size, and every box is computed by
// the same kernel from the same numbers wherever it is computed: the results are identical for every world size, and to the one-GPU run's.
//   gather level lg = the deepest tier's top (the leaf workgroups' roots when there is no tier): ranks own whole boxes of that level --
//   [gb_lo, gb_hi) -- hence the leaf workgroups [gb_lo, gb_hi) << (L - 6 - lg).  Exchanged: the weights of level lg (all-gather: 2^lg boxes of
//   256 bytes), per finer level the three boxes beyond either end of a rank's range, and four envelope values per seam between ranks.
int wfx_fmm_shard_geometry(uint64_t n, wfx_fmm_shard_geo *geo)
{
    int L = 0;
    if (!fmm_levels(n, &L)) return -1;
    // the top of the first (deepest) tier -- or, where that leaves fewer than 64 boxes, a level inside the tier: the tier is then run in two
    // parts (below the gather level by the ranks, above it by everybody), which computes every level exactly as the one-GPU run does
    int cur = L - FW, lg = cur;
    if (cur > 5) lg = std::min(cur, std::max(cur - std::min(FTD, cur - 2), 6));
    geo->L = L;
    geo->ltop = L - FW;
    geo->lg = lg;
    geo->smax = (int)((n + ((1ull << L) - 1)) >> L);
    return 0;
}

long long wfx_fmm_leaf_first_host(uint64_t n, int L, long long k) { return (k * (long long)n + ((1ll << L) - 1)) >> L; }

// device address of box b of level lev in the weights array (2 x 16 doubles per box); the array is (re)allocated by the first call for this n
int wfx_fmm_shard_weights(wfx_ctx *ctx, uint64_t n, int lev, long long b, double **ptr)
{
    fmm_plan P;
    int handled = 0;
    WFX_TRY(fmm_setup(ctx, n, P, &handled));
    if (!handled) return wfx_fail(ctx, WFX_ERR_BAD_ARG, "fmm: no multipole form for %llu samples", (unsigned long long)n);
    *ptr = P.Wg + fmm_box(lev, b) * 2 * FP;
    return 0;
}

int wfx_fmm_shard_edges(wfx_ctx *ctx, uint64_t n, long long wg, double **ptr)
{
    fmm_plan P;
    int handled = 0;
    WFX_TRY(fmm_setup(ctx, n, P, &handled));
    if (!handled) return wfx_fail(ctx, WFX_ERR_BAD_ARG, "fmm: no multipole form for %llu samples", (unsigned long long)n);
    *ptr = P.Eg + (size_t)wg * 8;
    return 0;
}

// phase A: notch + P2M + M2M of the rank's leaf workgroups, then its own boxes of the deepest tier up to the gather level.  raw_index0 /
// audio_index0: the sample number of raw[0] / audio[0] (the rank's buffers hold its own range and a halo)
int wfx_fmm_shard_up(wfx_ctx *ctx, const void *raw, long long raw_index0, int raw_kind, const double b[3], const double a[3], const double *ext18, double *audio,
                     long long audio_index0, uint64_t n, long long gb_lo, long long gb_hi, wfx_dev_scalars *clear)
{
    fmm_plan P;
    int handled = 0;
    WFX_TRY(fmm_setup(ctx, n, P, &handled));
    if (!handled) return wfx_fail(ctx, WFX_ERR_BAD_ARG, "fmm: no multipole form for %llu samples", (unsigned long long)n);
    if (raw_kind != WFX_IN_I16_MONO && raw_kind != WFX_IN_F64_MONO) return wfx_fail(ctx, WFX_ERR_BAD_ARG, "fmm: input kind %d", raw_kind);
    notch_coef c;
    notch_prepare(c, b, a, ext18);
    wfx_fmm_shard_geo geo;
    wfx_fmm_shard_geometry(n, &geo);
    const int sh = geo.ltop - geo.lg;
    const void *raw0 = raw_kind == WFX_IN_I16_MONO ? (const void *)((const short *)raw - raw_index0) : (const void *)((const double *)raw - raw_index0);
    wfx_prof_begin(ctx, K_FMM_UP);
    fmm_launch_up(ctx, P, nullptr, raw0, raw_kind, &c, audio - audio_index0, clear, (unsigned)(gb_lo << sh), (unsigned)(gb_hi << sh));
    wfx_prof_end(ctx);
    wfx_prof_begin(ctx, K_FMM_MID);
    if (sh > 0 && gb_hi > gb_lo)        // the part of the deepest tier below the gather level, own boxes
        hipLaunchKernelGGL(fmm_up_tier2, dim3((unsigned)(gb_hi - gb_lo)), dim3(FTH), 0, ctx->stream, P.g, P.T, P.Wg, geo.lg, sh, (int)gb_lo);
    wfx_prof_end(ctx);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return wfx_fail_hip(ctx, e, "launch fmm kernels");
    return 0;
}

// everybody: the deepest tier's part above the gather level, the other tiers, the top -- and down again to the gather level
static void fmm_shard_middle(wfx_ctx *ctx, const fmm_plan &P, const wfx_fmm_shard_geo &geo)
{
    const int up0 = P.ntier > 0 ? geo.lg - P.tier_a[0] : 0;
    if (up0 > 0) hipLaunchKernelGGL(fmm_up_tier2, dim3(1u << P.tier_a[0]), dim3(FTH), 0, ctx->stream, P.g, P.T, P.Wg, P.tier_a[0], up0, 0);
    for (int k = 1; k < P.ntier; ++k)
        hipLaunchKernelGGL(fmm_up_tier2, dim3(1u << P.tier_a[k]), dim3(FTH), 0, ctx->stream, P.g, P.T, P.Wg, P.tier_a[k], P.tier_d[k], 0);
    hipLaunchKernelGGL(fmm_top2, dim3(1), dim3(FTH), 0, ctx->stream, P.g, P.T, P.Wg, P.Lg, P.atop);
    for (int k = P.ntier - 1; k >= 1; --k)
        hipLaunchKernelGGL(fmm_down_tier2, dim3(1u << P.tier_a[k]), dim3(FTH), 0, ctx->stream, P.g, P.T, (const double *)P.Wg, P.Lg, P.tier_a[k], P.tier_d[k], 0);
    if (up0 > 0) hipLaunchKernelGGL(fmm_down_tier2, dim3(1u << P.tier_a[0]), dim3(FTH), 0, ctx->stream, P.g, P.T, (const double *)P.Wg, P.Lg, P.tier_a[0], up0, 0);
}

// phase B (the weights of level lg of ALL ranks and the halo boxes of the finer levels have arrived): the top of the tree (every rank the same),
// the rank's part of the deepest tier downwards, tree and leaf kernels of its workgroups.  audio as in phase A, with one filtered leaf in
// front of and behind the rank's own samples; env[0] = the envelope of sample env_index0.  Leaves the medians of all positions but two at
// either end of every workgroup (wfx_fmm_shard_seams) and the level-0 histogram of what it wrote.
int wfx_fmm_shard_down(wfx_ctx *ctx, const double *audio, long long audio_index0, uint64_t n, long long gb_lo, long long gb_hi, double *env, long long env_index0,
                       unsigned *l0hist)
{
    fmm_plan P;
    int handled = 0;
    WFX_TRY(fmm_setup(ctx, n, P, &handled));
    if (!handled) return wfx_fail(ctx, WFX_ERR_BAD_ARG, "fmm: no multipole form for %llu samples", (unsigned long long)n);
    wfx_fmm_shard_geo geo;
    wfx_fmm_shard_geometry(n, &geo);
    const int sh = geo.ltop - geo.lg;
    wfx_prof_begin(ctx, K_FMM_MID);
    fmm_shard_middle(ctx, P, geo);
    const unsigned wg_lo = (unsigned)(gb_lo << sh), wg_hi = (unsigned)(gb_hi << sh);
    if (wg_hi > wg_lo && sh > 0)         // own boxes: from the gather level to the leaf workgroups' roots
        hipLaunchKernelGGL(fmm_down_tier2, dim3((unsigned)(gb_hi - gb_lo)), dim3(FTH), 0, ctx->stream, P.g, P.T, (const double *)P.Wg, P.Lg, geo.lg, sh, (int)gb_lo);
    wfx_prof_end(ctx);
    if (wg_hi > wg_lo) {
        wfx_prof_begin(ctx, K_FMM_LEAF);
        fmm_launch_tree_leaf(ctx, P, audio - audio_index0, env - env_index0, 2, l0hist, wg_lo, wg_hi, 0);
        wfx_prof_end(ctx);
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return wfx_fail_hip(ctx, e, "launch fmm kernels");
    return 0;
}

// phase C (the neighbours' four envelope values at either end of the rank's range have arrived): the medians at the seams of workgroups
int wfx_fmm_shard_seams(wfx_ctx *ctx, uint64_t n, long long gb_lo, long long gb_hi, double *env, long long env_index0, unsigned *l0hist)
{
    fmm_plan P;
    int handled = 0;
    WFX_TRY(fmm_setup(ctx, n, P, &handled));
    if (!handled) return wfx_fail(ctx, WFX_ERR_BAD_ARG, "fmm: no multipole form for %llu samples", (unsigned long long)n);
    wfx_fmm_shard_geo geo;
    wfx_fmm_shard_geometry(n, &geo);
    const int sh = geo.ltop - geo.lg;
    const int wg_lo = (int)(gb_lo << sh), wg_hi = (int)(gb_hi << sh);
    if (wg_hi <= wg_lo) return 0;
    const int b_lo = wg_lo > 0 ? wg_lo - 1 : 0, b_hi = wg_hi < (int)P.nwg ? wg_hi : (int)P.nwg - 1;       // seams b: between workgroups b and b + 1
    wfx_prof_begin(ctx, K_FMM_LEAF);
    if (b_hi > b_lo)
        hipLaunchKernelGGL(fmm_edge_median, dim3((unsigned)(b_hi - b_lo + 255) / 256), dim3(256), 0, ctx->stream, P.g, (const double *)P.Eg, env - env_index0, l0hist,
                           b_lo, b_hi, wg_lo, wg_hi);
    wfx_prof_end(ctx);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return wfx_fail_hip(ctx, e, "launch fmm kernels");
    return 0;
}

// ---- the resampler of a sharded decode (plan 3 with a resampled capture): a rank owns whole boxes [gb_lo, gb_hi) of the RESAMPLER tree's
// gather level -- the same arcs of the circle as its boxes of the Hilbert tree behind it -- hence the sources [first(gb_lo), first(gb_hi)) and
// the targets of the same arcs.  x holds its sources and one leaf (<= 64) beyond either end, round the circle; x[0] = source x_index0.
int wfx_rs_shard_geometry(uint64_t n0, uint64_t num, wfx_fmm_shard_geo *geo)
{
    if (num >= n0 || (num & 1) || num < 2 || n0 >= (1ull << 32) || (double)n0 * (double)num >= 4.0e18) return -1;
    return wfx_fmm_shard_geometry(n0 + (n0 & 1), geo);
}

// weights of the one leaf beyond either end (the near field of the rank's first and last target leaves reads them)
__global__ void __launch_bounds__(128) rs_halo_weights(const double *__restrict__ x, const rs_geom rg, double *__restrict__ w, long long src_lo, long long src_hi,
                                                       long long x_index0)
{
    const int t = threadIdx.x;
    const long long nu = t < 64 ? src_lo - 64 + t : src_hi + (t - 64);         // unwrapped
    const long long n = nu < 0 ? nu + rg.s.n : (nu >= rg.s.n ? nu - rg.s.n : nu);
    double sv, cv;
    rs_sin_cos_r((rg.t.n * n) % (2 * rg.s.n), rg.s.n, rg.inv_n0, &sv, &cv);
    w[n] = x[nu - x_index0] * sv;
}

int wfx_rs_shard_up(wfx_ctx *ctx, const double *x, long long x_index0, uint64_t n0, uint64_t num, long long gb_lo, long long gb_hi, double **gsum)
{
    rs_plan R;
    int handled = 0;
    WFX_TRY(rs_setup(ctx, n0, num, R, &handled));
    if (!handled) return wfx_fail(ctx, WFX_ERR_BAD_ARG, "resampler: no multipole form for %llu -> %llu samples", (unsigned long long)n0, (unsigned long long)num);
    const fmm_plan &P = R.P;
    wfx_fmm_shard_geo geo;
    wfx_fmm_shard_geometry(n0 + (n0 & 1), &geo);
    const int sh = geo.ltop - geo.lg;
    const unsigned wg_lo = (unsigned)(gb_lo << sh), wg_hi = (unsigned)(gb_hi << sh);
    *gsum = R.gsum;
    wfx_prof_begin(ctx, K_RS_UP);
    if (wg_hi > wg_lo) {
        const long long src_lo = wfx_fmm_leaf_first_host(n0, P.L, (long long)wg_lo * FLV), src_hi = wfx_fmm_leaf_first_host(n0, P.L, (long long)wg_hi * FLV);
        hipLaunchKernelGGL(rs_up_leaf, dim3(wg_hi - wg_lo), dim3(FTH), P.lds_up2, ctx->stream, x, R.rg, P.T, P.Wg, R.w, R.cpart, (int)wg_lo, x_index0);
        hipLaunchKernelGGL(rs_halo_weights, dim3(1), dim3(128), 0, ctx->stream, x, R.rg, R.w, src_lo, src_hi, x_index0);
        hipLaunchKernelGGL(rs_csum_own, dim3((unsigned)(gb_hi - gb_lo + 255) / 256), dim3(256), 0, ctx->stream, (const double *)R.cpart, R.grp, R.gsum, (int)gb_lo, (int)gb_hi);
        if (sh > 0) hipLaunchKernelGGL(fmm_up_tier2, dim3((unsigned)(gb_hi - gb_lo)), dim3(FTH), 0, ctx->stream, P.g, P.T, P.Wg, geo.lg, sh, (int)gb_lo);
    }
    wfx_prof_end(ctx);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return wfx_fail_hip(ctx, e, "launch resampler kernels");
    return 0;
}

// (the gather level's weights and group sums of ALL ranks and the halo boxes of the finer levels have arrived)  y[0] = target y_index0
int wfx_rs_shard_down(wfx_ctx *ctx, const double *x, long long x_index0, uint64_t n0, uint64_t num, long long gb_lo, long long gb_hi, double *y, long long y_index0)
{
    rs_plan R;
    int handled = 0;
    WFX_TRY(rs_setup(ctx, n0, num, R, &handled));
    if (!handled) return wfx_fail(ctx, WFX_ERR_BAD_ARG, "resampler: no multipole form for %llu -> %llu samples", (unsigned long long)n0, (unsigned long long)num);
    const fmm_plan &P = R.P;
    wfx_fmm_shard_geo geo;
    wfx_fmm_shard_geometry(n0 + (n0 & 1), &geo);
    const int sh = geo.ltop - geo.lg;
    wfx_prof_begin(ctx, K_FMM_MID);
    hipLaunchKernelGGL(rs_csum2, dim3(1), dim3(256), 0, ctx->stream, (const double *)R.gsum, (int)(P.nwg / (unsigned)R.grp), R.csum);
    fmm_shard_middle(ctx, P, geo);
    const unsigned wg_lo = (unsigned)(gb_lo << sh), wg_hi = (unsigned)(gb_hi << sh);
    if (wg_hi > wg_lo && sh > 0)
        hipLaunchKernelGGL(fmm_down_tier2, dim3((unsigned)(gb_hi - gb_lo)), dim3(FTH), 0, ctx->stream, P.g, P.T, (const double *)P.Wg, P.Lg, geo.lg, sh, (int)gb_lo);
    wfx_prof_end(ctx);
    if (wg_hi > wg_lo) {
        wfx_prof_begin(ctx, K_FMM_TREE);
        hipLaunchKernelGGL(fmm_tree_leaf, dim3(wg_hi - wg_lo), dim3(FTH), P.lds_tree, ctx->stream, P.g, P.T, (const double *)P.Wg, (const double *)P.Lg, P.Cg, (int)wg_lo);
        wfx_prof_end(ctx);
        wfx_prof_begin(ctx, K_RS_LEAF);
        if (P.L < RS_POLY_BELOW)
            hipLaunchKernelGGL(rs_leaf<true>, dim3(wg_hi - wg_lo), dim3(FTH), R.lds_leaf, ctx->stream, (const double *)R.w, x, R.rg, (const double *)P.Cg,
                               (const double *)R.csum, y, R.swin, (int)wg_lo, 1, x_index0, y_index0);
        else
            hipLaunchKernelGGL(rs_leaf<false>, dim3(wg_hi - wg_lo), dim3(FTH), R.lds_leaf, ctx->stream, (const double *)R.w, x, R.rg, (const double *)P.Cg,
                               (const double *)R.csum, y, R.swin, (int)wg_lo, 1, x_index0, y_index0);
        wfx_prof_end(ctx);
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return wfx_fail_hip(ctx, e, "launch resampler kernels");
    return 0;
}

