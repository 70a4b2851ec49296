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

namespace {

constexpr int FP = 16;                      // Chebyshev nodes per box
constexpr int FW = 6;                       // levels inside a leaf workgroup's subtree
constexpr int FLV = 1 << FW;                // leaves per leaf workgroup
constexpr int FTH = 512;                    // its threads: 8 waves x 8 leaves
constexpr int FNEAR = 256;                  // near-field table: odd lags -255 .. 255
constexpr int FROW = 65;                    // padded row of the P2M transposition scratch (doubles)
constexpr int FXW = FLV * 64 + 2 * 64;      // capacity of a leaf workgroup's sample window (its leaves and one more on either side)
constexpr int FTD = 6;                      // deepest tier of the kernels between the leaf workgroups and the top
constexpr int FHB = 3;                      // boxes beyond either end of a subtree that its interaction lists reach

struct fmm_geom {
    long long n;
    int L;                                  // leaves = 2^L, each (k n / 2^L, (k+1) n / 2^L]: <= 64 samples
    double scale;                           // 2 / n
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

// M2M of one level inside LDS: nb parents from 2 nb children ([box][parity][16]); As = [2][16 i][16 j]
template <int NT>
__device__ __forceinline__ void fmm_m2m_level(const double *src, double *dst, const double *As, int nb, int t, double *gout)
{
    for (int it = t; it < nb * 2 * FP; it += NT) {
        const int j = it & 15, hh = (it >> 4) & 1, bb = it >> 5;
        const double *c0 = src + ((2 * bb) * 2 + hh) * FP, *c1 = c0 + 2 * FP;
        double w = 0.0;
#pragma unroll
        for (int i = 0; i < FP; ++i) w = fma(As[i * FP + j], c0[i], fma(As[FP * FP + i * FP + j], c1[i], w));
        dst[it] = w;
        gout[it] = w;
    }
}

// ---- P2M + M2M: a workgroup = 64 consecutive leaves -------------------------------------------------------------------------------
__global__ void __launch_bounds__(FTH, 4) fmm_up_leaf(const double *__restrict__ x, const fmm_geom g, const fmm_tabs T, double *__restrict__ Wg)
{
    extern __shared__ __align__(16) double fl[];
    double *scr = fl;                                   // [8 waves][8 rows][FROW]
    double *pw = scr + 8 * 8 * FROW;                    // [8 waves][8 rows][2][4]
    double *mus = pw + 8 * 64;                          // [8 waves][2][16]
    double *wb0 = mus + 8 * 32;                         // [64][2][16]
    double *wb1 = wb0 + FLV * 2 * FP;                   // [32][2][16]
    double *As = wb1 + (FLV / 2) * 2 * FP;              // [2][16][16] i-major
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, h = lane >> 5, l = lane & 31;
    for (int i = t; i < 2 * FP * FP; i += FTH) As[i] = T.Aj[i];
    double *ws = scr + wave * 8 * FROW, *pww = pw + wave * 64, *mw = mus + wave * 32;
    double cw[FP];                                      // this lane's column of Cw (j = l & 15)
#pragma unroll
    for (int k = 0; k < FP; ++k) cw[k] = T.Cw[k * FP + (l & 15)];
    const long long leaf0 = (long long)blockIdx.x * FLV;
    constexpr int LPW = FLV / 8;                        // leaves per wave
    // the first leaf's sample; the next one's is requested a step ahead
    auto sample_of = [&](long long k, long long &m, bool &valid) {
        const long long a = fmm_leaf_first(g, k), b = fmm_leaf_first(g, k + 1);
        m = a + ((h - a) & 1) + 2 * l;                  // lanes 0-31: even samples of the leaf, 32-63: odd ones
        valid = m < b;
    };
    long long m_cur;
    bool v_cur;
    sample_of(leaf0 + wave * LPW, m_cur, v_cur);
    double x_cur = v_cur ? x[m_cur] : 0.0;
    for (int q = 0; q < LPW; ++q) {
        const int lk = wave * LPW + q;
        const long long k = leaf0 + lk;
        long long m_nxt = 0;
        bool v_nxt = false;
        double x_nxt = 0.0;
        if (q + 1 < LPW) {
            sample_of(k + 1, m_nxt, v_nxt);
            x_nxt = v_nxt ? x[m_nxt] : 0.0;
        }
        const long long r = (m_cur << g.L) - k * g.n;   // position inside the leaf in units of 2^-L samples: [0, n)
        const double u = v_cur ? 2.0 * ((double)r / (double)g.n) - 1.0 : 0.0;
        const double xv = x_cur;
        double tk[FP];
        tk[0] = xv;
        tk[1] = u * xv;
        {
            double t0 = 1.0, t1 = u;
#pragma unroll
            for (int kk = 2; kk < FP; ++kk) {
                const double t2 = fma(2.0 * u, t1, -t0);
                tk[kk] = t2 * xv;
                t0 = t1;
                t1 = t2;
            }
        }
        // moments mu_k = sum over the half's 32 lanes of tk[k], eight k at a time: rows to LDS, (row, quarter) partial sums, four-way join
#pragma unroll
        for (int round = 0; round < 2; ++round) {
#pragma unroll
            for (int kk = 0; kk < 8; ++kk) ws[kk * FROW + lane] = tk[8 * round + kk];
            __builtin_amdgcn_wave_barrier();
            const int row = l & 7, quarter = l >> 3;
            const double *rp = ws + row * FROW + 32 * h + 8 * quarter;
            double part = 0.0;
#pragma unroll
            for (int s2 = 0; s2 < 8; ++s2) part += rp[s2];
            pww[(row * 2 + h) * 4 + quarter] = part;
            __builtin_amdgcn_wave_barrier();
            if (l < 8) {
                const double *pp = pww + (l * 2 + h) * 4;
                mw[h * FP + 8 * round + l] = (pp[0] + pp[1]) + (pp[2] + pp[3]);
            }
            __builtin_amdgcn_wave_barrier();
        }
        if (l < FP) {
            double w = 0.0;
#pragma unroll
            for (int kk = 0; kk < FP; ++kk) w = fma(cw[kk], mw[h * FP + kk], w);
            wb0[(lk * 2 + h) * FP + l] = w;
            Wg[(fmm_box(g.L, k) * 2 + h) * FP + l] = w;
        }
        __builtin_amdgcn_wave_barrier();
        m_cur = m_nxt;
        v_cur = v_nxt;
        x_cur = x_nxt;
    }
    // M2M: six levels inside the subtree, in LDS; every level also goes to memory
    double *src = wb0, *dst = wb1;
    for (int d = 1; d <= FW; ++d) {
        __syncthreads();
        const int nb = FLV >> d;
        fmm_m2m_level<FTH>(src, dst, As, nb, t, Wg + fmm_box(g.L - d, (long long)blockIdx.x * nb) * 2 * FP);
        double *tmp = src;
        src = dst;
        dst = tmp;
    }
}

// ---- a tier between the leaf workgroups and the top, upwards: a workgroup = the subtree of depth D under box `root` of level a --------------
__global__ void __launch_bounds__(256) fmm_up_tier(const fmm_geom g, const fmm_tabs T, double *__restrict__ Wg, int a, int D)
{
    __shared__ double b0[(1 << FTD) * 2 * FP], b1[(1 << (FTD - 1)) * 2 * FP], As[2 * FP * FP];
    const int t = threadIdx.x;
    for (int i = t; i < 2 * FP * FP; i += 256) As[i] = T.Aj[i];
    const double *ch = Wg + fmm_box(a + D, (long long)blockIdx.x << D) * 2 * FP;
    for (int i = t; i < (2 * FP) << D; i += 256) b0[i] = ch[i];
    double *src = b0, *dst = b1;
    for (int d = 1; d <= D; ++d) {
        __syncthreads();
        const int nb = 1 << (D - d);
        fmm_m2m_level<256>(src, dst, As, nb, t, Wg + fmm_box(a + D - d, (long long)blockIdx.x * nb) * 2 * FP);
        double *tmp = src;
        src = dst;
        dst = tmp;
    }
}

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

// weights of boxes b0 - 3 .. b0 + nb + 3 of a level with nbl boxes into LDS, round the circle
template <int NT>
__device__ __forceinline__ void fmm_load_halo(double *wst, const double *Wlev, long long b0, int nb, long long nbl, int t)
{
    for (int i = t; i < (nb + 2 * FHB) * 2 * FP; i += NT) {
        const long long sb = (b0 - FHB + (i >> 5) + nbl) & (nbl - 1);
        wst[i] = Wlev[sb * 2 * FP + (i & 31)];
    }
}

// ---- the top of the tree: levels 2 .. atop (<= 5), one workgroup ------------------------------------------------------------------------
__global__ void __launch_bounds__(256) fmm_top(const fmm_geom g, const fmm_tabs T, double *__restrict__ Wg, double *__restrict__ Lg, int atop)
{
    __shared__ double wall[64 * 2 * FP];                // weights of levels 2 .. atop at their global box index
    __shared__ double wst[(32 + 2 * FHB) * 2 * FP];
    __shared__ double la[32 * 2 * FP], lb[32 * 2 * FP], As[2 * FP * FP], At[2 * FP * FP], Gs[4 * FP * FP];
    const int t = threadIdx.x;
    for (int i = t; i < 2 * FP * FP; i += 256) {
        As[i] = T.Aj[i];
        At[i] = T.At[i];
    }
    const int ntop = 1 << atop;
    for (int i = t; i < ntop * 2 * FP; i += 256) wall[fmm_box(atop, 0) * 2 * FP + i] = Wg[fmm_box(atop, 0) * 2 * FP + i];
    for (int lev = atop - 1; lev >= 2; --lev) {
        __syncthreads();
        fmm_m2m_level<256>(wall + fmm_box(lev + 1, 0) * 2 * FP, wall + fmm_box(lev, 0) * 2 * FP, As, 1 << lev, t, Wg + fmm_box(lev, 0) * 2 * FP);
    }
    double *src = nullptr, *dst = la;
    for (int lev = 2; lev <= atop; ++lev) {
        __syncthreads();
        const int nb = 1 << lev;
        for (int i = t; i < 4 * FP * FP; i += 256) Gs[i] = T.G[(size_t)(lev - 2) * 4 * FP * FP + i];
        for (int i = t; i < (nb + 2 * FHB) * 2 * FP; i += 256) wst[i] = wall[(fmm_box(lev, ((i >> 5) - FHB + nb) & (nb - 1))) * 2 * FP + (i & 31)];
        __syncthreads();
        fmm_down_level<256>(src, dst, wst, Gs, At, 0, nb, nb, t);
        src = dst;
        dst = dst == la ? lb : la;
    }
    __syncthreads();
    for (int i = t; i < ntop * 2 * FP; i += 256) Lg[fmm_box(atop, 0) * 2 * FP + i] = src[i];
}

// ---- a tier downwards: from the local expansion of box `root` of level a to those of its 2^D descendants of level a + D --------------------------
__global__ void __launch_bounds__(256) fmm_down_tier(const fmm_geom g, const fmm_tabs T, const double *__restrict__ Wg, double *__restrict__ Lg, int a, int D)
{
    __shared__ double la[(1 << FTD) * 2 * FP], lb[(1 << (FTD - 1)) * 2 * FP], wst[((1 << FTD) + 2 * FHB) * 2 * FP], At[2 * FP * FP], Gs[4 * FP * FP];
    const int t = threadIdx.x;
    for (int i = t; i < 2 * FP * FP; i += 256) At[i] = T.At[i];
    double *src = (D & 1) ? lb : la, *dst = (D & 1) ? la : lb;          // (the last level ends in `la`)
    if (t < 2 * FP) src[t] = Lg[fmm_box(a, blockIdx.x) * 2 * FP + t];
    for (int d = 1; d <= D; ++d) {
        const int lev = a + d, nb = 1 << d;
        const long long nbl = 1ll << lev, b0 = (long long)blockIdx.x << d;
        __syncthreads();
        for (int i = t; i < 4 * FP * FP; i += 256) Gs[i] = T.G[(size_t)(lev - 2) * 4 * FP * FP + i];
        fmm_load_halo<256>(wst, Wg + fmm_box(lev, 0) * 2 * FP, b0, nb, nbl, t);
        __syncthreads();
        fmm_down_level<256>(src, dst, wst, Gs, At, b0, nb, nbl, t);
        double *tmp = src;
        src = dst;
        dst = tmp;
    }
    __syncthreads();
    double *o = Lg + fmm_box(a + D, (long long)blockIdx.x << D) * 2 * FP;
    for (int i = t; i < (2 * FP) << D; i += 256) o[i] = src[i];
}

// ---- L2L + M2L inside a leaf subtree, then the leaves -----------------------------------------------------------------------------------
template <int OUT_ENV>
__global__ void __launch_bounds__(FTH, 4) fmm_down_leaf(const double *__restrict__ x, const fmm_geom g, const fmm_tabs T, const double *__restrict__ Wg,
                                                        const double *__restrict__ Lg, double *__restrict__ out)
{
    extern __shared__ __align__(16) double fl[];
    double *la = fl;                                    // [64][2][16]
    double *lb = la + FLV * 2 * FP;                     // [32][2][16]
    double *At = lb + (FLV / 2) * 2 * FP;               // [2][16 j][16 i]
    double *gn = At + 2 * FP * FP;                      // [FNEAR]
    double *un = gn + FNEAR;                            // the tree phase: weights of a level + its M2L matrices; the leaf phase: the sample window
    double *wst = un, *Gs = un + (FLV + 2 * FHB) * 2 * FP;
    double *xw = un;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, h = lane >> 5, l = lane & 31;
    const long long leaf0 = (long long)blockIdx.x * FLV, nleaf = 1ll << g.L;
    // the workgroup's samples -- its 64 leaves and one more on either side, round the circle -- are requested now and wait in registers
    // until the tree phase is through with the LDS they go to
    const long long km = leaf0 == 0 ? nleaf - 1 : leaf0 - 1;
    const long long w0 = fmm_leaf_first(g, km) - (leaf0 == 0 ? g.n : 0);                       // may be negative
    const long long kl = leaf0 + FLV == nleaf ? 0 : leaf0 + FLV;                              // the leaf behind the last one
    const long long w1 = (leaf0 + FLV == nleaf ? g.n : 0) + fmm_leaf_first(g, kl + 1);
    const int wlen = (int)(w1 - w0);                                                            // <= FXW
    constexpr int XPT = (FXW + FTH - 1) / FTH;
    double xr[XPT];
#pragma unroll
    for (int q = 0; q < XPT; ++q) {
        const int idx = t + q * FTH;
        long long m = w0 + idx;
        m = m < 0 ? m + g.n : (m >= g.n ? m - g.n : m);
        xr[q] = idx < wlen ? x[m] : 0.0;
    }
    for (int i = t; i < 2 * FP * FP; i += FTH) At[i] = T.At[i];
    for (int i = t; i < FNEAR; i += FTH) gn[i] = T.gnear[i];
    const int ltop = g.L - FW;
    static_assert((FW & 1) == 0, "root in the large buffer: an even number of levels later the leaf level is there again");
    double *src = la, *dst = lb;
    if (t < 2 * FP) src[t] = Lg[fmm_box(ltop, blockIdx.x) * 2 * FP + t];
    for (int d = 1; d <= FW; ++d) {
        const int lev = ltop + d, nb = 1 << d;
        const long long nbl = 1ll << lev, b0 = (long long)blockIdx.x << d;
        __syncthreads();
        for (int i = t; i < 4 * FP * FP; i += FTH) Gs[i] = T.G[(size_t)(lev - 2) * 4 * FP * FP + i];
        fmm_load_halo<FTH>(wst, Wg + fmm_box(lev, 0) * 2 * FP, b0, nb, nbl, t);
        __syncthreads();
        fmm_down_level<FTH>(src, dst, wst, Gs, At, b0, nb, nbl, t);
        double *tmp = src;
        src = dst;
        dst = tmp;
    }
    __syncthreads();                                    // (src == la: the 64 leaves' nodal values)
    // nodal values -> Chebyshev coefficients, in place (one lane per leaf and parity)
    for (int it = t; it < FLV * 2; it += FTH) {
        double *p = la + it * FP;
        double v[FP], c[FP];
#pragma unroll
        for (int j = 0; j < FP; ++j) v[j] = p[j];
#pragma unroll
        for (int k = 0; k < FP; ++k) {
            double s2 = 0.0;
#pragma unroll
            for (int j = 0; j < FP; ++j) s2 = fma(T.Ca[j * FP + k], v[j], s2);
            c[k] = s2;
        }
#pragma unroll
        for (int k = 0; k < FP; ++k) p[k] = c[k];
    }
    // the samples take the place of the tree phase's weights
#pragma unroll
    for (int q = 0; q < XPT; ++q) {
        const int idx = t + q * FTH;
        if (idx < FXW) xw[idx] = xr[q];
    }
    __syncthreads();
    constexpr int LPW = FLV / 8;
    for (int q = 0; q < LPW; ++q) {
        const int lk = wave * LPW + q;
        const long long k = leaf0 + lk;
        // unwrapped sample positions (fmm_leaf_first continues round the circle: leaf -1 starts at a negative position, leaf 2^L at n)
        const long long a = fmm_leaf_first(g, k), b = fmm_leaf_first(g, k + 1), ws0 = fmm_leaf_first(g, k - 1), we = fmm_leaf_first(g, k + 2);
        const long long m = a + ((h - a) & 1) + 2 * l;   // lanes 0-31: even samples of the leaf, 32-63: odd ones
        const bool valid = m < b;
        const long long r = (m << g.L) - k * g.n;
        const double u = valid ? 2.0 * ((double)r / (double)g.n) - 1.0 : 0.0;
        // far field: sum_k a_k T_k(u)
        const double *ac = la + (lk * 2 + h) * FP;
        double t0 = 1.0, t1 = u, far = fma(ac[1], u, ac[0]);
#pragma unroll
        for (int kk = 2; kk < FP; ++kk) {
            const double t2 = fma(2.0 * u, t1, -t0);
            far = fma(ac[kk], t2, far);
            t0 = t1;
            t1 = t2;
        }
        // near field: every sample of the OTHER parity in leaves k - 1 .. k + 1, kernel from the table of odd lags
        const long long ps0 = ws0 + (((1 - h) - ws0) & 1);
        const int cnt = (int)((we - ps0 + 1) >> 1);
        const double *sp = xw + (ps0 - w0);
        int gi = (int)((m - ps0 + (FNEAR - 1)) >> 1);
        double n0 = 0.0, n1 = 0.0;
        int s2 = 0;
        for (; s2 + 1 < cnt; s2 += 2) {
            n0 = fma(gn[gi], sp[2 * s2], n0);
            n1 = fma(gn[gi - 1], sp[2 * s2 + 2], n1);
            gi -= 2;
        }
        if (s2 < cnt) n0 = fma(gn[gi], sp[2 * s2], n0);
        const double H = g.scale * (far + (n0 + n1));
        if (valid) {
            if (OUT_ENV) {
                const double xv = xw[m - w0];
                out[m] = sqrt(fma(xv, xv, H * H));
            } else {
                out[m] = H;
            }
        }
    }
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

// env_raw[i] = |x[i] + i H[i]| (out_env) or H itself, for an even n large enough for the tree; *handled = 0 otherwise (the caller
// runs the transform path)
int wfx_dev_hilbert_fmm(wfx_ctx *ctx, const double *x, uint64_t n, double *out, int out_env, int *handled)
{
    *handled = 0;
    if (n % 2 || n < 32768 || n > (1ull << 32)) return 0;
    int L = 0;
    while (((double)n / (double)(1ull << L)) > 64.0) ++L;                 // leaf size in (32, 64]
    if (L < FW + 2 || L > 26) return 0;                                   // ((sample << L) stays inside 63 bits; at least level 2 above the leaf roots)
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
    fmm_tabs T;
    T.At = dt;
    T.Aj = dt + 2 * FP * FP;
    T.Cw = dt + 4 * FP * FP;
    T.Ca = dt + 5 * FP * FP;
    T.G = dt + off_g;
    T.gnear = dt + off_n;
    fmm_geom g;
    g.n = (long long)n;
    g.L = L;
    g.scale = 2.0 / (double)n;
    const size_t nbox = (size_t)1 << (L + 1);                             // all levels
    WFX_TRY(wfx_reserve(ctx, ctx->b_work, nbox * 2 * FP * 8));           // weights W
    WFX_TRY(wfx_reserve(ctx, ctx->b_work2, ((size_t)1 << (L - FW + 1)) * 2 * FP * 8 + 64));    // local expansions down to the leaf workgroups' roots
    double *Wg = (double *)ctx->b_work.p, *Lg = (double *)ctx->b_work2.p;
    const unsigned nwg = 1u << (L - FW);
    const size_t lds_up = (size_t)(8 * 8 * FROW + 8 * 64 + 8 * 32 + FLV * 2 * FP + (FLV / 2) * 2 * FP + 2 * FP * FP) * 8;
    const size_t lds_dn = (size_t)(FLV * 2 * FP + (FLV / 2) * 2 * FP + 2 * FP * FP + FNEAR + FXW) * 8;
    static_assert(FXW >= (FLV + 2 * FHB) * 2 * FP + 4 * FP * FP, "the sample window is also the tree phase's staging area");
    static bool attr_done = false;
    if (!attr_done) {
        WFX_HIP(ctx, hipFuncSetAttribute((const void *)fmm_up_leaf, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_up));
        WFX_HIP(ctx, hipFuncSetAttribute((const void *)fmm_down_leaf<0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_dn));
        WFX_HIP(ctx, hipFuncSetAttribute((const void *)fmm_down_leaf<1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_dn));
        attr_done = true;
    }
    // tiers between the leaf workgroups' roots (level L - 6) and the top (levels 2 .. atop <= 5): at most six levels each
    int tier_a[8], tier_d[8], ntier = 0, cur = L - FW;
    while (cur > 5) {
        const int D = std::min(FTD, cur - 2);
        tier_a[ntier] = cur - D;
        tier_d[ntier] = D;
        ++ntier;
        cur -= D;
    }
    const int atop = cur;
    wfx_prof_begin(ctx, K_FFT_FWD);
    hipLaunchKernelGGL(fmm_up_leaf, dim3(nwg), dim3(FTH), lds_up, ctx->stream, x, g, T, Wg);
    wfx_prof_end(ctx);
    wfx_prof_begin(ctx, K_BS_CHIRP);
    for (int k = 0; k < ntier; ++k) hipLaunchKernelGGL(fmm_up_tier, dim3(1u << tier_a[k]), dim3(256), 0, ctx->stream, g, T, Wg, tier_a[k], tier_d[k]);
    hipLaunchKernelGGL(fmm_top, dim3(1), dim3(256), 0, ctx->stream, g, T, Wg, Lg, atop);
    for (int k = ntier - 1; k >= 0; --k)
        hipLaunchKernelGGL(fmm_down_tier, dim3(1u << tier_a[k]), dim3(256), 0, ctx->stream, g, T, (const double *)Wg, Lg, tier_a[k], tier_d[k]);
    wfx_prof_end(ctx);
    wfx_prof_begin(ctx, K_FFT_INV);
    if (out_env)
        hipLaunchKernelGGL(fmm_down_leaf<1>, dim3(nwg), dim3(FTH), lds_dn, ctx->stream, x, g, T, (const double *)Wg, (const double *)Lg, out);
    else
        hipLaunchKernelGGL(fmm_down_leaf<0>, dim3(nwg), dim3(FTH), lds_dn, ctx->stream, x, g, T, (const double *)Wg, (const double *)Lg, out);
    wfx_prof_end(ctx);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return wfx_fail_hip(ctx, e, "launch fmm kernels");
    *handled = 1;
    return 0;
}
