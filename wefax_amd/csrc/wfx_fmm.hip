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
constexpr int FW = 7;                       // levels inside a workgroup's subtree
constexpr int FLV = 1 << FW;                // leaves per workgroup
constexpr int FTH = 512;                    // threads: 8 waves x 16 leaves
constexpr int FNEAR = 256;                  // near-field table: odd lags -255 .. 255
constexpr int FROW = 65;                    // padded row of the P2M transposition scratch (doubles)

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

// ---- P2M + M2M ------------------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(FTH, 4) fmm_up_leaf(const double *__restrict__ x, const fmm_geom g, const fmm_tabs T, double *__restrict__ Wg)
{
    extern __shared__ __align__(16) double fl[];
    double *scr = fl;                                   // [8 waves][16][FROW]
    double *As = fl + 8 * FP * FROW;                    // [2][16][16] i-major (M2M)
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, h = lane >> 5, l = lane & 31;
    for (int i = t; i < 2 * FP * FP; i += FTH) As[i] = T.Aj[i];
    double *ws = scr + wave * FP * FROW;
    // this lane's row of Cw (j = l < 16)
    double cw[FP];
#pragma unroll
    for (int k = 0; k < FP; ++k) cw[k] = T.Cw[k * FP + (l & 15)];
    const long long leaf0 = (long long)blockIdx.x * FLV;
    for (int q = 0; q < FLV / 8; ++q) {
        const long long k = leaf0 + wave * (FLV / 8) + q;
        const long long a = fmm_leaf_first(g, k), b = fmm_leaf_first(g, k + 1);
        const long long m = a + ((h - a) & 1) + 2 * l;                   // lanes 0-31: even samples of the leaf, 32-63: odd ones
        const bool valid = m < b;
        const long long r = (m << g.L) - k * g.n;                         // position inside the leaf, in units of 1 / 2^L of a sample: [0, n)
        const double u = valid ? 2.0 * ((double)r / (double)g.n) - 1.0 : 0.0;
        const double xv = valid ? x[m] : 0.0;
        double t0 = 1.0, t1 = u;
        ws[0 * FROW + lane] = xv;
        ws[1 * FROW + lane] = u * xv;
#pragma unroll
        for (int kk = 2; kk < FP; ++kk) {
            const double t2 = fma(2.0 * u, t1, -t0);
            ws[kk * FROW + lane] = t2 * xv;
            t0 = t1;
            t1 = t2;
        }
        __builtin_amdgcn_wave_barrier();
        // lanes l < 16 of each half: moment j = l over the half's 32 samples, then the nodal weights
        double mu = 0.0;
        if (l < FP) {
            const double *row = ws + l * FROW + 32 * h;
#pragma unroll 8
            for (int s = 0; s < 32; ++s) mu += row[s];
        }
        __builtin_amdgcn_wave_barrier();
        if (l < FP) ws[h * FP + l] = mu;                                  // (row 0 is free again)
        __builtin_amdgcn_wave_barrier();
        if (l < FP) {
            double w = 0.0;
#pragma unroll
            for (int kk = 0; kk < FP; ++kk) w = fma(cw[kk], ws[h * FP + kk], w);
            Wg[(fmm_box(g.L, k) * 2 + h) * FP + l] = w;
        }
        __builtin_amdgcn_wave_barrier();
    }
    // M2M: seven levels inside the subtree, children read back from memory (written by this workgroup)
    for (int d = 1; d <= FW; ++d) {
        __threadfence_block();
        __syncthreads();
        const int lev = g.L - d, nb = FLV >> d;
        const long long b0 = (long long)blockIdx.x * nb;
        for (int it = t; it < nb * 2 * FP; it += FTH) {
            const int j = it & 15, hh = (it >> 4) & 1, bb = it >> 5;
            const double *c0 = Wg + (fmm_box(lev + 1, 2 * (b0 + bb)) * 2 + hh) * FP, *c1 = c0 + 2 * FP;
            double w = 0.0;
#pragma unroll
            for (int i = 0; i < FP; ++i) w = fma(As[i * FP + j], c0[i], fma(As[FP * FP + i * FP + j], c1[i], w));
            Wg[(fmm_box(lev, b0 + bb) * 2 + hh) * FP + j] = w;
        }
    }
}

// M2L into one box: target box tb of a level with nb boxes, target parity h hears source parity 1 - h; lanes run over the target node i
__device__ __forceinline__ double fmm_m2l(const double *__restrict__ Wlev, const double *__restrict__ Gl, long long tb, long long nb, int h, int i)
{
    // interaction list: children of the parent's neighbours that do not touch the box.  raw offset r = target - source
    double acc = 0.0;
    const int par = (int)(tb & 1);
    const int offs[3] = {par ? -3 : -2, par ? -2 : 2, par ? 2 : 3};
    const int cnt = nb == 4 ? 1 : 3;                                       // level 2: only the box opposite
#pragma unroll
    for (int s = 0; s < 3; ++s) {
        if (s >= cnt) break;
        const int off = nb == 4 ? 2 : offs[s];
        const long long sb = (tb + off + nb) & (nb - 1);
        const int r = -off;                                               // target - source
        const double *w = Wlev + (sb * 2 + (1 - h)) * FP;
        // r > 0: G_r[i][j], read j-major; r < 0: -G_|r|[j][i]
        const double *gm = Gl + ((size_t)((r > 0 ? r : -r) - 2) * 2 + (r > 0 ? 0 : 1)) * FP * FP;
        double a = 0.0;
#pragma unroll
        for (int j = 0; j < FP; ++j) a = fma(gm[j * FP + i], w[j], a);
        acc += r > 0 ? a : -a;
    }
    return acc;
}

// ---- the levels above the workgroups' subtrees, one workgroup -----------------------------------------------------------------------
__global__ void __launch_bounds__(1024) fmm_top(const fmm_geom g, const fmm_tabs T, double *__restrict__ Wg, double *__restrict__ Lg)
{
    __shared__ double As[2 * FP * FP], At[2 * FP * FP];
    const int t = threadIdx.x;
    for (int i = t; i < 2 * FP * FP; i += 1024) {
        As[i] = T.Aj[i];
        At[i] = T.At[i];
    }
    const int ltop = g.L - FW;                                            // level of the subtree roots
    for (int lev = ltop - 1; lev >= 2; --lev) {                           // M2M
        __threadfence_block();
        __syncthreads();
        const long long nb = 1ll << lev;
        for (long long it = t; it < nb * 2 * FP; it += 1024) {
            const int j = (int)(it & 15), hh = (int)((it >> 4) & 1);
            const long long bb = it >> 5;
            const double *c0 = Wg + (fmm_box(lev + 1, 2 * bb) * 2 + hh) * FP, *c1 = c0 + 2 * FP;
            double w = 0.0;
#pragma unroll
            for (int i = 0; i < FP; ++i) w = fma(As[i * FP + j], c0[i], fma(As[FP * FP + i * FP + j], c1[i], w));
            Wg[(fmm_box(lev, bb) * 2 + hh) * FP + j] = w;
        }
    }
    for (int lev = 2; lev <= ltop; ++lev) {                               // M2L + L2L
        __threadfence_block();
        __syncthreads();
        const long long nb = 1ll << lev;
        const double *Wlev = Wg + fmm_box(lev, 0) * 2 * FP;
        const double *Gl = T.G + (size_t)(lev - 2) * 4 * FP * FP;
        for (long long it = t; it < nb * 2 * FP; it += 1024) {
            const int i = (int)(it & 15), hh = (int)((it >> 4) & 1);
            const long long bb = it >> 5;
            double v = fmm_m2l(Wlev, Gl, bb, nb, hh, i);
            if (lev > 2) {
                const double *lp = Lg + (fmm_box(lev - 1, bb >> 1) * 2 + hh) * FP;
                const double *a = At + (bb & 1) * FP * FP;
#pragma unroll
                for (int j = 0; j < FP; ++j) v = fma(a[j * FP + i], lp[j], v);
            }
            Lg[(fmm_box(lev, bb) * 2 + hh) * FP + i] = v;
        }
    }
}

// ---- L2L + M2L inside a subtree, then the leaves ---------------------------------------------------------------------------------------
template <int OUT_ENV>
__global__ void __launch_bounds__(FTH, 4) fmm_down_leaf(const double *__restrict__ x, const fmm_geom g, const fmm_tabs T, const double *__restrict__ Wg,
                                                        const double *__restrict__ Lg, double *__restrict__ out)
{
    extern __shared__ __align__(16) double fl[];
    double *la = fl;                                    // [128][2][16]  ping
    double *lb = la + FLV * 2 * FP;                     // [64][2][16]   pong (the last level is written to `la`)
    double *At = lb + (FLV / 2) * 2 * FP;               // [2][16 j][16 i]
    double *Gs = At + 2 * FP * FP;                      // [2 r][2][16][16] of the current level
    double *gn = Gs + 4 * FP * FP;                      // [FNEAR]
    double *win = gn + FNEAR;                           // [8 waves][200]
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, h = lane >> 5, l = lane & 31;
    for (int i = t; i < 2 * FP * FP; i += FTH) At[i] = T.At[i];
    for (int i = t; i < FNEAR; i += FTH) gn[i] = T.gnear[i];
    const int ltop = g.L - FW;
    if (t < 2 * FP) lb[t] = Lg[fmm_box(ltop, blockIdx.x) * 2 * FP + t];
    // seven levels down: the current level's values alternate between `lb` and `la`, the leaf level ends in `la`
    for (int d = 1; d <= FW; ++d) {
        const int lev = ltop + d, nb = 1 << d;
        double *src = (d & 1) ? lb : la, *dst = (d & 1) ? la : lb;
        __syncthreads();
        for (int i = t; i < 4 * FP * FP; i += FTH) Gs[i] = T.G[(size_t)(lev - 2) * 4 * FP * FP + i];
        __syncthreads();
        const long long nbl = 1ll << lev, b0 = (long long)blockIdx.x * nb;
        const double *Wlev = Wg + fmm_box(lev, 0) * 2 * FP;
        for (int it = t; it < nb * 2 * FP; it += FTH) {
            const int i = it & 15, hh = (it >> 4) & 1, bb = it >> 5;
            double v = fmm_m2l(Wlev, Gs, b0 + bb, nbl, hh, i);
            const double *lp = src + ((bb >> 1) * 2 + hh) * FP;
            const double *a = At + (bb & 1) * FP * FP;
#pragma unroll
            for (int j = 0; j < FP; ++j) v = fma(a[j * FP + i], lp[j], v);
            dst[(bb * 2 + hh) * FP + i] = v;
        }
    }
    static_assert(FW & 1, "the leaf level must end in the large buffer");
    __syncthreads();
    // nodal values -> Chebyshev coefficients, in place (one lane per (leaf, parity): 16 values in, 16 out)
    for (int it = t; it < FLV * 2; it += FTH) {
        double *p = la + it * FP;
        double v[FP], c[FP];
#pragma unroll
        for (int j = 0; j < FP; ++j) v[j] = p[j];
#pragma unroll
        for (int k = 0; k < FP; ++k) {
            double s = 0.0;
#pragma unroll
            for (int j = 0; j < FP; ++j) s = fma(T.Ca[j * FP + k], v[j], s);
            c[k] = s;
        }
#pragma unroll
        for (int k = 0; k < FP; ++k) p[k] = c[k];
    }
    __syncthreads();
    // the leaves: lanes 0-31 the even samples of a leaf, 32-63 the odd ones
    double *ww = win + wave * 200;
    const long long leaf0 = (long long)blockIdx.x * FLV, nleaf = 1ll << g.L;
    for (int q = 0; q < FLV / 8; ++q) {
        const int lk = wave * (FLV / 8) + q;
        const long long k = leaf0 + lk;
        const long long a = fmm_leaf_first(g, k), b = fmm_leaf_first(g, k + 1);
        // window: leaves k - 1 .. k + 1 on the circle, positions relative to its first sample
        const long long km = k == 0 ? nleaf - 1 : k - 1, kp = k + 1 == nleaf ? 0 : k + 1;
        const long long w0 = fmm_leaf_first(g, km) - (k == 0 ? g.n : 0);            // may be negative
        const long long w1 = (k + 1 == nleaf ? g.n : 0) + fmm_leaf_first(g, kp + 1);   // one past the window
        const int wlen = (int)(w1 - w0);                                              // <= 192
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int s = 0; s < 200; s += 64) {
            const int idx = s + lane;
            if (idx < 200) {
                long long m = w0 + idx;
                m = m < 0 ? m + g.n : (m >= g.n ? m - g.n : m);
                ww[idx] = idx < wlen ? x[m] : 0.0;
            }
        }
        __builtin_amdgcn_wave_barrier();
        const long long m = a + ((h - a) & 1) + 2 * l;
        const bool valid = m < b;
        const long long r = (m << g.L) - k * g.n;
        const double u = valid ? 2.0 * ((double)r / (double)g.n) - 1.0 : 0.0;
        const int rel = (int)(m - w0);                                                 // position of the target in the window
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
        // near field: every sample of the other parity in the window (absolute parity: (w0 + p) & 1 == 1 - h)
        const int p0 = (int)(((1 - h) - w0) & 1);                                     // first window position of the source parity
        int gi = (rel - p0 + (FNEAR - 1)) >> 1;                                       // table index of lag rel - p0 (odd)
        double n0 = 0.0, n1 = 0.0;
        const double *sp = ww + p0;
#pragma unroll 4
        for (int s = 0; s < 96; s += 2) {
            n0 = fma(gn[gi & (FNEAR - 1)], sp[2 * s], n0);
            n1 = fma(gn[(gi - 1) & (FNEAR - 1)], sp[2 * s + 2], n1);
            gi -= 2;
        }
        const double H = g.scale * (far + (n0 + n1));
        if (valid) {
            if (OUT_ENV) {
                const double xv = ww[rel];
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
    if (n % 2 || n < (64ull << (FW + 2)) || n > (1ull << 40)) return 0;
    int L = 0;
    while (((double)n / (double)(1ull << L)) > 64.0) ++L;                 // leaf size in (32, 64]
    if (L < FW + 2 || L > 26) return 0;                                   // ((sample << L) stays inside 63 bits)
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
    WFX_TRY(wfx_reserve(ctx, ctx->b_work2, ((size_t)1 << (L - FW + 1)) * 2 * FP * 8 + 64));    // local expansions of the top levels
    double *Wg = (double *)ctx->b_work.p, *Lg = (double *)ctx->b_work2.p;
    const unsigned nwg = 1u << (L - FW);
    const size_t lds_up = (size_t)(8 * FP * FROW + 2 * FP * FP) * 8;
    const size_t lds_dn = (size_t)(FLV * 2 * FP + (FLV / 2) * 2 * FP + 2 * FP * FP + 4 * FP * FP + FNEAR + 8 * 200) * 8;
    static bool attr_done = false;
    if (!attr_done) {
        WFX_HIP(ctx, hipFuncSetAttribute((const void *)fmm_up_leaf, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_up));
        WFX_HIP(ctx, hipFuncSetAttribute((const void *)fmm_down_leaf<0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_dn));
        WFX_HIP(ctx, hipFuncSetAttribute((const void *)fmm_down_leaf<1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_dn));
        attr_done = true;
    }
    wfx_prof_begin(ctx, K_FFT_FWD);
    hipLaunchKernelGGL(fmm_up_leaf, dim3(nwg), dim3(FTH), lds_up, ctx->stream, x, g, T, Wg);
    wfx_prof_end(ctx);
    wfx_prof_begin(ctx, K_BS_CHIRP);
    hipLaunchKernelGGL(fmm_top, dim3(1), dim3(1024), 0, ctx->stream, g, T, Wg, Lg);
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
