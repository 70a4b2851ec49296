// Mixed-radix cyclic transform for the analytic-signal convolution (exact mode).
//
// For even N the Hilbert transform is ONE cyclic convolution of the packed sequence
// z[q] = x[2q+1] + i x[2q] (length L = N/2) with a real kernel whose L-point spectrum
// has a closed form:
//       G[0] = 0,   G[k] = -i * exp(-2 pi i k / N)   (0 < k < L)
// (derivation in DESIGN.md 3.2: the N-point spectrum of the Hilbert kernel is -i*sgn,
// and g = its odd polyphase component).  When L factors into primes <= 13 the
// convolution is done with L-point transforms directly -- no zero padding to a power of
// two (2.3x fewer points for the 10-minute capture), no stored kernel spectrum:
//
//       V = IDFT_L( G .* DFT_L(z) ),   Re V[p] = H[2p],  Im V[p] = H[2p-1]
//
// Engine: Stockham autosort passes, natural order in and out, out of place between two
// buffers.  A pass with radix R (a product of small primes, <= 256) and P = product of
// the previous radices: thread-column j in [0, L/R), k = j mod P,
//       in_m  = x[j + m L/R] * W_{P R}^{k m},   X = DFT_R(in),   y[(j-k) R + k + q P] = X[q].
// A workgroup owns T = 4096/R columns: loads are T-element runs, the R-point transform
// runs in LDS as in-place decimation-in-frequency stages over the prime factors (roots
// from a per-block LDS table, p-point butterflies in registers), the digit-reversed
// result is read out in natural order.  Input twiddles come from a two-level table
// (W^t = hi[t >> 11] * lo[t & 2047]).  Like the power-of-two engine the passes are
// HBM-bound: read 16 B + write 16 B per point and pass.
#include <algorithm>
#include <cstdlib>
#include <mutex>
#include <utility>

#include "wfx_internal.h"

#define MR_TILE 4096
#define MR_MAXPASS 5
#define MR_LO_BITS 11
#define MR_LO (1 << MR_LO_BITS)

__device__ __forceinline__ cplx mcmul(cplx a, cplx b)
{
    return make_double2(fma(a.x, b.x, -(a.y * b.y)), fma(a.x, b.y, a.y * b.x));
}
__device__ __forceinline__ cplx mconj_if(cplx a, bool inv) { return inv ? make_double2(a.x, -a.y) : a; }

// cos / sin of 2 pi i / P as compile-time constants (immediates: no registers, no LDS reads)
template <int P>
struct mr_roots {
    static constexpr double c[1] = {1.0};
    static constexpr double s[1] = {0.0};
};
template <> struct mr_roots<3> {
    static constexpr double c[3] = {1.0, -0.4999999999999998, -0.5000000000000004};
    static constexpr double s[3] = {0.0, 0.8660254037844387, -0.8660254037844384};
};
template <> struct mr_roots<5> {
    static constexpr double c[5] = {1.0, 0.30901699437494745, -0.8090169943749473, -0.8090169943749476, 0.30901699437494723};
    static constexpr double s[5] = {0.0, 0.9510565162951535, 0.5877852522924732, -0.587785252292473, -0.9510565162951536};
};
template <> struct mr_roots<7> {
    static constexpr double c[7] = {1.0, 0.6234898018587336, -0.22252093395631434, -0.900968867902419, -0.9009688679024191, -0.2225209339563146, 0.6234898018587334};
    static constexpr double s[7] = {0.0, 0.7818314824680298, 0.9749279121818236, 0.43388373911755823, -0.433883739117558, -0.9749279121818236, -0.7818314824680299};
};
template <> struct mr_roots<11> {
    static constexpr double c[11] = {1.0, 0.8412535328311812, 0.41541501300188644, -0.142314838273285, -0.654860733945285, -0.9594929736144974, -0.9594929736144975, -0.6548607339452852, -0.14231483827328523, 0.41541501300188605, 0.8412535328311812};
    static constexpr double s[11] = {0.0, 0.5406408174555976, 0.9096319953545183, 0.9898214418809328, 0.7557495743542583, 0.28173255684142967, -0.2817325568414294, -0.7557495743542582, -0.9898214418809327, -0.9096319953545186, -0.5406408174555974};
};
template <> struct mr_roots<13> {
    static constexpr double c[13] = {1.0, 0.8854560256532099, 0.5680647467311559, 0.120536680255323, -0.35460488704253545, -0.7485107481711012, -0.970941817426052, -0.9709418174260521, -0.7485107481711013, -0.3546048870425359, 0.1205366802553232, 0.5680647467311548, 0.88545602565321};
    static constexpr double s[13] = {0.0, 0.4647231720437685, 0.8229838658936564, 0.992708874098054, 0.9350162426854148, 0.6631226582407952, 0.23931566428755768, -0.23931566428755743, -0.663122658240795, -0.9350162426854147, -0.992708874098054, -0.822983865893657, -0.4647231720437684};
};

template <> struct mr_roots<9> {
    static constexpr double c[9] = {1.0, 0.766044443118978, 0.17364817766693041, -0.4999999999999998, -0.9396926207859083, -0.9396926207859084, -0.5000000000000004, 0.17364817766692997, 0.7660444431189778};
    static constexpr double s[9] = {0.0, 0.6427876096865393, 0.984807753012208, 0.8660254037844387, 0.3420201433256689, -0.34202014332566866, -0.8660254037844384, -0.9848077530122081, -0.6427876096865396};
};
template <> struct mr_roots<8> {
    static constexpr double c[8] = {1.0, 0.7071067811865476, 0.0, -0.7071067811865475, -1.0, -0.7071067811865477, 0.0, 0.7071067811865474};
    static constexpr double s[8] = {0.0, 0.7071067811865475, 1.0, 0.7071067811865476, 0.0, -0.7071067811865475, -1.0, -0.7071067811865477};
};
template <> struct mr_roots<16> {
    static constexpr double c[16] = {1.0, 0.9238795325112867, 0.7071067811865476, 0.38268343236508984, 0.0, -0.3826834323650897, -0.7071067811865475, -0.9238795325112867, -1.0, -0.9238795325112868, -0.7071067811865477, -0.38268343236509034, 0.0, 0.38268343236509, 0.7071067811865474, 0.9238795325112865};
    static constexpr double s[16] = {0.0, 0.3826834323650898, 0.7071067811865475, 0.9238795325112867, 1.0, 0.9238795325112867, 0.7071067811865476, 0.3826834323650899, 0.0, -0.38268343236508967, -0.7071067811865475, -0.9238795325112865, -1.0, -0.9238795325112866, -0.7071067811865477, -0.3826834323650904};
};
// composite odd radices of the register-resident two-level passes (the conjugate-pair form holds for any odd P)
template <> struct mr_roots<15> {
    static constexpr double c[15] = {1.0, 0.9135454576426009, 0.6691306063588582, 0.30901699437494745, -0.10452846326765333, -0.4999999999999998, -0.8090169943749473, -0.9781476007338057, -0.9781476007338057, -0.8090169943749476, -0.5000000000000004, -0.10452846326765423, 0.30901699437494723, 0.6691306063588585, 0.913545457642601};
    static constexpr double s[15] = {0.0, 0.40673664307580015, 0.7431448254773941, 0.9510565162951535, 0.9945218953682734, 0.8660254037844387, 0.5877852522924732, 0.20791169081775931, -0.20791169081775907, -0.587785252292473, -0.8660254037844384, -0.9945218953682733, -0.9510565162951536, -0.743144825477394, -0.40673664307580015};
};
template <> struct mr_roots<25> {
    static constexpr double c[25] = {1.0, 0.9685831611286311, 0.8763066800438636, 0.7289686274214116, 0.5358267949789965, 0.30901699437494745, 0.06279051952931353, -0.1873813145857246, -0.4257792915650727, -0.6374239897486897, -0.8090169943749473, -0.9297764858882513, -0.9921147013144778, -0.9921147013144779, -0.9297764858882515, -0.8090169943749478, -0.6374239897486895, -0.42577929156507216, -0.18738131458572463, 0.06279051952931283, 0.30901699437494723, 0.5358267949789968, 0.7289686274214112, 0.8763066800438631, 0.968583161128631};
    static constexpr double s[25] = {0.0, 0.2486898871648548, 0.4817536741017153, 0.6845471059286886, 0.8443279255020151, 0.9510565162951535, 0.9980267284282716, 0.9822872507286887, 0.9048270524660195, 0.7705132427757893, 0.5877852522924732, 0.36812455268467814, 0.12533323356430454, -0.12533323356430429, -0.3681245526846779, -0.5877852522924727, -0.7705132427757894, -0.9048270524660198, -0.9822872507286887, -0.9980267284282716, -0.9510565162951536, -0.844327925502015, -0.684547105928689, -0.4817536741017161, -0.24868988716485535};
};

// p-point DFT of v[0..P) in registers; sg = -1 forward (W = e^{-i..}), +1 inverse.  Each output is
// handed to `put(q, value)` as soon as it is complete (keeps the register footprint at v + a + b).
// Odd primes use the conjugate-pair form: with a_m = v_m + v_{P-m}, b_m = v_m - v_{P-m},
//   y[q], y[P-q] = v_0 + sum_m a_m cos(2 pi m q / P)  +-  i sg sum_m b_m sin(2 pi m q / P)
// i.e. (P-1)^2 real FMAs instead of 4 (P-1)^2.
template <int P, typename PUT>
__device__ __forceinline__ void dft_small(const cplx *v, const double sg, PUT put)
{
    if (P == 2) {
        put(0, make_double2(v[0].x + v[1].x, v[0].y + v[1].y));
        put(1, make_double2(v[0].x - v[1].x, v[0].y - v[1].y));
        return;
    }
    if (P == 4) {
        const cplx t0 = make_double2(v[0].x + v[2].x, v[0].y + v[2].y), t1 = make_double2(v[0].x - v[2].x, v[0].y - v[2].y);
        const cplx t2 = make_double2(v[1].x + v[3].x, v[1].y + v[3].y), d = make_double2(v[1].x - v[3].x, v[1].y - v[3].y);
        const cplx t3 = make_double2(-sg * d.y, sg * d.x);     // d * (i sg)
        put(0, make_double2(t0.x + t2.x, t0.y + t2.y));
        put(2, make_double2(t0.x - t2.x, t0.y - t2.y));
        put(1, make_double2(t1.x + t3.x, t1.y + t3.y));
        put(3, make_double2(t1.x - t3.x, t1.y - t3.y));
        return;
    }
    constexpr int H = (P - 1) / 2;
    cplx a[H > 0 ? H : 1], b[H > 0 ? H : 1];
    cplx y0 = v[0];
#pragma unroll
    for (int m = 1; m <= H; ++m) {
        a[m - 1] = make_double2(v[m].x + v[P - m].x, v[m].y + v[P - m].y);
        b[m - 1] = make_double2(v[m].x - v[P - m].x, v[m].y - v[P - m].y);
        y0.x += a[m - 1].x;
        y0.y += a[m - 1].y;
    }
    put(0, y0);
#pragma unroll
    for (int q = 1; q <= H; ++q) {
        double ax = v[0].x, ay = v[0].y, bx = 0.0, by = 0.0;
#pragma unroll
        for (int m = 1; m <= H; ++m) {
            const double wc = mr_roots<P>::c[(m * q) % P], ws = sg * mr_roots<P>::s[(m * q) % P];
            ax = fma(a[m - 1].x, wc, ax);
            ay = fma(a[m - 1].y, wc, ay);
            bx = fma(b[m - 1].x, ws, bx);
            by = fma(b[m - 1].y, ws, by);
        }
        // v_m w + v_{P-m} conj(w) = a Re w + i b Im w ;  i (bx + i by) = (-by, bx)
        put(q, make_double2(ax - by, ay + bx));
        put(P - q, make_double2(ax + by, ay - bx));
    }
}

// one in-place DIF stage of radix P over the LDS tile [R][T]
template <int P>
__device__ __forceinline__ void mr_stage(cplx *tile, const cplx *wr, int R, int log2t, int Ls, bool inv)
{
    const int sub = Ls / P;
    const int nb = (R / P) << log2t;
    const int tstep = R / Ls;
    const int tmask = (1 << log2t) - 1;
    const double sg = inv ? 1.0 : -1.0;
    for (int b = threadIdx.x; b < nb; b += 256) {
        const int c = b & tmask, bi = b >> log2t;
        const int gi = bi / sub, j = bi - gi * sub;
        const int row0 = gi * Ls + j;
        cplx v[P];
#pragma unroll
        for (int m = 0; m < P; ++m) v[m] = tile[((row0 + m * sub) << log2t) + c];
        const int jt = j * tstep;              // j q tstep < sub P tstep = R: no reduction needed
        // every thread owns its P positions: they can be overwritten as the outputs complete
        dft_small<P>(v, sg, [&](int q, cplx y) {
            if (q > 0) y = mcmul(y, mconj_if(wr[jt * q], inv));
            tile[((row0 + q * sub) << log2t) + c] = y;
        });
    }
}

// workgroup barrier that orders LDS traffic only: the register prefetch of the next tile stays
// in flight across it (a __syncthreads() would drain it with s_waitcnt vmcnt(0))
__device__ __forceinline__ void mr_lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// Twiddle index of the column whose layout index is kl = j mod P.  Plain transform: kl itself.  Distributed transform
// (wfx_dist.hip): the array is a slab [.][B] whose innermost index kk stands for the first-pass output k1 = map(kk) of the
// global plan, or a range of columns starting at global column kb0:  k = map(kl mod B) * kscale + (kl / B) * kstep.
__device__ __forceinline__ int mr_twiddle_k(const mr_pass_desc &d, int kl)
{
    if (!d.dist) return kl;
    const int kk = kl % d.B, kq = kl / d.B;
    return (kk < d.kc0 ? d.kb0 + kk : d.kb1 + (kk - d.kc0)) * d.kscale + kq * d.kstep;
}
// global frequency index of layout index o (last forward pass of a slab: the Hilbert spectrum is a function of it)
__device__ __forceinline__ long long mr_global_index(const mr_pass_desc &d, long long o)
{
    if (!d.dist) return o;
    const unsigned q = (unsigned)o / (unsigned)d.B;                    // (a rank's part of a transform has fewer than 2^31 points)
    const int kk = (int)((unsigned)o - q * (unsigned)d.B);
    return (long long)(kk < d.kc0 ? d.kb0 + kk : d.kb1 + (kk - d.kc0)) + (long long)q * (long long)d.kstep;
}

// IN_MODE 0: complex input array; 1: packed real input z[q] = x[2q+1] + i x[2q];
// 2: int16 samples read as z[q] = x[2q] + i x[2q+1] (the resampler's first pass on an int16 capture: no f64 copy)
// OUT_MODE 0: plain; 1: multiply by the Hilbert spectrum G[k] / L (last forward pass)
// Persistent workgroups: each walks tiles blockIdx.x, blockIdx.x + gridDim.x, ... and fetches the
// next tile into registers while the current one is transformed, so HBM stays busy during the
// LDS stages.
template <int IN_MODE, int OUT_MODE, int INVERSE>
__global__ void __launch_bounds__(256, 2)
mr_pass(const cplx *__restrict__ in, cplx *__restrict__ out, mr_pass_desc d, const cplx *__restrict__ tw_lo, const cplx *__restrict__ tw_hi,
        int ntiles)
{
    constexpr int NPRE = MR_TILE / 256;
    __shared__ cplx tile[MR_TILE];
    __shared__ cplx wr[256];
    __shared__ unsigned short rev[256];       // rev[q] = position of X[q] after the in-place DIF stages
    constexpr bool inv = INVERSE != 0;
    const int R = d.R, T = d.T, log2t = d.log2t, tmask = T - 1;
    const int nelem = R << log2t;
    // all index arithmetic is 32-bit: L < 2^31
    const int ncol = (int)d.ncol, P = (int)d.P;
    if ((int)threadIdx.x < R) {
        double s, c;
        sincospi(2.0 * (double)threadIdx.x / (double)R, &s, &c);
        wr[threadIdx.x] = make_double2(c, -s);                    // W_R^t (forward)
        int u = 0, rem = (int)threadIdx.x, span = R;
        for (int st = 0; st < d.nf; ++st) {
            const int p = d.f[st];
            span /= p;
            u += (rem % p) * span;
            rem /= p;
        }
        rev[threadIdx.x] = (unsigned short)u;
    }
    // 256 is a multiple of T: a thread keeps its column c and visits rows m0, m0 + ms, ...
    const int c = (int)threadIdx.x & tmask;
    const int m0 = (int)threadIdx.x >> log2t, ms = 256 >> log2t;
    cplx pre[NPRE];
    auto prefetch = [&](int tix) {
        const int j = tix * T + c;
        const bool col_ok = j < ncol;
#pragma unroll
        for (int i = 0; i < NPRE; ++i) {
            const int m = m0 + i * ms;
            cplx v = make_double2(0.0, 0.0);
            if (col_ok && m < R) {
                const long long a = (long long)j + (long long)m * ncol;
                if (IN_MODE == 1) {
                    const cplx x2 = in[a];
                    v = make_double2(x2.y, x2.x);
                } else if (IN_MODE == 2) {
                    const short2 x2 = ((const short2 *)in)[a];
                    v = make_double2((double)x2.x, (double)x2.y);
                } else {
                    v = in[a];
                }
            }
            pre[i] = v;
        }
    };
    const double inv_l = 1.0 / (double)d.Ltw;
    int tix = blockIdx.x;
    if (tix < ntiles) prefetch(tix);
    for (; tix < ntiles; tix += gridDim.x) {
        const int j0 = tix * T;
        const int tn = min(T, ncol - j0);                         // columns in this tile
        mr_lds_barrier();                                         // the previous tile has been read out
        // 1. registers -> LDS with the input twiddle W_{P R}^{k m} (recurrence along m from two table look-ups)
        {
            const int j = j0 + c;
            cplx w = make_double2(1.0, 0.0), wstep = make_double2(1.0, 0.0);
            if (P > 1 && c < tn) {
                const int k = mr_twiddle_k(d, j % P);
                const int t0 = k * m0, ts = k * ms;               // < P R <= L
                w = mconj_if(mcmul(tw_hi[t0 >> MR_LO_BITS], tw_lo[t0 & (MR_LO - 1)]), inv);
                wstep = mconj_if(mcmul(tw_hi[ts >> MR_LO_BITS], tw_lo[ts & (MR_LO - 1)]), inv);
            }
#pragma unroll
            for (int i = 0; i < NPRE; ++i) {
                const int e = (int)threadIdx.x + 256 * i;
                if (e < nelem) {
                    cplx v = pre[i];
                    if (P > 1) {
                        v = mcmul(v, w);
                        w = mcmul(w, wstep);
                    }
                    tile[e] = v;                                  // e == m * T + c
                }
            }
        }
        mr_lds_barrier();
        if (tix + (int)gridDim.x < ntiles) prefetch(tix + gridDim.x);      // in flight during the stages below
        // 2. R-point transform of every column: in-place DIF stages over the factors
        int Ls = R;
        for (int st = 0; st < d.nf; ++st) {
            const int p = d.f[st];
            switch (p) {
            case 2: mr_stage<2>(tile, wr, R, log2t, Ls, inv); break;
            case 3: mr_stage<3>(tile, wr, R, log2t, Ls, inv); break;
            case 4: mr_stage<4>(tile, wr, R, log2t, Ls, inv); break;
            case 5: mr_stage<5>(tile, wr, R, log2t, Ls, inv); break;
            case 7: mr_stage<7>(tile, wr, R, log2t, Ls, inv); break;
            case 11: mr_stage<11>(tile, wr, R, log2t, Ls, inv); break;
            case 13: mr_stage<13>(tile, wr, R, log2t, Ls, inv); break;
            default: break;
            }
            Ls /= p;
            mr_lds_barrier();
        }
        // 3. natural-order read-out: X[q] sits at position rev[q]
        auto emit = [&](int q, int cc) {
            cplx v = tile[((int)rev[q] << log2t) + cc];
            const int j = j0 + cc;
            const int k = P > 1 ? j % P : 0;
            const long long o = (long long)(j - k) * R + k + (long long)q * P;
            if (d.skip_hi) {
                const long long os = mr_global_index(d, o);
                if (os > d.skip_lo && os < d.skip_hi) return;
            }
            if (OUT_MODE == 1) {
                // G[o] / L,  G[k] = -i exp(-i pi k / L) = (-sin, -cos)(pi k / L), G[0] = 0  (o: the GLOBAL frequency index)
                const long long og = mr_global_index(d, o);
                double sn, cs;
                sincospi((double)og / (double)d.Ltw, &sn, &cs);
                const cplx g = og == 0 ? make_double2(0.0, 0.0) : make_double2(-sn * inv_l, -cs * inv_l);
                v = mcmul(v, g);
            }
            out[o] = v;
        };
        if (P == 1) {
            // the tile's outputs are one contiguous run [j0 R, (j0 + tn) R): walk it in output order
            int q = (int)threadIdx.x % R, cc = (int)threadIdx.x / R;
            const int dq = 256 % R, dc = 256 / R;
            for (int e = threadIdx.x; e < R * tn; e += 256) {
                emit(q, cc);
                q += dq;
                cc += dc;
                if (q >= R) {
                    q -= R;
                    ++cc;
                }
            }
        } else if (c < tn) {
            // for a fixed q the columns of the tile are consecutive outputs
            for (int e = threadIdx.x; e < nelem; e += 256) emit(e >> log2t, c);
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Two-level register-resident pass for R = RA * RB (same index maps as mr_pass, so the two kinds
// of pass can be mixed in one plan).  The per-prime LDS stages of mr_pass cost ~250 vector
// instructions per point (index arithmetic, a twiddle per stage, five LDS round trips); here a
// thread loads the RA rows {a RB + b} of its column straight from global memory into registers,
// transforms them with compile-time indices, applies ONE twiddle W_R^{b qa}, and a second thread
// picks the RB values {qa RB + b} up from LDS for the second transform and stores the results:
// one LDS round trip, two barriers per tile, ~50 vector instructions per point.
//   level A items (c, b):  T * RB, RA points each      level B items (c, qa):  T * RA, RB points each
// The next tile's loads are issued before level B and stay in flight through it.
// ---------------------------------------------------------------------------------------------
// Q0..Q1: the output pairs (q, P - q) this call produces (1..H = all); EMIT0: also output 0.  Two threads can share one
// transform by splitting the pairs (each prepares the sums and differences for itself).
template <int P, int Q0, int Q1, bool EMIT0, typename PUT>
__device__ __forceinline__ void dft_odd_part(cplx *v, const double sg, PUT put)
{
    constexpr int H = (P - 1) / 2;
    cplx y0 = v[0];
#pragma unroll
    for (int m = 1; m <= H; ++m) {                 // a_m -> v[m], b_m -> v[P - m]
        const cplx am = make_double2(v[m].x + v[P - m].x, v[m].y + v[P - m].y);
        const cplx bm = make_double2(v[m].x - v[P - m].x, v[m].y - v[P - m].y);
        v[m] = am;
        v[P - m] = bm;
        y0.x += am.x;
        y0.y += am.y;
    }
    const cplx v0 = v[0];
    if (EMIT0) put(0, y0);
#pragma unroll
    for (int q = Q0; q <= Q1; ++q) {
        double ax = v0.x, ay = v0.y, bx = 0.0, by = 0.0;
#pragma unroll
        for (int m = 1; m <= H; ++m) {
            const double wc = mr_roots<P>::c[(m * q) % P], ws = sg * mr_roots<P>::s[(m * q) % P];
            ax = fma(v[m].x, wc, ax);
            ay = fma(v[m].y, wc, ay);
            bx = fma(v[P - m].x, ws, bx);
            by = fma(v[P - m].y, ws, by);
        }
        put(q, make_double2(ax - by, ay + bx));
        put(P - q, make_double2(ax + by, ay - bx));
    }
}

template <int P, typename PUT>
__device__ __forceinline__ void dft_odd_inplace(cplx *v, const double sg, PUT put)
{
    constexpr int H = (P - 1) / 2;
    cplx y0 = v[0];
#pragma unroll
    for (int m = 1; m <= H; ++m) {                 // a_m -> v[m], b_m -> v[P - m]
        const cplx am = make_double2(v[m].x + v[P - m].x, v[m].y + v[P - m].y);
        const cplx bm = make_double2(v[m].x - v[P - m].x, v[m].y - v[P - m].y);
        v[m] = am;
        v[P - m] = bm;
        y0.x += am.x;
        y0.y += am.y;
    }
    const cplx v0 = v[0];
    put(0, y0);
#pragma unroll
    for (int q = 1; q <= H; ++q) {
        double ax = v0.x, ay = v0.y, bx = 0.0, by = 0.0;
#pragma unroll
        for (int m = 1; m <= H; ++m) {
            const double wc = mr_roots<P>::c[(m * q) % P], ws = sg * mr_roots<P>::s[(m * q) % P];
            ax = fma(v[m].x, wc, ax);
            ay = fma(v[m].y, wc, ay);
            bx = fma(v[P - m].x, ws, bx);
            by = fma(v[P - m].y, ws, by);
        }
        put(q, make_double2(ax - by, ay + bx));
        put(P - q, make_double2(ax + by, ay - bx));
    }
}

// 25 points as 5 x 5 (Cooley-Tukey in registers, all indices compile-time): n = 5 n1 + n0, q = qa + 5 qb
//   A[n0][qa] = sum_n1 x[5 n1 + n0] W5^(n1 qa);   X[qa + 5 qb] = sum_n0 (A[n0][qa] W25^(n0 qa)) W5^(n0 qb)
// 224 FMAs instead of the 576 of the direct conjugate-pair form.
template <typename PUT>
__device__ __forceinline__ void dft25_ct(cplx *u, const double sg, PUT put)
{
#pragma unroll
    for (int n0 = 0; n0 < 5; ++n0) {
        cplx x5[5], y5[5];
#pragma unroll
        for (int n1 = 0; n1 < 5; ++n1) x5[n1] = u[5 * n1 + n0];
        dft_odd_inplace<5>(x5, sg, [&](int qa, cplx y) { y5[qa] = y; });
#pragma unroll
        for (int qa = 0; qa < 5; ++qa) {
            cplx y = y5[qa];
            if (n0 > 0 && qa > 0) y = mcmul(y, make_double2(mr_roots<25>::c[n0 * qa], sg * mr_roots<25>::s[n0 * qa]));
            u[5 * qa + n0] = y;                   // row qa now holds A[.][qa]
        }
    }
#pragma unroll
    for (int qa = 0; qa < 5; ++qa) {
        cplx x5[5];
#pragma unroll
        for (int n0 = 0; n0 < 5; ++n0) x5[n0] = u[5 * qa + n0];
        dft_odd_inplace<5>(x5, sg, [&](int qb, cplx y) { put(qa + 5 * qb, y); });
    }
}

// N = 2, 4, 8, 16 points in registers: decimation in time, X = E[k] +- W_N^k O[k]; x is read with a compile-time stride
template <int N>
__device__ __forceinline__ void dft_pow2_rec(const cplx *x, int stride, cplx *X, const double sg)
{
    if constexpr (N == 1) {
        X[0] = x[0];
    } else if constexpr (N == 2) {
        const cplx a = x[0], b = x[stride];
        X[0] = make_double2(a.x + b.x, a.y + b.y);
        X[1] = make_double2(a.x - b.x, a.y - b.y);
    } else {
        cplx E[N / 2], O[N / 2];
        dft_pow2_rec<N / 2>(x, 2 * stride, E, sg);
        dft_pow2_rec<N / 2>(x + stride, 2 * stride, O, sg);
#pragma unroll
        for (int k = 0; k < N / 2; ++k) {
            cplx t;
            if (k == 0)
                t = O[0];
            else if (4 * k == N)
                t = make_double2(-sg * O[k].y, sg * O[k].x);          // times (0, sg): W = exp(sg i pi / 2)
            else
                t = mcmul(O[k], make_double2(mr_roots<16>::c[k * (16 / N)], sg * mr_roots<16>::s[k * (16 / N)]));
            X[k] = make_double2(E[k].x + t.x, E[k].y + t.y);
            X[k + N / 2] = make_double2(E[k].x - t.x, E[k].y - t.y);
        }
    }
}

// Composite transforms "through rows": row(r) returns a pointer to the item's r-th point -- a register array (level A, all
// indices compile-time) or the item's own LDS rows (level B: written and read back by the same thread, no barrier), where the
// N1 x N2 intermediate then costs no registers and a pass has room for the next tile's prefetch.
//   dft_ct_rows:  n = N2 n1 + n2, k = k1 + N1 k2, twiddles W_N^(n2 k1) between the steps (the arithmetic of dft25_ct)
//   dft_pfa_rows: gcd(N1, N2) = 1, n = (N2 n1 + N1 n2) mod N, k = k1 mod N1 = k2 mod N2 (Good-Thomas): no twiddles at all
template <int N1, int N2, typename ROW, typename PUT>
__device__ __forceinline__ void dft_ct_rows(ROW row, const double sg, PUT put)
{
    constexpr int N = N1 * N2;
#pragma unroll
    for (int n2 = 0; n2 < N2; ++n2) {
        cplx x[N1];
#pragma unroll
        for (int n1 = 0; n1 < N1; ++n1) x[n1] = *row(N2 * n1 + n2);
        dft_odd_inplace<N1>(x, sg, [&](int k1, cplx y) {
            if (n2 > 0 && k1 > 0) y = mcmul(y, make_double2(mr_roots<N>::c[n2 * k1], sg * mr_roots<N>::s[n2 * k1]));
            *row(N2 * k1 + n2) = y;
        });
    }
#pragma unroll
    for (int k1 = 0; k1 < N1; ++k1) {
        cplx x[N2];
#pragma unroll
        for (int n2 = 0; n2 < N2; ++n2) x[n2] = *row(N2 * k1 + n2);
        dft_odd_inplace<N2>(x, sg, [&](int k2, cplx y) { put(k1 + N1 * k2, y); });
    }
}

constexpr int mr_inv_mod(int a, int m)
{
    for (int x = 1; x < m; ++x)
        if ((a * x) % m == 1) return x;
    return 0;
}

template <int N1, int N2, typename ROW, typename PUT>
__device__ __forceinline__ void dft_pfa_rows(ROW row, const double sg, PUT put)
{
    constexpr int N = N1 * N2;
    constexpr int E1 = N2 * mr_inv_mod(N2 % N1, N1), E2 = N1 * mr_inv_mod(N1 % N2, N2);     // k = (E1 k1 + E2 k2) mod N
#pragma unroll
    for (int n2 = 0; n2 < N2; ++n2) {
        cplx x[N1];
#pragma unroll
        for (int n1 = 0; n1 < N1; ++n1) x[n1] = *row((N2 * n1 + N1 * n2) % N);
        dft_odd_inplace<N1>(x, sg, [&](int k1, cplx y) { *row((N2 * k1 + N1 * n2) % N) = y; });
    }
#pragma unroll
    for (int k1 = 0; k1 < N1; ++k1) {
        cplx x[N2];
#pragma unroll
        for (int n2 = 0; n2 < N2; ++n2) x[n2] = *row((N2 * k1 + N1 * n2) % N);
        dft_odd_inplace<N2>(x, sg, [&](int k2, cplx y) { put((E1 * k1 + E2 * k2) % N, y); });
    }
}

// radices whose level-B transform goes through the item's LDS rows
constexpr bool mr2_rows(int rb) { return rb == 25; }

template <int P, typename ROW, typename PUT>
__device__ __forceinline__ void dft_rows(ROW row, const double sg, PUT put)
{
    if constexpr (P == 25)
        dft_ct_rows<5, 5>(row, sg, put);
    else
        dft_pfa_rows<3, 5>(row, sg, put);
}

template <int P, bool PFA = false, typename PUT>
__device__ __forceinline__ void dft_any(cplx *v, const double sg, PUT put)
{
    if constexpr (P == 25) {
        dft25_ct(v, sg, put);
    } else if constexpr (P == 15 && PFA) {
        dft_pfa_rows<3, 5>([&](int r) { return &v[r]; }, sg, put);      // 3 x 5 without twiddles: 134 operations against 230
    } else if constexpr (P == 2 || P == 4 || P == 8 || P == 16) {
        cplx X[P];
        dft_pow2_rec<P>(v, 1, X, sg);
#pragma unroll
        for (int q = 0; q < P; ++q) put(q, X[q]);
    } else {
        dft_odd_inplace<P>(v, sg, put);
    }
}

// tile width (columns) and prefetch policy of a pair, shared by the kernel and the host.  (The build-time A/B switches of rounds
// 2-3 -- 25-point level through LDS rows, Good-Thomas 15-point level, 512-lane workgroups for radices above 128, 64-column
// radix-64 tiles, int16 staging -- are settled: DESIGN.md 3.2 has what each measured; the winners are what is written here.)
constexpr int MR2_PF_BUDGET = 140;       // registers a pass may spend on the tile in flight
constexpr int MR2_FUSED_LB = 2;          // workgroups per CU of the fused spectral pass
constexpr int mr2_nt(int) { return 256; }       // lanes per workgroup (512 lanes on 32-column tiles for radices above 128 lost to barriers)
constexpr int mr2_log2t(int r) { return r <= 16 ? 8 : r <= 32 ? 7 : r <= 63 ? 6 : r <= 128 ? 5 : 4; }
constexpr bool mr2_prefetch(int ra, int rb)
{
    const int t = 1 << mr2_log2t(ra * rb);
    const int na = (t * rb + mr2_nt(ra * rb) - 1) / mr2_nt(ra * rb);
    if (mr2_rows(rb)) return na * ra * 4 + 40 <= 140;     // level B works five points at a time (dft_rows)
    // registers: a tile in flight + a level-B transform (the 16-point level keeps two half-size scratch arrays: with the tile in
    // flight the (15,16) pass spilled 87 registers and took 580 us on the 86.4 M-point transform, 433 without)
    return na * ra * 4 + rb * 4 + (rb == 25 || ra == 25 ? 40 : 0) + (rb == 16 ? 24 : 0) <= MR2_PF_BUDGET;
}

// int16 first pass (IN_MODE 2): a tile's row segments are only T * 4 bytes (128 for radix 64).  The workgroup stages the rows of
// SUP consecutive tiles in LDS as int16 pairs -- 256..512-byte segments, the next block's loads in flight in registers across the
// SUP tiles of this one -- and the tiles read their points from there.
constexpr int mr2_i16_sup(int r)
{
    const int t = 1 << mr2_log2t(r);
    for (int sw = 128; sw > t; sw >>= 1)                          // staged columns; two workgroups per CU must still fit (80 KB each)
        if (t * r * 16 + r * sw * 4 + r * 16 + 2048 <= 80 * 1024) return sw / t;
    return 1;
}
// NTL: the input is read with non-temporal loads (mr_pass_desc::nt_in: arrays the Infinity Cache cannot hold).  A compile-time choice --
// decided per tile at run time it cost the forward passes 3-8 % -- so the passes the 60-minute captures use have two instantiations
template <int RA, int RB, int IN_MODE, int OUT_MODE, int INVERSE, int NTL = 0>
__global__ void __launch_bounds__(mr2_nt(RA * RB), mr2_nt(RA * RB) == 512 ? 1 : OUT_MODE == 2 ? MR2_FUSED_LB : 2)
mr2_pass(const cplx *__restrict__ in, cplx *__restrict__ out, mr_pass_desc d, const cplx *__restrict__ tw_lo, const cplx *__restrict__ tw_hi,
         int ntiles)
{
    constexpr int LOG2T = mr2_log2t(RA * RB);
    constexpr int SUP = IN_MODE == 2 ? mr2_i16_sup(RA * RB) : 1;  // tiles per staged block of int16 rows
    constexpr int PF = (SUP == 1 && mr2_prefetch(RA, RB) && (OUT_MODE == 0 || OUT_MODE == 3 || ((OUT_MODE == 1 || OUT_MODE == 4) && mr2_rows(RB)))) ? 1 : 0;
    constexpr int R = RA * RB, T = 1 << LOG2T;
    constexpr int NT = mr2_nt(R);                                 // lanes per workgroup
    constexpr int NA = (T * RB + NT - 1) / NT, NB = (T * RA + NT - 1) / NT;
    constexpr bool inv = INVERSE != 0;
    constexpr double sg = inv ? 1.0 : -1.0;
    // column-major read-out of a first pass (a column's R outputs contiguous): lanes run over columns, R points apart -- for R a multiple of
    // 16 that is ONE bank pair for every lane (mr2_pass<8,8>: 0.21 of its LDS cycles were conflicts, profiles/r05_v3/sq_counters_iq.txt):
    // such radices keep one point of padding per column
    constexpr int CS = R + ((R % 16 == 0) ? 1 : 0);
    __shared__ cplx tile[T * CS];
    __shared__ cplx wr[R];
    const int t = (int)threadIdx.x;
    const int ncol = (int)d.ncol, P = (int)d.P;
    const long long in_rs = d.in_rs ? d.in_rs : (long long)ncol;  // distance between the rows of a column's R inputs
    for (int i = t; i < R; i += NT) {
        double sn, cs;
        sincospi(2.0 * (double)i / (double)R, &sn, &cs);
        wr[i] = make_double2(cs, inv ? sn : -sn);                 // W_R^i in the direction of this pass
    }
    __shared__ cplx gpow[(OUT_MODE == 1 || OUT_MODE == 2) ? RB : 1];   // exp(-i pi qb RA P / L): the spectrum's step along qb
    if ((OUT_MODE == 1 || OUT_MODE == 2) && t < RB) {
        double sn, cs;
        sincospi((double)((long long)t * RA * d.Ptw) / (double)d.Ltw, &sn, &cs);
        gpow[t] = make_double2(cs, -sn);
    }
    cplx pre[NA][RA];
    const long long in_lim = ((IN_MODE == 1 || IN_MODE == 0) && d.in_len > 0) ? d.in_len : 0x7fffffffffffffffll;     // zero padding behind the input
    constexpr int SW = SUP * T, NSL = SUP > 1 ? (R * SW + NT - 1) / NT : 1;
    __shared__ short2 stage[SUP > 1 ? R * SW : 1];
    short2 sreg[NSL];
    auto stage_fetch = [&](int st) {                              // rows of block st -> registers (row r, staged column c: item r SW + c)
#pragma unroll
        for (int i = 0; i < NSL; ++i) {
            const int item = t + NT * i;
            const int c = item & (SW - 1), r = item / SW;
            const long long j = (long long)st * SW + c;
            short2 v = make_short2(0, 0);
            if (r < R && j < ncol) v = ((const short2 *)in)[j + (long long)r * in_rs];
            sreg[i] = v;
        }
    };
    auto prefetch_as = [&](int tix, auto ntc) {
        constexpr bool NTLOAD = decltype(ntc)::value;    // non-temporal loads (arrays the Infinity Cache cannot hold: mr_pass_desc::nt_in)
#pragma unroll
        for (int ia = 0; ia < NA; ++ia) {
            const int item = t + NT * ia;
            const int c = item & (T - 1), b = item >> LOG2T;
            const int j = tix * T + c;
            const bool ok = item < T * RB && j < ncol;
#pragma unroll
            for (int a = 0; a < RA; ++a) {
                cplx v = make_double2(0.0, 0.0);
                const long long adr = (long long)j + (long long)(a * RB + b) * in_rs;
                if (ok && ((IN_MODE != 1 && IN_MODE != 0) || adr < in_lim)) {
                    if (IN_MODE == 2) {
                        const short2 x2 = SUP > 1 ? stage[(a * RB + b) * SW + (tix & (SUP - 1)) * T + c] : ((const short2 *)in)[adr];
                        v = make_double2((double)x2.x, (double)x2.y);
                    } else {
                        if (NTLOAD) {
                            typedef double mr_v2d __attribute__((ext_vector_type(2)));
                            const mr_v2d q = __builtin_nontemporal_load((const mr_v2d *)&in[adr]);
                            v = make_double2(q.x, q.y);
                        } else
                            v = in[adr];
                        if (IN_MODE == 1) v = make_double2(v.y, v.x);
                    }
                }
                pre[ia][a] = v;
            }
        }
    };
    auto prefetch = [&](int tix) { prefetch_as(tix, std::integral_constant<bool, NTL == 1>()); };
    auto lookup = [&](int e) { return mconj_if(mcmul(tw_hi[e >> MR_LO_BITS], tw_lo[e & (MR_LO - 1)]), inv); };
    int tix = blockIdx.x * SUP;
    if (PF && tix < ntiles) prefetch(tix);
    if (SUP > 1 && tix < ntiles) stage_fetch(blockIdx.x);
    // a workgroup takes blocks of SUP consecutive tiles, gridDim.x blocks apart (SUP == 1: every gridDim.x-th tile)
    for (; tix < ntiles; tix = ((tix + 1) & (SUP - 1)) ? tix + 1 : tix + 1 + ((int)gridDim.x - 1) * SUP) {
        if (SUP > 1 && (tix & (SUP - 1)) == 0) {
            mr_lds_barrier();                                     // the previous block's tiles have read the stage
#pragma unroll
            for (int i = 0; i < NSL; ++i)
                if (t + NT * i < R * SW) stage[t + NT * i] = sreg[i];
            mr_lds_barrier();
            const int st = tix / SUP + (int)gridDim.x;
            if (st * SUP < ntiles) stage_fetch(st);               // in flight across this block's tiles
        }
        const int j0 = tix * T;
        const int tn = min(T, ncol - j0);
        if (!PF) prefetch(tix);                                   // register-hungry radices: no tile kept in flight across level B
        mr_lds_barrier();                                         // the previous tile has been read out of LDS
        // ---- level A ----
#pragma unroll
        for (int ia = 0; ia < NA; ++ia) {
            const int item = t + NT * ia;
            if (item < T * RB) {
                const int c = item & (T - 1), b = item >> LOG2T;
                cplx v[RA];
#pragma unroll
                for (int a = 0; a < RA; ++a) v[a] = pre[ia][a];
                if (P > 1 && c < tn) {                            // input twiddle W_{P R}^{k (a RB + b)}: two look-ups, then a recurrence
                    const int k = mr_twiddle_k(d, (j0 + c) % P);
                    cplx w = lookup(k * b);
                    const cplx wstep = lookup(k * RB);
#pragma unroll
                    for (int a = 0; a < RA; ++a) {
                        v[a] = mcmul(v[a], w);
                        if (a + 1 < RA) w = mcmul(w, wstep);
                    }
                }
                dft_any<RA>(v, sg, [&](int qa, cplx y) {
                    if (qa > 0 && b > 0) y = mcmul(y, wr[b * qa]);
                    tile[((qa * RB + b) << LOG2T) + c] = y;
                });
            }
        }
        mr_lds_barrier();
        if (PF && tix + (int)gridDim.x < ntiles) prefetch(tix + gridDim.x);       // in flight during level B
        // ---- level B ----
        if (P == 1) {
            // first pass: a column's R outputs are contiguous in memory.  Results go back to LDS in output order
            // (column-major), then the workgroup copies one contiguous run.
            cplx res[NB][RB];
#pragma unroll
            for (int ib = 0; ib < NB; ++ib) {
                const int item = t + NT * ib;
                if (item < T * RA) {
                    const int c = item & (T - 1), qa = item >> LOG2T;
                    cplx u[RB];
#pragma unroll
                    for (int b = 0; b < RB; ++b) u[b] = tile[((qa * RB + b) << LOG2T) + c];
                    dft_any<RB>(u, sg, [&](int qb, cplx y) { res[ib][qb] = y; });
                }
            }
            mr_lds_barrier();
#pragma unroll
            for (int ib = 0; ib < NB; ++ib) {
                const int item = t + NT * ib;
                if (item < T * RA) {
                    const int c = item & (T - 1), qa = item >> LOG2T;
#pragma unroll
                    for (int qb = 0; qb < RB; ++qb) tile[c * CS + qa + RA * qb] = res[ib][qb];
                }
            }
            mr_lds_barrier();
            if (d.qmap) {
                // distributed transform: every output goes where the following exchange sends it from (wfx_dist.hip, E2):
                // consecutive q of one destination are consecutive addresses
                for (int e = t; e < R * tn; e += NT) {
                    const int c = e / R, q = e - c * R;
                    const mr_qmap m = d.qmap[q];
                    ((cplx *)m.base)[(long long)(j0 + c) * m.stride] = tile[c * CS + q];
                }
            } else {
                const long long o0 = (long long)j0 * R;
                for (int e = t; e < R * tn; e += NT) out[o0 + e] = tile[CS == R ? e : e + e / R];
            }
        } else if (OUT_MODE == 2) {
            // LAST forward pass fused with the FIRST inverse pass.  With the inverse's radices taken in the reverse order, column j
            // of this pass produces X[j + q P], q < R -- exactly what column j of an inverse first pass of the same radix reads.
            // The thread that finishes the forward transform for (column, qa) holds X[qa + RA qb] for all qb: it multiplies by the
            // spectrum and runs the inverse's RB-point level on them IN REGISTERS (inverse split q = qa + RA qb, n = nb + RB na:
            // x[n] = sum_qa W_R^(-qa nb) W_RA^(-qa na) sum_qb X[qa + RA qb] W_RB^(-qb nb)), so the spectrum never travels to
            // memory and only one more LDS round trip than a plain first pass is needed.
#pragma unroll
            for (int ib = 0; ib < NB; ++ib) {
                const int item = t + NT * ib;
                if (item < T * RA) {
                    const int c = item & (T - 1), qa = item >> LOG2T;
                    cplx u[RB];
#pragma unroll
                    for (int b = 0; b < RB; ++b) u[b] = tile[((qa * RB + b) << LOG2T) + c];
                    const long long o = (long long)(j0 + c) + (long long)qa * P;         // + qb RA P
                    const double il = 1.0 / (double)d.Ltw;
                    double sn, cs;
                    sincospi((double)o * il, &sn, &cs);
                    const cplx g0 = make_double2(-sn * il, -cs * il);
                    // the spectrum values go through the item's own LDS rows (written and read back by the same thread) instead of
                    // a second register array: the pass already sits at the register limit of two workgroups per CU
                    dft_any<RB>(u, sg, [&](int qb, cplx v) {
                        tile[((qa * RB + qb) << LOG2T) + c] =
                            (o == 0 && qb == 0) ? make_double2(0.0, 0.0) : mcmul(v, mcmul(g0, gpow[qb]));       // times G[o + qb RA P] / L
                    });
#pragma unroll
                    for (int b = 0; b < RB; ++b) u[b] = tile[((qa * RB + b) << LOG2T) + c];
                    dft_any<RB>(u, 1.0, [&](int nb, cplx v) {
                        if (qa > 0 && nb > 0) {
                            const cplx w = wr[qa * nb];
                            v = mcmul(v, make_double2(w.x, -w.y));
                        }
                        tile[((qa * RB + nb) << LOG2T) + c] = v;       // the rows this item read: in place
                    });
                }
            }
            mr_lds_barrier();
            cplx xo[NA][RA];
#pragma unroll
            for (int ia = 0; ia < NA; ++ia) {                     // inverse RA-point level over qa, for (column, nb)
                const int item = t + NT * ia;
                if (item < T * RB) {
                    const int c = item & (T - 1), nb = item >> LOG2T;
                    cplx v[RA];
#pragma unroll
                    for (int a = 0; a < RA; ++a) v[a] = tile[((a * RB + nb) << LOG2T) + c];
                    dft_any<RA>(v, 1.0, [&](int na, cplx r) { xo[ia][na] = r; });
                }
            }
            mr_lds_barrier();
#pragma unroll
            for (int ia = 0; ia < NA; ++ia) {                     // first-pass output order: a column's R values are contiguous
                const int item = t + NT * ia;
                if (item < T * RB) {
                    const int c = item & (T - 1), nb = item >> LOG2T;
#pragma unroll
                    for (int na = 0; na < RA; ++na) tile[c * CS + nb + RB * na] = xo[ia][na];
                }
            }
            mr_lds_barrier();
            const long long o0 = (long long)j0 * R;
            for (int e = t; e < R * tn; e += NT) out[o0 + e] = tile[CS == R ? e : e + e / R];
        } else {
            // few, long transforms (T * RA <= 128 items): two threads share one, each producing half of the output pairs
            constexpr int SPLIT = (T * RA <= NT / 2 && RB >= 9 && RB != 25 && !mr2_rows(RB)) ? 2 : 1;
            constexpr int NBS = (T * RA * SPLIT + NT - 1) / NT;
            constexpr int HB = (RB - 1) / 2, HB1 = SPLIT == 2 ? (HB + 1) / 2 : HB;
#pragma unroll
            for (int ib = 0; ib < NBS; ++ib) {
                const int item = t + NT * ib;
                const int part = SPLIT == 2 ? (item >= T * RA ? 1 : 0) : 0;
                const int it2 = item - part * (T * RA);
                const int c = it2 & (T - 1), qa = it2 >> LOG2T;
                if (item < T * RA * SPLIT && c < tn) {
                    constexpr bool ROWS = mr2_rows(RB);
                    cplx u[ROWS ? 1 : RB];
                    if constexpr (!ROWS) {
#pragma unroll
                        for (int b = 0; b < RB; ++b) u[b] = tile[((qa * RB + b) << LOG2T) + c];
                    }
                    const int j = j0 + c;
                    const int k = j % P;
                    // (out_rs: the last inverse pass of a distributed transform, P == ncol -- output row q of column j goes to
                    // j + q out_rs, rows with a halo between them)
                    const long long orow = d.out_rs ? d.out_rs : (long long)P;
                    const long long obase = (long long)(j - k) * R + k + (long long)qa * orow;   // + qb * RA * P
                    const long long ostep = (long long)RA * orow;
                    cplx g0 = make_double2(0.0, 0.0);
                    long long gbase = 0;                          // global frequency index of obase (== obase unless distributed)
                    if (OUT_MODE == 1) {
                        const double il = 1.0 / (double)d.Ltw;
                        gbase = mr_global_index(d, obase);
                        double sn, cs;
                        sincospi((double)gbase * il, &sn, &cs);
                        g0 = make_double2(-sn * il, -cs * il);
                    }
                    auto emit = [&](int qb, cplx y) {
                        const long long o = obase + (long long)qb * ostep;
                        if (OUT_MODE == 3) {                       // plain forward pass that leaves a range of (GLOBAL) bins unstored
                            const long long os = mr_global_index(d, o);
                            if (os > d.skip_lo && os < d.skip_hi) return;
                        }
                        if (OUT_MODE == 1) {
                            // times G[o] / L, G[o] = -i exp(-i pi o / L) (G[0] = 0) = g0 * gstep^qb: the powers of the
                            // (kernel-uniform) step come from LDS, the start from one sincospi per thread
                            y = (gbase == 0 && qb == 0) ? make_double2(0.0, 0.0) : mcmul(y, mcmul(g0, gpow[qb]));
                        }
                        if (OUT_MODE == 4) y = mcmul(y, d.gtab[o]);        // zero-padded convolution: the transformed kernel, from memory
                        out[o] = y;
                    };
                    if constexpr (ROWS) {
                        dft_rows<RB>([&](int r) { return &tile[((qa * RB + r) << LOG2T) + c]; }, sg, emit);
                    } else if (SPLIT == 1) {
                        dft_any<RB, true>(u, sg, emit);
                    } else if (part == 0) {
                        dft_odd_part<RB, 1, HB1, true>(u, sg, emit);
                    } else {
                        dft_odd_part<RB, HB1 + 1, HB, false>(u, sg, emit);
                    }
                }
            }
        }
    }
}

// two-level table of the Hilbert kernel spectrum G[o] / L = (-i / L) exp(-i pi o / L), o > 0:
// glo[i] = (-i / L) exp(-i pi i / L) (i < 2048), ghi[i] = exp(-i pi 2048 i / L); G[o] / L = ghi[o >> 11] * glo[o & 2047]
__global__ void __launch_bounds__(256) mr_fill_gtables(cplx *__restrict__ glo, cplx *__restrict__ ghi, long long L, int nhi)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    const double inv_l = 1.0 / (double)L;
    if (i < MR_LO) {
        double s, c;
        sincospi((double)i / (double)L, &s, &c);
        glo[i] = make_double2(-s * inv_l, -c * inv_l);
    }
    if (i < nhi) {
        const long long t = ((long long)i << MR_LO_BITS) % (2 * L);
        double s, c;
        sincospi((double)t / (double)L, &s, &c);
        ghi[i] = make_double2(c, -s);
    }
}

// two-level twiddle table for modulus `mod`: lo[i] = W^i (i < 2048), hi[i] = W^(2048 i)
__global__ void __launch_bounds__(256) mr_fill_tables(cplx *__restrict__ lo, cplx *__restrict__ hi, long long mod, int nhi)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < MR_LO) {
        double s, c;
        sincospi(2.0 * (double)(i % mod) / (double)mod, &s, &c);
        lo[i] = make_double2(c, -s);
    }
    if (i < nhi) {
        const long long t = ((long long)i << MR_LO_BITS) % mod;
        double s, c;
        sincospi(2.0 * (double)t / (double)mod, &s, &c);
        hi[i] = make_double2(c, -s);
    }
}

// Radix pairs with a register-resident two-level pass (RA <= RB; both from {4,5,7,8,9,11,13,15,16,25}).
// Every pair costs four kernels (inverse, forward, forward from packed reals, forward + Hilbert spectrum).
#define WFX_MR2_PAIRS(X)                                                                                       \
    X(16, 16) X(15, 16) X(15, 15) X(13, 16) X(13, 15) X(13, 13) X(11, 16) X(11, 15) X(11, 13) X(11, 11)        \
    X(9, 25) X(9, 16) X(9, 15) X(9, 13) X(9, 11) X(9, 9) X(8, 25) X(8, 16) X(8, 15) X(8, 13) X(8, 11) X(8, 9) \
    X(8, 8) X(7, 25) X(7, 16) X(7, 15) X(7, 13) X(7, 11) X(7, 9) X(7, 8) X(7, 7) X(5, 25) X(5, 16) X(5, 15)   \
    X(5, 13) X(5, 11) X(5, 9) X(5, 8) X(5, 7) X(4, 25) X(4, 16) X(4, 15) X(4, 13)

struct mr2_pair {
    int ra, rb;
};
// pairs whose passes have a non-temporal-load instantiation: the radices of the whole-second captures (5, 7, 8, 9, 15, 16, 25); a long
// capture whose length brings 4, 11 or 13 into play runs the default loads (each instantiation is compile time: 43 pairs x 16 forms)
constexpr bool mr2_nt_pair(int ra, int rb)
{
    return ra != 4 && ra != 11 && ra != 13 && rb != 11 && rb != 13;
}
template <int RA, int RB, int IN_MODE, int OUT_MODE, int INVERSE>
static auto mr2_pass_nt()
{
    if constexpr (mr2_nt_pair(RA, RB))
        return &mr2_pass<RA, RB, IN_MODE, OUT_MODE, INVERSE, 1>;
    else
        return &mr2_pass<RA, RB, IN_MODE, OUT_MODE, INVERSE, 0>;
}
static const mr2_pair g_mr2_pairs[] = {
#define X(a, b) {a, b},
    WFX_MR2_PAIRS(X)
#undef X
};

// ---- planning (host) -----------------------------------------------------------------
struct mr_plan_host {
    long long L = 0;
    int npass = 0;
    mr_pass_desc pass[MR_MAXPASS];
    size_t lo_off[MR_MAXPASS], hi_off[MR_MAXPASS];
    size_t glo_off = 0, ghi_off = 0;          // Hilbert spectrum tables (mr_fill_gtables)
    size_t table_elems = 0;
};

// prime-stage list of a radix for the per-prime passes (pairs of 2s as radix 4)
static int mr_stage_factors(int R, int *f)
{
    static const int primes[] = {13, 11, 7, 5, 3};
    int nf = 0, twos = 0;
    for (int p : primes)
        while (R % p == 0) {
            f[nf++] = p;
            R /= p;
        }
    while (R % 2 == 0) {
        ++twos;
        R /= 2;
    }
    for (; twos >= 2; twos -= 2) f[nf++] = 4;
    if (twos) f[nf++] = 2;
    return R == 1 ? nf : -1;
}

static void mr_fill_pass(mr_plan_host &pl, int i, int R, int ra, int rb, long long P, size_t &off)
{
    mr_pass_desc &d = pl.pass[i];
    d.ra = ra;
    d.rb = rb;
    d.R = R;
    d.nf = mr_stage_factors(R, d.f);
    for (int k = d.nf; k < MR_MAXF; ++k) d.f[k] = 1;
    d.P = P;
    d.L = pl.L;
    d.dist = 0;
    d.B = 1;
    d.kb0 = d.kc0 = d.kb1 = d.kscale = d.kstep = 0;
    d.Ptw = P;
    d.Ltw = pl.L;
    d.qmap = nullptr;
    d.skip_lo = d.skip_hi = 0;
    d.in_len = 0;
    d.gtab = nullptr;
    d.in_rs = d.out_rs = 0;
    d.ncol = pl.L / R;
    const int T = MR_TILE / R;
    int t2 = 1, l2 = 0;
    while (t2 * 2 <= T && t2 < 256) {         // power of two <= 256: cheap index arithmetic, a thread keeps its column
        t2 *= 2;
        ++l2;
    }
    d.T = t2;
    d.log2t = l2;
    const long long mod = P * R;
    pl.lo_off[i] = off;
    off += MR_LO;
    pl.hi_off[i] = off;
    off += (size_t)(mod >> MR_LO_BITS) + 2;
}

// relative cost of one pass over the data with a pair (measured: (7,13) 21.5, (15,15) 25, (7,25) 30 us per 115 MB)
static double mr2_pair_cost(int ra, int rb)
{
    double c = 1.0;
    if (rb == 25 || ra == 25) c += 0.25;
    if (ra <= 5) c += 0.15;                    // many short level-A transforms: more twiddles per point
    if (ra * rb < 64) c += 0.2;                // short rows: less work per byte of tile bookkeeping
    // (measured on the 86.4 M-point transform of the 60-minute 48 kHz resampler, 2.76 GB per pass: (5,15) 409 us, (15,16) 549 us as a
    // last pass that skips the unread bins, (8,15) 675 us, (8,16) 605 us, first passes from int16 (8,8) 676 / (5,15) 631 / (8,16) 1102 us:
    // a penalty for radices above 128 -- 16-column tiles -- picked (5,15)(5,15)(8,15)(8,16) and lost 0.3 ms; removed again)
    return c;
}

// cheapest decomposition of L into pair radices (depth-first over the pair table); false if there is none
static bool mr2_search(long long rem, int depth, double cost, std::vector<int> &cur, double &best_cost, std::vector<int> &best)
{
    if (rem == 1) {
        if (cost < best_cost) {
            best_cost = cost;
            best = cur;
        }
        return true;
    }
    if (depth == MR_MAXPASS || cost + 1.0 >= best_cost) return false;
    bool any = false;
    const int np = (int)(sizeof(g_mr2_pairs) / sizeof(g_mr2_pairs[0]));
    const int start = cur.empty() ? 0 : cur.back();        // non-increasing table order: each multiset is visited once
    for (int i = start; i < np; ++i) {
        const long long r = (long long)g_mr2_pairs[i].ra * g_mr2_pairs[i].rb;
        if (rem % r) continue;
        cur.push_back(i);
        any |= mr2_search(rem / r, depth + 1, cost + mr2_pair_cost(g_mr2_pairs[i].ra, g_mr2_pairs[i].rb), cur, best_cost, best);
        cur.pop_back();
    }
    return any;
}

// Plan for a 13-smooth L.  First choice: every pass a radix pair with a register-resident kernel (fewest / cheapest
// passes).  Otherwise: primes grouped greedily into radices <= 256 for the per-prime passes (a group that happens
// to be a pair product still gets the pair kernel).  false when L is not 13-smooth.
static bool mr_make_plan(long long L, mr_plan_host &pl)
{
    static const int primes[] = {13, 11, 7, 5, 3, 2};
    std::vector<int> fs;
    long long rem = L;
    for (int p : primes)
        while (rem % p == 0) {
            fs.push_back(p);
            rem /= p;
        }
    if (rem != 1 || L < 2) return false;
    pl.L = L;
    size_t off = 0;
    long long P = 1;
    std::vector<int> cur, best;
    double best_cost = 1e30;
    bool forced = false;
    if (const char *e = WFX_LAB_ENV("WFX_MR2_PLAN")) {       // experiments: "7x13,7x25,15x15" = the passes in this order
        const int np = (int)(sizeof(g_mr2_pairs) / sizeof(g_mr2_pairs[0]));
        long long prod = 1;
        int a = 0, b = 0, used = 0;
        while (sscanf(e, "%dx%d%n", &a, &b, &used) == 2) {
            int idx = -1;
            for (int i = 0; i < np; ++i)
                if (g_mr2_pairs[i].ra == a && g_mr2_pairs[i].rb == b) idx = i;
            if (idx < 0) break;
            best.push_back(idx);
            prod *= (long long)a * b;
            e += used;
            if (*e == ',') ++e;
        }
        forced = prod == L && (int)best.size() <= MR_MAXPASS;
        if (!forced) best.clear();
    }
    if (forced || (mr2_search(L, 0, 0.0, cur, best_cost, best) && !best.empty())) {
        // small radices first: the first pass pays an extra LDS transposition, the last forward pass the spectrum
        if (!forced)
        std::sort(best.begin(), best.end(), [](int a, int b) {
            return g_mr2_pairs[a].ra * g_mr2_pairs[a].rb < g_mr2_pairs[b].ra * g_mr2_pairs[b].rb;
        });
        pl.npass = (int)best.size();
        for (int i = 0; i < pl.npass; ++i) {
            const mr2_pair &pr = g_mr2_pairs[best[i]];
            mr_fill_pass(pl, i, pr.ra * pr.rb, pr.ra, pr.rb, P, off);
            P *= (long long)pr.ra * pr.rb;
        }
    } else {
        {   // pairs of 2s become radix-4 stages (no multiplications), listed first among the small factors
            int twos = 0;
            std::vector<int> other;
            for (int p : fs) (p == 2 ? ++twos : (other.push_back(p), 0));
            fs = other;
            for (; twos >= 2; twos -= 2) fs.push_back(4);
            if (twos) fs.push_back(2);
            std::sort(fs.begin(), fs.end(), [](int x, int y) { return x > y; });
        }
        // greedy: largest factors first, each into the smallest group that still fits
        std::vector<int> prod, cnt;
        for (int p : fs) {
            int bi = -1;
            for (size_t g = 0; g < prod.size(); ++g)
                if (prod[g] * p <= 256 && cnt[g] < MR_MAXF && (bi < 0 || prod[g] < prod[bi])) bi = (int)g;
            if (bi < 0) {
                prod.push_back(p);
                cnt.push_back(1);
            } else {
                prod[bi] *= p;
                ++cnt[bi];
            }
        }
        if ((int)prod.size() > MR_MAXPASS) return false;
        pl.npass = (int)prod.size();
        for (int i = 0; i < pl.npass; ++i) {
            int ra = 0, rb = 0;
            for (const mr2_pair &pr : g_mr2_pairs)
                if (pr.ra * pr.rb == prod[i]) {
                    ra = pr.ra;
                    rb = pr.rb;
                    break;
                }
            mr_fill_pass(pl, i, prod[i], ra, rb, P, off);
            P *= prod[i];
        }
    }
    pl.glo_off = off;
    off += MR_LO;
    pl.ghi_off = off;
    off += (size_t)(L >> MR_LO_BITS) + 2;
    pl.table_elems = off;
    return true;
}

// the passes of `fwd` in the reverse order (tables appended behind fwd's); false when a pass has no pair kernel or there is only one
static bool mr_make_reverse(const mr_plan_host &fwd, mr_plan_host &inv)
{
    inv = mr_plan_host();
    if (fwd.npass < 2) return false;
    for (int i = 0; i < fwd.npass; ++i)
        if (fwd.pass[i].ra <= 0) return false;
    inv.L = fwd.L;
    inv.npass = fwd.npass;
    size_t off = fwd.table_elems;
    long long P = 1;
    for (int i = 0; i < fwd.npass; ++i) {
        const mr_pass_desc &r = fwd.pass[fwd.npass - 1 - i];
        mr_fill_pass(inv, i, r.R, r.ra, r.rb, P, off);
        P *= r.R;
    }
    inv.table_elems = off;
    return true;
}

struct mr_plan_cache {
    mr_plan_host h;
    mr_plan_host hinv;          // the same radices in the reverse order (inverse behind the fused spectral pass); npass 0: not available
    wfx_devbuf tables;
    bool use_mr2 = true;        // WFX_MR2=0 in the environment forces the per-prime LDS stages (A/B comparisons)
};
static std::map<std::pair<const void *, long long>, mr_plan_cache> g_mr_plans;   // per (context, L)
static std::mutex g_mr_mutex;

// "7x13,7x25,15x15" (pairs) / "98" (per-prime radix) of the plan for L; empty when L is not 13-smooth (diagnostics, tools/)
extern "C" uint64_t wfx_plan_padded_length(uint64_t min_len) { return (uint64_t)wfx_mr_padded_length((long long)min_len); }

extern "C" int wfx_plan_describe(uint64_t L, char *buf, int cap)
{
    mr_plan_host pl;
    if (!buf || cap < 2) return -1;
    buf[0] = 0;
    if (!mr_make_plan((long long)L, pl)) return 0;
    int n = 0;
    for (int i = 0; i < pl.npass; ++i) {
        const mr_pass_desc &d = pl.pass[i];
        n += d.ra > 0 ? snprintf(buf + n, cap - n, "%s%dx%d", i ? "," : "", d.ra, d.rb) : snprintf(buf + n, cap - n, "%s%d", i ? "," : "", d.R);
        if (n >= cap - 1) break;
    }
    return pl.npass;
}

bool wfx_mr_supported(uint64_t L)
{
    mr_plan_host pl;
    return mr_make_plan((long long)L, pl);
}

static int mr_get_plan(wfx_ctx *ctx, long long L, mr_plan_cache **out)
{
    std::lock_guard<std::mutex> lock(g_mr_mutex);
    auto key = std::make_pair((const void *)ctx, L);
    auto it = g_mr_plans.find(key);
    if (it != g_mr_plans.end()) {
        *out = &it->second;
        return 0;
    }
    mr_plan_cache pc;
    if (!mr_make_plan(L, pc.h)) return wfx_fail(ctx, WFX_ERR_BAD_ARG, "length %lld is not 13-smooth", L);
    {
        const char *e = WFX_LAB_ENV("WFX_MR2");
        pc.use_mr2 = !(e && e[0] == '0');
    }
    // The fused spectral pass (last forward + first inverse pass in one kernel, inverse radices reversed) saves one trip of the
    // array through memory.  That pays where the array lives in HBM (the 60-minute captures: 317 MB per array, -30 us of 1140);
    // on the 10-minute capture (57 MB, Infinity-Cache resident) the passes are bound by their on-chip work and the two orders
    // measure 161 against 158 us, so the classic order stays there.  WFX_FUSED_SPECTRUM=1 / 0 forces it on / off.
    const char *fe = WFX_LAB_ENV("WFX_FUSED_SPECTRUM");
    const bool want_fused = fe ? fe[0] != '0' : L >= (1ll << 23);
    const bool have_inv = pc.use_mr2 && want_fused && mr_make_reverse(pc.h, pc.hinv);
    if (!have_inv) pc.hinv.npass = 0;
    WFX_TRY(wfx_reserve(ctx, pc.tables, (have_inv ? pc.hinv.table_elems : pc.h.table_elems) * sizeof(cplx)));
    for (int i = 0; i < pc.h.npass; ++i) {
        const long long mod = pc.h.pass[i].P * pc.h.pass[i].R;
        const int nhi = (int)(mod >> MR_LO_BITS) + 2;
        const int n = nhi > MR_LO ? nhi : MR_LO;
        cplx *base = (cplx *)pc.tables.p;
        WFX_LAUNCH(ctx, K_BS_CHIRP, mr_fill_tables, dim3((n + 255) / 256), dim3(256), base + pc.h.lo_off[i], base + pc.h.hi_off[i], mod, nhi);
    }
    for (int i = 0; have_inv && i < pc.hinv.npass; ++i) {
        const long long mod = pc.hinv.pass[i].P * pc.hinv.pass[i].R;
        const int nhi = (int)(mod >> MR_LO_BITS) + 2;
        const int n = nhi > MR_LO ? nhi : MR_LO;
        cplx *base = (cplx *)pc.tables.p;
        WFX_LAUNCH(ctx, K_BS_CHIRP, mr_fill_tables, dim3((n + 255) / 256), dim3(256), base + pc.hinv.lo_off[i], base + pc.hinv.hi_off[i], mod, nhi);
    }
    {
        const int nhi = (int)(L >> MR_LO_BITS) + 2;
        const int n = nhi > MR_LO ? nhi : MR_LO;
        cplx *base = (cplx *)pc.tables.p;
        WFX_LAUNCH(ctx, K_BS_CHIRP, mr_fill_gtables, dim3((n + 255) / 256), dim3(256), base + pc.h.glo_off, base + pc.h.ghi_off, L, nhi);
    }
    auto ins = g_mr_plans.emplace(key, pc);
    *out = &ins.first->second;
    return 0;
}

static void mr_padded_release(wfx_ctx *ctx);
static void czt_release(wfx_ctx *ctx);

void wfx_mr_release(wfx_ctx *ctx)
{
    mr_padded_release(ctx);
    std::lock_guard<std::mutex> lock(g_mr_mutex);
    for (auto it = g_mr_plans.begin(); it != g_mr_plans.end();) {
        if (it->first.first == (const void *)ctx) {
            if (it->second.tables.p) (void)hipFree(it->second.tables.p);
            it = g_mr_plans.erase(it);
        } else {
            ++it;
        }
    }
}

// hi table = lo + 2048 in every plan (mr_fill_pass lays them out that way)
int wfx_mr_launch_pair(wfx_ctx *ctx, const mr_pass_desc &d_in, const cplx *tw, int in_mode, int out_mode, int dir, const void *src_v, cplx *dst)
{
    mr_pass_desc d = d_in;
    {
        const char *e = getenv("WFX_MR_NT");                                                 // (A/B and test switch: 0 never, 1 always)
        const int forced = e ? atoi(e) : -1;
        const double in_bytes = (double)d.ncol * (double)d.R * (in_mode == 2 ? 4.0 : 16.0);
        d.nt_in = forced >= 0 ? forced : (in_bytes > 128.0 * 1048576.0);
    }
    const cplx *src = (const cplx *)src_v;
    const cplx *lo = tw, *hi = tw + MR_LO;
    const int kid = dir == 0 ? K_FFT_FWD : K_FFT_INV;
    const int lt = mr2_log2t(d.R);
    const int nt = (int)((d.ncol + (1 << lt) - 1) >> lt);
    const int slots = mr2_nt(d.R) == 512 ? 256 : 512;           // resident workgroups on 256 CUs
    const int sup = in_mode == 2 ? mr2_i16_sup(d.R) : 1;         // the int16 pass hands out blocks of `sup` tiles
    const int nblk = (nt + sup - 1) / sup;
    const unsigned g2 = (unsigned)(nblk < slots ? nblk : slots);
    if (nt <= 0) return 0;
    bool done = false;
#define X(RA_, RB_)                                                                                                                   \
    if (!done && d.ra == (RA_) && d.rb == (RB_)) {                                                                                    \
        if (in_mode == 2 && dir == 0 && out_mode == 0)                                                                               \
            WFX_LAUNCH(ctx, kid, (mr2_pass<RA_, RB_, 2, 0, 0>), dim3(g2), dim3(mr2_nt((RA_) * (RB_))), src, dst, d, lo, hi, nt);                        \
        else if (dir == 1 && in_mode == 0 && out_mode == 0 && d.nt_in)                                                               \
            WFX_LAUNCH(ctx, kid, (mr2_pass_nt<RA_, RB_, 0, 0, 1>()), dim3(g2), dim3(mr2_nt((RA_) * (RB_))), src, dst, d, lo, hi, nt);                     \
        else if (dir == 0 && in_mode == 0 && out_mode == 0 && d.nt_in)                                                               \
            WFX_LAUNCH(ctx, kid, (mr2_pass_nt<RA_, RB_, 0, 0, 0>()), dim3(g2), dim3(mr2_nt((RA_) * (RB_))), src, dst, d, lo, hi, nt);                     \
        else if (dir == 1 && in_mode == 0 && out_mode == 0)                                                                          \
            WFX_LAUNCH(ctx, kid, (mr2_pass<RA_, RB_, 0, 0, 1>), dim3(g2), dim3(mr2_nt((RA_) * (RB_))), src, dst, d, lo, hi, nt);                        \
        else if (dir == 0 && out_mode == 1 && in_mode == 0)                                                                          \
            WFX_LAUNCH(ctx, kid, (mr2_pass<RA_, RB_, 0, 1, 0>), dim3(g2), dim3(mr2_nt((RA_) * (RB_))), src, dst, d, lo, hi, nt);                        \
        else if (dir == 0 && out_mode == 2 && in_mode == 0 && d.nt_in)                                                               \
            WFX_LAUNCH(ctx, kid, (mr2_pass_nt<RA_, RB_, 0, 2, 0>()), dim3(g2), dim3(mr2_nt((RA_) * (RB_))), src, dst, d, lo, hi, nt);                     \
        else if (dir == 0 && out_mode == 3 && in_mode == 0 && d.nt_in)                                                               \
            WFX_LAUNCH(ctx, kid, (mr2_pass_nt<RA_, RB_, 0, 3, 0>()), dim3(g2), dim3(mr2_nt((RA_) * (RB_))), src, dst, d, lo, hi, nt);                     \
        else if (dir == 0 && in_mode == 1 && out_mode == 0 && d.nt_in)                                                               \
            WFX_LAUNCH(ctx, kid, (mr2_pass_nt<RA_, RB_, 1, 0, 0>()), dim3(g2), dim3(mr2_nt((RA_) * (RB_))), src, dst, d, lo, hi, nt);                     \
        else if (dir == 0 && out_mode == 2 && in_mode == 0)                                                                          \
            WFX_LAUNCH(ctx, kid, (mr2_pass<RA_, RB_, 0, 2, 0>), dim3(g2), dim3(mr2_nt((RA_) * (RB_))), src, dst, d, lo, hi, nt);                        \
        else if (dir == 0 && out_mode == 3 && in_mode == 0)                                                                          \
            WFX_LAUNCH(ctx, kid, (mr2_pass<RA_, RB_, 0, 3, 0>), dim3(g2), dim3(mr2_nt((RA_) * (RB_))), src, dst, d, lo, hi, nt);                        \
        else if (dir == 1 && out_mode == 3 && in_mode == 0)                                                                          \
            WFX_LAUNCH(ctx, kid, (mr2_pass<RA_, RB_, 0, 3, 1>), dim3(g2), dim3(mr2_nt((RA_) * (RB_))), src, dst, d, lo, hi, nt);                        \
        else if (dir == 0 && out_mode == 4 && in_mode == 0)                                                                          \
            WFX_LAUNCH(ctx, kid, (mr2_pass<RA_, RB_, 0, 4, 0>), dim3(g2), dim3(mr2_nt((RA_) * (RB_))), src, dst, d, lo, hi, nt);                        \
        else if (dir == 0 && in_mode == 1 && out_mode == 0)                                                                          \
            WFX_LAUNCH(ctx, kid, (mr2_pass<RA_, RB_, 1, 0, 0>), dim3(g2), dim3(mr2_nt((RA_) * (RB_))), src, dst, d, lo, hi, nt);                        \
        else if (dir == 0 && in_mode == 0 && out_mode == 0)                                                                          \
            WFX_LAUNCH(ctx, kid, (mr2_pass<RA_, RB_, 0, 0, 0>), dim3(g2), dim3(mr2_nt((RA_) * (RB_))), src, dst, d, lo, hi, nt);                        \
        else                                                                                                                          \
            return wfx_fail(ctx, WFX_ERR_BAD_ARG, "no pass kernel for in_mode %d out_mode %d dir %d", in_mode, out_mode, dir);        \
        done = true;                                                                                                                  \
    }
    WFX_MR2_PAIRS(X)
#undef X
    if (!done) return wfx_fail(ctx, WFX_ERR_BAD_ARG, "no register-resident pass for the radix pair (%d, %d)", d.ra, d.rb);
    return 0;
}

// Any 13-smooth length: the planner's passes (radix pairs where a decomposition into pairs exists, per-prime groups otherwise)
bool wfx_mr_general_plan(long long L, std::vector<wfx_mr_radix> &out)
{
    out.clear();
    mr_plan_host pl;
    if (!mr_make_plan(L, pl)) return false;
    for (int i = 0; i < pl.npass; ++i) out.push_back({pl.pass[i].R, pl.pass[i].ra, pl.pass[i].rb});
    return true;
}

void wfx_mr_general_desc(mr_pass_desc &d, const wfx_mr_radix &r, long long P, long long ncol, long long L)
{
    if (r.ra > 0) {
        wfx_mr_pair_desc(d, r.ra, r.rb, P, ncol, L);
        return;
    }
    d = mr_pass_desc();
    d.ra = d.rb = 0;
    d.R = r.R;
    d.nf = mr_stage_factors(r.R, d.f);
    for (int k = d.nf < 0 ? 0 : d.nf; k < MR_MAXF; ++k) d.f[k] = 1;
    d.P = P;
    d.ncol = ncol;
    d.L = L;
    int t2 = 1, l2 = 0;
    while (t2 * 2 <= MR_TILE / r.R && t2 < 256) {
        t2 *= 2;
        ++l2;
    }
    d.T = t2;
    d.log2t = l2;
    d.dist = 0;
    d.B = 1;
    d.Ptw = P;
    d.Ltw = L;
}

int wfx_mr_launch(wfx_ctx *ctx, const mr_pass_desc &d, const cplx *tw, int in_mode, int out_mode, int dir, const void *src_v, cplx *dst)
{
    if (d.ra > 0) return wfx_mr_launch_pair(ctx, d, tw, in_mode, out_mode, dir, src_v, dst);
    const cplx *src = (const cplx *)src_v;
    const cplx *lo = tw, *hi = tw + MR_LO;
    const int kid = dir == 0 ? K_FFT_FWD : K_FFT_INV;
    const int ntiles = (int)((d.ncol + d.T - 1) / d.T);
    if (ntiles <= 0) return 0;
    const unsigned grid = (unsigned)(ntiles < 512 ? ntiles : 512);
    if (dir == 1 && in_mode == 0 && out_mode == 0)
        WFX_LAUNCH(ctx, kid, (mr_pass<0, 0, 1>), dim3(grid), dim3(256), src, dst, d, lo, hi, ntiles);
    else if (dir == 0 && in_mode == 2 && out_mode == 0)
        WFX_LAUNCH(ctx, kid, (mr_pass<2, 0, 0>), dim3(grid), dim3(256), src, dst, d, lo, hi, ntiles);
    else if (dir == 0 && in_mode == 1 && out_mode == 1)
        WFX_LAUNCH(ctx, kid, (mr_pass<1, 1, 0>), dim3(grid), dim3(256), src, dst, d, lo, hi, ntiles);
    else if (dir == 0 && in_mode == 1 && out_mode == 0)
        WFX_LAUNCH(ctx, kid, (mr_pass<1, 0, 0>), dim3(grid), dim3(256), src, dst, d, lo, hi, ntiles);
    else if (dir == 0 && in_mode == 0 && out_mode == 1)
        WFX_LAUNCH(ctx, kid, (mr_pass<0, 1, 0>), dim3(grid), dim3(256), src, dst, d, lo, hi, ntiles);
    else if (dir == 0 && in_mode == 0 && out_mode == 0)
        WFX_LAUNCH(ctx, kid, (mr_pass<0, 0, 0>), dim3(grid), dim3(256), src, dst, d, lo, hi, ntiles);
    else
        return wfx_fail(ctx, WFX_ERR_BAD_ARG, "no pass kernel for in_mode %d out_mode %d dir %d", in_mode, out_mode, dir);
    return 0;
}

void wfx_mr_all_pairs(std::vector<std::pair<int, int>> &out)
{
    out.clear();
    for (const mr2_pair &pr : g_mr2_pairs) out.emplace_back(pr.ra, pr.rb);
}

bool wfx_mr_is_pair(int ra, int rb)
{
    for (const mr2_pair &pr : g_mr2_pairs)
        if (pr.ra == ra && pr.rb == rb) return true;
    return false;
}

bool wfx_mr_pair_plan(long long L, std::vector<std::pair<int, int>> &pairs)
{
    pairs.clear();
    if (L < 2) return false;
    long long rem = L;
    for (int p : {13, 11, 7, 5, 3, 2})
        while (rem % p == 0) rem /= p;
    if (rem != 1) return false;
    std::vector<int> cur, best;
    double best_cost = 1e30;
    if (!mr2_search(L, 0, 0.0, cur, best_cost, best) || best.empty()) return false;
    std::sort(best.begin(), best.end(), [](int a, int b) {
        return g_mr2_pairs[a].ra * g_mr2_pairs[a].rb < g_mr2_pairs[b].ra * g_mr2_pairs[b].rb;
    });
    for (int i : best) pairs.emplace_back(g_mr2_pairs[i].ra, g_mr2_pairs[i].rb);
    return true;
}

void wfx_mr_pair_desc(mr_pass_desc &d, int ra, int rb, long long P, long long ncol, long long L)
{
    d = mr_pass_desc();
    d.ra = ra;
    d.rb = rb;
    d.R = ra * rb;
    d.nf = mr_stage_factors(d.R, d.f);
    for (int k = d.nf < 0 ? 0 : d.nf; k < MR_MAXF; ++k) d.f[k] = 1;
    d.P = P;
    d.ncol = ncol;
    d.L = L;
    d.T = 1 << mr2_log2t(d.R);
    d.log2t = mr2_log2t(d.R);
    d.dist = 0;
    d.B = 1;
    d.Ptw = P;
    d.Ltw = L;
}

size_t wfx_mr_table_elems(long long mod) { return (size_t)MR_LO + (size_t)(mod >> MR_LO_BITS) + 2; }

int wfx_mr_fill_table(wfx_ctx *ctx, cplx *base, long long mod)
{
    const int nhi = (int)(mod >> MR_LO_BITS) + 2;
    const int n = nhi > MR_LO ? nhi : MR_LO;
    WFX_LAUNCH(ctx, K_BS_CHIRP, mr_fill_tables, dim3((n + 255) / 256), dim3(256), base, base + MR_LO, mod, nhi);
    return 0;
}

// One transform of the plan: `dir` 0 forward / 1 inverse, reading `src`, ping-ponging between A and B; the last pass
// writes to `final_dst` when given.  hilbert: the packed-real load swap on the first forward pass and the multiplication
// by the Hilbert spectrum on the last one.  Unnormalised; natural order in and out (self-sorting passes).
static int mr_run(wfx_ctx *ctx, mr_plan_cache *pc, const cplx *src, cplx *A, cplx *B, int dir, bool hilbert, cplx *final_dst, cplx **result,
                  bool src_i16 = false, long long skip_lo = 0, long long skip_hi = 0)
{
    const cplx *tb = (const cplx *)pc->tables.p;
    const int np = pc->h.npass;
    cplx *dst = (src == A) ? B : A;
    for (int i = 0; i < np; ++i) {
        mr_pass_desc d = pc->h.pass[i];
        if (i == np - 1 && skip_hi > skip_lo + 1 && np > 1) {      // (a first pass stores whole tiles: never skipped; inverse: the padded forms' unread outputs)
            d.skip_lo = skip_lo;
            d.skip_hi = skip_hi;
        }
        const cplx *lo = tb + pc->h.lo_off[i], *hi = tb + pc->h.hi_off[i];
        const int kid = dir == 0 ? K_FFT_FWD : K_FFT_INV;
        const bool first = hilbert && dir == 0 && i == 0, last_fwd = hilbert && dir == 0 && i == np - 1;
        const bool first16 = src_i16 && dir == 0 && i == 0 && !hilbert;       // int16 pairs as the complex input
        if (i == np - 1 && final_dst) dst = final_dst;
        // register-resident two-level pass when the radix is a pair (mr2_pass); per-prime LDS stages otherwise
        bool done = false;
        if (pc->use_mr2 && d.ra > 0 && !(first && last_fwd)) {
            const bool skipping = d.skip_hi != 0 && !first16 && !first && !last_fwd;      // the pair kernels skip in a variant of their own
            if (!skipping) d.skip_lo = d.skip_hi = 0;
            WFX_TRY(wfx_mr_launch_pair(ctx, d, lo, first16 ? 2 : (first ? 1 : 0), last_fwd ? 1 : (skipping ? 3 : 0), dir, src, dst));
            done = true;
        }
        if (!done) {
            const int ntiles = (int)((d.ncol + d.T - 1) / d.T);
            const unsigned grid = (unsigned)(ntiles < 512 ? ntiles : 512);       // 2 persistent workgroups per CU
            if (dir == 1)
                WFX_LAUNCH(ctx, kid, (mr_pass<0, 0, 1>), dim3(grid), dim3(256), src, dst, d, lo, hi, ntiles);
            else if (first16)
                WFX_LAUNCH(ctx, kid, (mr_pass<2, 0, 0>), dim3(grid), dim3(256), src, dst, d, lo, hi, ntiles);
            else if (first && last_fwd)
                WFX_LAUNCH(ctx, kid, (mr_pass<1, 1, 0>), dim3(grid), dim3(256), src, dst, d, lo, hi, ntiles);
            else if (first)
                WFX_LAUNCH(ctx, kid, (mr_pass<1, 0, 0>), dim3(grid), dim3(256), src, dst, d, lo, hi, ntiles);
            else if (last_fwd)
                WFX_LAUNCH(ctx, kid, (mr_pass<0, 1, 0>), dim3(grid), dim3(256), src, dst, d, lo, hi, ntiles);
            else
                WFX_LAUNCH(ctx, kid, (mr_pass<0, 0, 0>), dim3(grid), dim3(256), src, dst, d, lo, hi, ntiles);
        }
        src = dst;
        dst = (dst == A) ? B : A;
    }
    *result = (cplx *)src;
    return 0;
}

// H = Hilbert transform of the even-length real signal x, packed: Re V[p] = H[2p], Im V[p] = H[2p-1 mod N].
// Returns the device pointer holding V (L = n/2 complex values) in *V_out.
int wfx_dev_hilbert_conv_mr(wfx_ctx *ctx, const double *x, uint64_t n, cplx **V_out)
{
    const long long L = (long long)(n / 2);
    mr_plan_cache *pc = nullptr;
    WFX_TRY(mr_get_plan(ctx, L, &pc));
    WFX_TRY(wfx_reserve(ctx, ctx->b_work, (size_t)L * sizeof(cplx)));
    WFX_TRY(wfx_reserve(ctx, ctx->b_work2, (size_t)L * sizeof(cplx)));
    cplx *A = (cplx *)ctx->b_work.p, *B = (cplx *)ctx->b_work2.p;
    if (pc->hinv.npass >= 2) {
        // forward passes 1 .. np-1, then ONE kernel for the last forward pass + spectrum + first inverse pass (the inverse takes its
        // radices in the reverse order, so that pass works on the same columns), then the remaining inverse passes: 2 np - 1 launches
        const cplx *tb = (const cplx *)pc->tables.p;
        const int np = pc->h.npass;
        const cplx *src = (const cplx *)x;
        cplx *dst = A;
        for (int i = 0; i < np; ++i) {
            const bool last = i == np - 1;
            WFX_TRY(wfx_mr_launch_pair(ctx, pc->h.pass[i], tb + pc->h.lo_off[i], i == 0 ? 1 : 0, last ? 2 : 0, 0, src, dst));
            src = dst;
            dst = dst == A ? B : A;
        }
        for (int i = 1; i < np; ++i) {
            WFX_TRY(wfx_mr_launch_pair(ctx, pc->hinv.pass[i], tb + pc->hinv.lo_off[i], 0, 0, 1, src, dst));
            src = dst;
            dst = dst == A ? B : A;
        }
        *V_out = (cplx *)src;
        return 0;
    }
    cplx *mid = nullptr;
    // the packed real input IS x viewed as complex pairs (swapped on load)
    WFX_TRY(mr_run(ctx, pc, (const cplx *)x, A, B, 0, true, nullptr, &mid));
    return mr_run(ctx, pc, mid, A, B, 1, true, nullptr, V_out);
}

// ---------------------------------------------------------------------------------------------
// Any even capture length (wefax.py:174 takes whatever the wav holds).  When L = N/2 has a prime factor above 13 the cyclic
// convolution of length L is embedded in one of length M >= 2L - 1: z zero-padded, the kernel laid out on both sides of
// index 0 (g_ext[j] = kh[2j - 1], |j| < L), the first L outputs are the cyclic result.  M is the cheapest 13-smooth length
// with a radix-pair plan in [2L - 1, 1.08 (2L - 1)] (a power of two can be 2x above 2L - 1; these are 0-2 % above), the
// kernel's M-point transform is computed once per N on the same passes and multiplied in by the last forward pass.
// ---------------------------------------------------------------------------------------------
static void smooth_candidates(long long lo, long long hi, long long cur, int pi, std::vector<long long> &out)
{
    static const int primes[] = {2, 3, 5, 7, 11, 13};
    if (cur >= lo) out.push_back(cur);
    for (int i = pi; i < 6; ++i) {
        if (cur > hi / primes[i]) break;
        smooth_candidates(lo, hi, cur * primes[i], i, out);
    }
}

void wfx_mr_smooth_numbers(long long lo, long long hi, std::vector<long long> &out)
{
    out.clear();
    smooth_candidates(lo, hi, 1, 0, out);
    std::sort(out.begin(), out.end());
}

long long wfx_mr_padded_length(long long min_len)
{
    if (min_len >= (1ll << 31)) return 0;
    if (min_len < 4096) return 0;
    const long long hi = min_len + min_len / 12;
    std::vector<long long> cand;
    smooth_candidates(min_len, hi, 1, 0, cand);
    long long best = 0;
    double best_cost = 1e300;
    for (long long m : cand) {
        std::vector<int> cur, plan;
        double c = 1e30;
        if (!mr2_search(m, 0, 0.0, cur, c, plan) || plan.size() < 2) continue;
        const double total = c * (double)m;
        if (total < best_cost) {
            best_cost = total;
            best = m;
        }
    }
    return best;
}

__device__ __forceinline__ double mr_hilbert_tap_even(long long r, long long N)      // kh[r], N even: (2/N) cot(pi r / N) on odd lags
{
    // reduced to (-N/2, N/2]: a lag just below N is a SMALL negative lag, and r / N next to 1 would have lost the digits of its distance from 1
    // (relative error N x 1e-16 in the largest taps: the padded form was 2e-10 .. 8e-10 from scipy at 14 .. 40 M samples until round 6)
    r %= N;
    if (r > N / 2) r -= N;
    if (r < -(N / 2)) r += N;
    if ((r & 1) == 0) return 0.0;
    double s, c;
    sincospi((double)r / (double)N, &s, &c);
    return (2.0 / (double)N) * (c / s);
}

__global__ void __launch_bounds__(256) mr_padded_fill(cplx *__restrict__ G, long long N, long long L, long long M, double inv_m)
{
    for (long long i = blockIdx.x * 256ll + threadIdx.x; i < M; i += (long long)gridDim.x * 256ll) {
        double v = 0.0;
        if (i < L || M - i < L) {
            const long long j = i < L ? i : i - M;
            v = mr_hilbert_tap_even(2 * j - 1, N) * inv_m;
        }
        G[i] = make_double2(v, 0.0);
    }
}

// rows of the same padded kernel for a distributed transform: points [p0, p0 + count) of g_ext / M (wfx_shard.hip)
__global__ void __launch_bounds__(256) mr_padded_fill_range(cplx *__restrict__ G, long long p0, long long count, long long N, long long L, long long M, double inv_m)
{
    for (long long e = blockIdx.x * 256ll + threadIdx.x; e < count; e += (long long)gridDim.x * 256ll) {
        const long long i = p0 + e;
        double v = 0.0;
        if (i < L || M - i < L) {
            const long long j = i < L ? i : i - M;
            v = mr_hilbert_tap_even(2 * j - 1, N) * inv_m;
        }
        G[e] = make_double2(v, 0.0);
    }
}

// odd N: kh[r] = cot(pi r / 2N) / N on odd lags, -tan(pi r / 2N) / N on even ones (r reduced to (-N/2, N/2]); the kernel is real
// and so are the samples
__device__ __forceinline__ double mr_hilbert_tap_odd(long long r, long long N)
{
    r %= N;
    if (r > N / 2) r -= N;
    if (r < -(N / 2)) r += N;
    if (r == 0) return 0.0;
    double s, c;
    sincospi((double)r / (2.0 * (double)N), &s, &c);
    return ((r & 1) ? (c / s) : -(s / c)) / (double)N;
}

// rows of the odd-length kernel for a distributed PACKED transform of Mh points: point q = (g[2q], g[2q + 1]) / M, M = 2 Mh
// (mr_real_kernel_fill over a range)
__global__ void __launch_bounds__(256) mr_real_kernel_fill_range(cplx *__restrict__ G, long long p0, long long count, long long N, long long Mh, double inv_m)
{
    const long long M = 2 * Mh;
    for (long long e = blockIdx.x * 256ll + threadIdx.x; e < count; e += (long long)gridDim.x * 256ll) {
        const long long q = p0 + e;
        double v[2];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const long long i = 2 * q + h;
            v[h] = (i < N || M - i < N) ? mr_hilbert_tap_odd(i < N ? i : i - M, N) * inv_m : 0.0;
        }
        G[e] = make_double2(v[0], v[1]);
    }
}

int wfx_dev_hilbert_kernel_rows_real(wfx_ctx *ctx, cplx *dst, long long p0, long long count, long long N, long long Mh)
{
    if (count <= 0) return 0;
    WFX_LAUNCH(ctx, K_BS_CHIRP, mr_real_kernel_fill_range, dim3(wfx_stream_grid((uint64_t)count, 256)), dim3(256), dst, p0, count, N, Mh, 0.5 / (double)Mh);
    return 0;
}

int wfx_dev_hilbert_kernel_rows(wfx_ctx *ctx, cplx *dst, long long p0, long long count, long long N, long long M)
{
    if (count <= 0) return 0;
    WFX_LAUNCH(ctx, K_BS_CHIRP, mr_padded_fill_range, dim3(wfx_stream_grid((uint64_t)count, 256)), dim3(256), dst, p0, count, N, N / 2, M, 1.0 / (double)M);
    return 0;
}

struct mr_padded_cache {
    long long M = 0;
    wfx_devbuf ghat;
    bool ready = false;          // the kernel's transform has been enqueued in full (a failed launch in between leaves it false)
};
static std::map<std::pair<const void *, long long>, mr_padded_cache> g_mr_padded;      // per (context, N)

int wfx_dev_hilbert_conv_mr_padded(wfx_ctx *ctx, const double *x, uint64_t n, cplx **V_out, int *handled)
{
    *handled = 0;
    const long long L = (long long)(n / 2);
    mr_padded_cache *pd = nullptr;
    {
        std::lock_guard<std::mutex> lock(g_mr_mutex);
        auto key = std::make_pair((const void *)ctx, (long long)n);
        auto it = g_mr_padded.find(key);
        if (it == g_mr_padded.end()) {
            mr_padded_cache c;
            c.M = wfx_mr_padded_length(2 * L - 1);
            // (one kernel spectrum per context is kept: a context decodes captures of one length over and over)
            for (auto jt = g_mr_padded.begin(); jt != g_mr_padded.end();) {
                if (jt->first.first == (const void *)ctx) {
                    (void)hipStreamSynchronize(ctx->stream);
                    if (jt->second.ghat.p) (void)hipFree(jt->second.ghat.p);
                    jt = g_mr_padded.erase(jt);
                } else {
                    ++jt;
                }
            }
            it = g_mr_padded.emplace(key, c).first;
        }
        pd = &it->second;
    }
    if (pd->M == 0) return 0;
    const long long M = pd->M;
    mr_plan_cache *pc = nullptr;
    WFX_TRY(mr_get_plan(ctx, M, &pc));
    const int np = pc->h.npass;
    for (int i = 0; i < np; ++i)
        if (pc->h.pass[i].ra <= 0) return 0;                       // (the search only returns pair plans; WFX_MR2_PLAN may force others)
    if (np < 2 || !pc->use_mr2) return 0;
    WFX_TRY(wfx_reserve(ctx, ctx->b_work, (size_t)M * sizeof(cplx)));
    WFX_TRY(wfx_reserve(ctx, ctx->b_work2, (size_t)M * sizeof(cplx)));
    cplx *A = (cplx *)ctx->b_work.p, *B = (cplx *)ctx->b_work2.p;
    const cplx *tb = (const cplx *)pc->tables.p;
    if (!pd->ready) {
        WFX_TRY(wfx_reserve(ctx, pd->ghat, (size_t)M * sizeof(cplx)));
        cplx *G = (cplx *)pd->ghat.p;
        WFX_LAUNCH(ctx, K_BS_CHIRP, mr_padded_fill, dim3(wfx_stream_grid((uint64_t)M, 256)), dim3(256), G, (long long)n, L, M, 1.0 / (double)M);
        cplx *res = nullptr;
        WFX_TRY(mr_run(ctx, pc, G, A, B, 0, false, G, &res));      // (np >= 2: G is read by the first pass only)
        pd->ready = true;        // only now: a decode that failed above rebuilds the spectrum instead of multiplying by a partial one
    }
    // forward: first pass from the packed reals (the first L points; zeros behind them), last pass times the kernel's transform
    const cplx *src = (const cplx *)x;
    cplx *dst = A;
    for (int i = 0; i < np; ++i) {
        mr_pass_desc d = pc->h.pass[i];
        if (i == 0) d.in_len = L;
        if (i == np - 1) d.gtab = (const double2 *)pd->ghat.p;
        WFX_TRY(wfx_mr_launch_pair(ctx, d, tb + pc->h.lo_off[i], i == 0 ? 1 : 0, i == np - 1 ? 4 : 0, 0, src, dst));
        src = dst;
        dst = dst == A ? B : A;
    }
    // (the envelope kernel reads V[0 .. L) only -- its index L is V[0] again: the last inverse pass does not store the rest)
    WFX_TRY(mr_run(ctx, pc, src, A, B, 1, false, nullptr, V_out, false, L - 1, M));
    *handled = 1;
    return 0;
}

// ---------------------------------------------------------------------------------------------
// ODD capture lengths (round 4): scipy's kernel for odd N has taps on every lag, so nothing packs the way even lengths do -- but
// samples and kernel are REAL, and a real linear convolution embedded in a cyclic one of M >= 2N points (M even) is two packed
// transforms of M/2 points with one glue pass between them (the resampler's form, resample_mr_glue):
//   z'[q] = x[2q+1] + i x[2q] (the passes' packed-real load; zeros behind the capture), Z' = FFT_{M/2}(z')
//   Z[k] = i conj Z'[M/2 - k] is the transform of x[2q] + i x[2q+1];  X[k] = (Z[k] + conj Z[M/2-k]) / 2 - (i/2) e^{-2 pi i k/M} (Z[k] - conj Z[M/2-k])
//   Y[k] = X[k] * i c[k]   (the kernel g_ext is real and ODD in the lag: its transform is purely imaginary; c[k] / M from a table
//                           made once per N by these very passes)
//   W[k] = (Y[k] + conj Y[M/2-k]) + i e^{2 pi i k/M} (Y[k] - conj Y[M/2-k]);  IFFT_{M/2}(W)[q] = H[2q] + i H[2q+1]
// Half the points of the complex-transform-of-a-real-sequence form rounds 1-3 used for odd N (2^24 points for the 10-minute
// capture plus one sample: 2.24x the headline), at the price of the glue pass: one read and one write of the M/2-point array.
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) mr_real_kernel_fill(cplx *__restrict__ G, long long N, long long Mh, double inv_m)
{
    const long long M = 2 * Mh;
    for (long long q = blockIdx.x * 256ll + threadIdx.x; q < Mh; q += (long long)gridDim.x * 256ll) {
        double v[2];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const long long i = 2 * q + h;
            v[h] = (i < N || M - i < N) ? mr_hilbert_tap_odd(i < N ? i : i - M, N) * inv_m : 0.0;
        }
        G[q] = make_double2(v[0], v[1]);
    }
}

// c[k] = Im G[k], k = 0 .. M/2, from the packed transform Zg of (g[2q], g[2q+1])
__global__ void __launch_bounds__(256) mr_real_kernel_untangle(const cplx *__restrict__ Zg, long long Mh, double *__restrict__ c)
{
    for (long long k = blockIdx.x * 256ll + threadIdx.x; k <= Mh; k += (long long)gridDim.x * 256ll) {
        const cplx zk = Zg[k == Mh ? 0 : k], zr = Zg[k == 0 || k == Mh ? 0 : Mh - k];
        const cplx zc = make_double2(zr.x, -zr.y);
        double sn, cs;
        sincospi((double)k / (double)Mh, &sn, &cs);                      // e^{-2 pi i k / M} = cs - i sn
        const cplx dif = make_double2(zk.x - zc.x, zk.y - zc.y);
        // -(i/2) (cs - i sn) dif: imaginary part -(cs dif.x + sn dif.y) / 2
        c[k] = 0.5 * (zk.y + zc.y) - 0.5 * (cs * dif.x + sn * dif.y);
    }
}

__global__ void __launch_bounds__(256) mr_real_conv_glue(cplx *__restrict__ Z, long long Mh, const double *__restrict__ c)
{
    for (long long k = blockIdx.x * 256ll + threadIdx.x; 2 * k <= Mh; k += (long long)gridDim.x * 256ll) {
        if (k == 0) {                                                    // Y[0] = Y[M/2] = 0: the kernel sums to zero both ways
            Z[0] = make_double2(0.0, 0.0);
            continue;
        }
        const long long r = Mh - k;
        const cplx a = Z[k], b = Z[r];                                   // Z'[k], Z'[M/2 - k]
        const cplx zk = make_double2(b.y, b.x), zm = make_double2(a.y, a.x);      // Z[k] = i conj Z'[r], Z[r] = i conj Z'[k]
        double sn, cs;
        sincospi((double)k / (double)Mh, &sn, &cs);                      // theta_k = 2 pi k / M; theta_r = pi - theta_k
        // X[k] and X[r] from the pair (zk, conj zm)
        const cplx zmc = make_double2(zm.x, -zm.y), zkc = make_double2(zk.x, -zk.y);
        const cplx dk = make_double2(zk.x - zmc.x, zk.y - zmc.y), dr = make_double2(zm.x - zkc.x, zm.y - zkc.y);
        // -(i/2) e^{-i t} d = ((-sn d.x + cs d.y) / 2 ... ) with e^{-i t} = cs - i sn:  -(i/2)(cs - i sn)(d.x + i d.y) = ( cs d.y - sn d.x,  -(cs d.x + sn d.y) ) / 2
        const cplx xk = make_double2(0.5 * (zk.x + zmc.x) + 0.5 * (cs * dk.y - sn * dk.x), 0.5 * (zk.y + zmc.y) - 0.5 * (cs * dk.x + sn * dk.y));
        // for r: e^{-i (pi - t)} = -cs - i sn
        const cplx xr = make_double2(0.5 * (zm.x + zkc.x) + 0.5 * (-cs * dr.y - sn * dr.x), 0.5 * (zm.y + zkc.y) - 0.5 * (-cs * dr.x + sn * dr.y));
        // Y = X * i c
        const double ck = c[k], cr = c[r];
        const cplx yk = make_double2(-xk.y * ck, xk.x * ck), yr = make_double2(-xr.y * cr, xr.x * cr);
        // W[k] = (Y[k] + conj Y[r]) + i e^{i t} (Y[k] - conj Y[r]);  i (cs + i sn)(d.x + i d.y) = ( -(sn d.x + cs d.y), cs d.x - sn d.y )
        const cplx yrc = make_double2(yr.x, -yr.y), ykc = make_double2(yk.x, -yk.y);
        const cplx sk = make_double2(yk.x + yrc.x, yk.y + yrc.y), ek = make_double2(yk.x - yrc.x, yk.y - yrc.y);
        const cplx sr = make_double2(yr.x + ykc.x, yr.y + ykc.y), er = make_double2(yr.x - ykc.x, yr.y - ykc.y);
        Z[k] = make_double2(sk.x - (sn * ek.x + cs * ek.y), sk.y + (cs * ek.x - sn * ek.y));
        // for r: e^{i (pi - t)} = -cs + i sn:  i (-cs + i sn)(d.x + i d.y) = ( -(sn d.x - cs d.y), -(cs d.x + sn d.y) )
        if (r != k) Z[r] = make_double2(sr.x - (sn * er.x - cs * er.y), sr.y - (cs * er.x + sn * er.y));
    }
}

struct mr_real_cache {
    long long Mh = 0;
    wfx_devbuf ctab;
    bool ready = false;
};
static std::map<std::pair<const void *, long long>, mr_real_cache> g_mr_real;          // per (context, N)

// H as a flat array of doubles: H[n] = ((double *)*V_out)[n], n < N
int wfx_dev_hilbert_conv_mr_real(wfx_ctx *ctx, const double *x, uint64_t n, cplx **V_out, int *handled)
{
    *handled = 0;
    mr_real_cache *pd = nullptr;
    {
        std::lock_guard<std::mutex> lock(g_mr_mutex);
        auto key = std::make_pair((const void *)ctx, (long long)n);
        auto it = g_mr_real.find(key);
        if (it == g_mr_real.end()) {
            mr_real_cache c;
            c.Mh = wfx_mr_padded_length((long long)n);                     // 13-smooth, radix pairs, at most a few per cent above n
            for (auto jt = g_mr_real.begin(); jt != g_mr_real.end();) {    // (one table per context)
                if (jt->first.first == (const void *)ctx) {
                    (void)hipStreamSynchronize(ctx->stream);
                    if (jt->second.ctab.p) (void)hipFree(jt->second.ctab.p);
                    jt = g_mr_real.erase(jt);
                } else {
                    ++jt;
                }
            }
            it = g_mr_real.emplace(key, c).first;
        }
        pd = &it->second;
    }
    if (pd->Mh == 0) return 0;
    const long long Mh = pd->Mh;
    mr_plan_cache *pc = nullptr;
    WFX_TRY(mr_get_plan(ctx, Mh, &pc));
    const int np = pc->h.npass;
    for (int i = 0; i < np; ++i)
        if (pc->h.pass[i].ra <= 0) return 0;
    if (np < 2 || !pc->use_mr2) return 0;
    WFX_TRY(wfx_reserve(ctx, ctx->b_work, (size_t)Mh * sizeof(cplx)));
    WFX_TRY(wfx_reserve(ctx, ctx->b_work2, (size_t)Mh * sizeof(cplx)));
    cplx *A = (cplx *)ctx->b_work.p, *B = (cplx *)ctx->b_work2.p;
    const cplx *tb = (const cplx *)pc->tables.p;
    if (!pd->ready) {
        WFX_TRY(wfx_reserve(ctx, pd->ctab, (size_t)(Mh + 1) * sizeof(double) + 64));
        WFX_LAUNCH(ctx, K_BS_CHIRP, mr_real_kernel_fill, dim3(wfx_stream_grid((uint64_t)Mh, 256)), dim3(256), A, (long long)n, Mh, 0.5 / (double)Mh);
        cplx *res = nullptr;
        WFX_TRY(mr_run(ctx, pc, A, A, B, 0, false, nullptr, &res));
        WFX_LAUNCH(ctx, K_BS_CHIRP, mr_real_kernel_untangle, dim3(wfx_stream_grid((uint64_t)Mh + 1, 256)), dim3(256), (const cplx *)res, Mh, (double *)pd->ctab.p);
        pd->ready = true;
    }
    // forward: the first pass packs the reals (points behind (n + 1) / 2 count as zero; the caller keeps x[n] = 0)
    const cplx *src = (const cplx *)x;
    cplx *dst = A;
    for (int i = 0; i < np; ++i) {
        mr_pass_desc d = pc->h.pass[i];
        if (i == 0) d.in_len = (long long)((n + 1) / 2);
        WFX_TRY(wfx_mr_launch_pair(ctx, d, tb + pc->h.lo_off[i], i == 0 ? 1 : 0, 0, 0, src, dst));
        src = dst;
        dst = dst == A ? B : A;
    }
    cplx *Zs = (cplx *)src;
    WFX_LAUNCH(ctx, K_RESAMPLE_PW, mr_real_conv_glue, dim3(wfx_stream_grid((uint64_t)Mh / 2 + 1, 256)), dim3(256), Zs, Mh, (const double *)pd->ctab.p);
    // (H[0 .. n) = the first (n + 1) / 2 points: the last inverse pass does not store the rest)
    WFX_TRY(mr_run(ctx, pc, Zs, A, B, 1, false, nullptr, V_out, false, (long long)((n + 1) / 2) - 1, Mh));
    *handled = 1;
    return 0;
}

static void mr_padded_release(wfx_ctx *ctx)
{
    czt_release(ctx);
    {
        std::lock_guard<std::mutex> lock(g_mr_mutex);
        for (auto it = g_mr_real.begin(); it != g_mr_real.end();) {
            if (it->first.first == (const void *)ctx) {
                if (it->second.ctab.p) (void)hipFree(it->second.ctab.p);
                it = g_mr_real.erase(it);
            } else {
                ++it;
            }
        }
    }
    std::lock_guard<std::mutex> lock(g_mr_mutex);
    for (auto it = g_mr_padded.begin(); it != g_mr_padded.end();) {
        if (it->first.first == (const void *)ctx) {
            if (it->second.ghat.p) (void)hipFree(it->second.ghat.p);
            it = g_mr_padded.erase(it);
        } else {
            ++it;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// scipy.signal.resample (wefax.py:384) for even n0 and num with 13-smooth halves (every whole-second capture at
// the usual rates): rfft and irfft as packed complex transforms of n0/2 and num/2 points on the passes above.
//   Z = FFT_{n0/2}(x[2q] + i x[2q+1]);   X[k] = (Z[k] + conj Z[M-k])/2 - (i/2) e^{-2 pi i k/n0} (Z[k] - conj Z[M-k])
//   Y = the kept bins of X (scipy's Nyquist-bin factors; irfft ignores Im Y[0], Im Y[K])
//   W[k] = (Y[k] + conj Y[K-k]) + i e^{2 pi i k/num} (Y[k] - conj Y[K-k]);  out[2q] + i out[2q+1] = IFFT_K(W)[q] / n0
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) resample_mr_glue(const cplx *__restrict__ Z, long long n0, long long num, cplx *__restrict__ W, int nt)
{
    const long long M = n0 / 2, K = num / 2, nmin = n0 < num ? n0 : num, half = nmin / 2;
    const double edge = (nmin % 2 == 0) ? (num < n0 ? 2.0 : (num > n0 ? 0.5 : 1.0)) : 1.0;
    const double inv_n0 = 1.0 / (double)n0;
    auto bin = [&](long long j) {                       // Y[j], j in [0, K]
        if (j > half) return make_double2(0.0, 0.0);
        const long long a = j >= M ? j - M : j, b = (j == 0 || j >= M) ? 0 : M - j;       // j % M, (M - j) % M for j <= M
        const cplx zk = wfx_ld(Z + a, nt), zb = wfx_ld(Z + b, nt), zc = make_double2(zb.x, -zb.y);
        double sn, cs;
        sincospi(2.0 * (double)j / (double)n0, &sn, &cs);
        const cplx dif = make_double2(zk.x - zc.x, zk.y - zc.y);
        // -(i/2) (cs - i sn) dif = (-(sn dif.x) + cs dif.y, -(cs dif.x) - sn dif.y) / 2
        cplx y = make_double2(0.5 * (zk.x + zc.x) + 0.5 * (cs * dif.y - sn * dif.x), 0.5 * (zk.y + zc.y) - 0.5 * (cs * dif.x + sn * dif.y));
        if (j == half) {
            y.x *= edge;
            y.y *= edge;
        }
        if (j == 0 || j == K) y.y = 0.0;
        return y;
    };
    // W[k] and W[K - k] are made of the same two bins: one lane does both (each spectrum value is read once)
    auto emit = [&](long long k, const cplx &yk, const cplx &yr) {
        const cplx yc = make_double2(yr.x, -yr.y);
        const cplx sum = make_double2(yk.x + yc.x, yk.y + yc.y), dif = make_double2(yk.x - yc.x, yk.y - yc.y);
        double sn, cs;
        sincospi(2.0 * (double)k / (double)num, &sn, &cs);
        // i (cs + i sn) dif = (-(sn dif.x) - cs dif.y, cs dif.x - sn dif.y)
        W[k] = make_double2((sum.x - (sn * dif.x + cs * dif.y)) * inv_n0, (sum.y + (cs * dif.x - sn * dif.y)) * inv_n0);
    };
    for (long long k = (long long)blockIdx.x * 256 + threadIdx.x; 2 * k <= K; k += (long long)gridDim.x * 256) {
        const cplx yk = bin(k);
        const cplx yr = bin(K - k);
        emit(k, yk, yr);
        if (k != 0 && K - k != k) emit(K - k, yr, yk);
    }
}

// ---------------------------------------------------------------------------------------------
// scipy.signal.resample (wefax.py:384) for ANY lengths (round 4).  A recording is as long as it is: n0 and num = int(11025 n0 / fs)
// are the reference's to choose, half of them are odd and almost none has 13-smooth halves -- rounds 1-3 sent those to Bluestein
// convolutions on power-of-two transforms of >= 2 n0 points (41 ms for a 60-minute 48 kHz capture less two samples, against 4.5
// for the whole-second length).  Here: two CHIRP-Z transforms on the mixed-radix passes, sized by what is actually needed.
//   forward   z[q] = x[2q] + i x[2q+1], q < L1 = ceil(n0 / 2) (any parity: x[n0] reads as zero);  the resampler keeps the bins
//             |k| <= h = min(n0, num) / 2 only, and   Z[k] = sum_q z[q] e^{-4 pi i q k / n0},  k in [-h, h],   separates into
//             E[k] = (Z[k] + conj Z[-k]) / 2,  O[k] = (Z[k] - conj Z[-k]) / 2i,  X[k] = E[k] + e^{-2 pi i k / n0} O[k]  (= rfft(x)[k]).
//             With c1[m] = e^{-2 pi i m^2 / n0}:  Z[k] = c1[k] sum_q (z[q] c1[q]) conj c1[k - q]: ONE cyclic convolution of
//             M1 >= L1 + 2h points (13-smooth, radix pairs) -- not 2 n0 - 1, and no power of two.
//   bins      Y[k] as scipy.signal.resample copies and scales them, extended to k in [-h, h] the way irfft reads them
//   inverse   y[2p] + i y[2p+1] = (1 / n0) sum_k Yh[k] (1 + i e^{2 pi i k / num}) e^{4 pi i k p / num},  p < P = ceil(num / 2):
//             with c2[m] = e^{2 pi i m^2 / num} again one cyclic convolution, of M2 >= P + 2h points.
// The chirps' arguments are reduced modulo n0 / num in integers first.  Both convolutions multiply by their kernel's transform
// (made once per length pair by these very passes) inside the last forward pass, and their last inverse pass stores only the
// outputs that are read.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ cplx czt_chirp(long long m, long long N, double sign)      // e^{sign 2 pi i m^2 / N}
{
    const unsigned long long a = (unsigned long long)(m < 0 ? -m : m);
    const unsigned long long r = (a * a) % (unsigned long long)N;
    double sn, cs;
    sincospi(2.0 * (double)r / (double)N, &sn, &cs);
    return make_double2(cs, sign * sn);
}

// kernel of the forward convolution: conj c1[m] / M1 at lag m in [-(h + L1 - 1), h], zero elsewhere
__global__ void __launch_bounds__(256) czt_fill_b1(cplx *__restrict__ B, long long n0, long long L1, long long h, long long M1, double inv_m)
{
    for (long long i = blockIdx.x * 256ll + threadIdx.x; i < M1; i += (long long)gridDim.x * 256ll) {
        cplx v = make_double2(0.0, 0.0);
        if (i <= h || M1 - i <= h + L1 - 1) {
            const cplx c = czt_chirp(i <= h ? i : i - M1, n0, +1.0);            // conj of e^{-..}
            v = make_double2(c.x * inv_m, c.y * inv_m);
        }
        B[i] = v;
    }
}

// kernel of the inverse convolution: conj c2[m + h] / M2 at lag m in [-2h, P - 1]
__global__ void __launch_bounds__(256) czt_fill_b2(cplx *__restrict__ B, long long num, long long P, long long h, long long M2, double inv_m)
{
    for (long long i = blockIdx.x * 256ll + threadIdx.x; i < M2; i += (long long)gridDim.x * 256ll) {
        cplx v = make_double2(0.0, 0.0);
        if (i <= P - 1 || M2 - i <= 2 * h) {
            const cplx c = czt_chirp((i <= P - 1 ? i : i - M2) + h, num, -1.0);
            v = make_double2(c.x * inv_m, c.y * inv_m);
        }
        B[i] = v;
    }
}

// A[q] = (x[2q] + i x[2q+1]) c1[q] for q < L1 (the first pass counts everything behind as zero)
template <typename T>
__global__ void __launch_bounds__(256) czt_prologue(const T *__restrict__ x, long long n0, long long L1, cplx *__restrict__ A)
{
    for (long long q = blockIdx.x * 256ll + threadIdx.x; q < L1; q += (long long)gridDim.x * 256ll) {
        const double a = (double)x[2 * q], b = 2 * q + 1 < n0 ? (double)x[2 * q + 1] : 0.0;
        const cplx c = czt_chirp(q, n0, -1.0);
        A[q] = make_double2(a * c.x - b * c.y, a * c.y + b * c.x);
    }
}

// the forward convolution's outputs C[k mod M1], |k| <= h  ->  the inverse convolution's input a2[h + k] = Yh[k] (1 + i e^{2 pi i k / num}) c2[k] / n0
__global__ void __launch_bounds__(256) czt_glue(const cplx *__restrict__ C, long long n0, long long num, long long h, long long M1, cplx *__restrict__ A2)
{
    const long long nmin = n0 < num ? n0 : num;
    const double edge = (nmin % 2 == 0) ? (num < n0 ? 2.0 : (num > n0 ? 0.5 : 1.0)) : 1.0;
    const double inv_n0 = 1.0 / (double)n0;
    for (long long k = blockIdx.x * 256ll + threadIdx.x; k <= h; k += (long long)gridDim.x * 256ll) {
        const cplx ck = czt_chirp(k, n0, -1.0);
        const cplx zk = mcmul(ck, C[k]), zm = mcmul(ck, C[k == 0 ? 0 : M1 - k]);           // Z[k], Z[-k]
        const cplx zc = make_double2(zm.x, -zm.y);
        const cplx E = make_double2(0.5 * (zk.x + zc.x), 0.5 * (zk.y + zc.y));
        const cplx O = make_double2(0.5 * (zk.y - zc.y), -0.5 * (zk.x - zc.x));                // (zk - zc) / 2i
        double sn, cs;
        sincospi(2.0 * (double)k / (double)n0, &sn, &cs);                                 // e^{-2 pi i k / n0} = cs - i sn
        cplx y = make_double2(E.x + cs * O.x + sn * O.y, E.y + cs * O.y - sn * O.x);           // X[k]
        if (nmin % 2 == 0 && k == h) {
            y.x *= edge;
            y.y *= edge;
        }
        cplx yp, ym;                                                                      // Yh[k], Yh[-k]
        if (k == 0) {
            yp = ym = make_double2(y.x, 0.0);                                             // irfft ignores Im Y[0]
        } else if (num % 2 == 0 && 2 * k == num) {
            yp = ym = make_double2(0.5 * y.x, 0.0);                                       // the output's Nyquist bin: real, once
        } else {
            yp = y;
            ym = make_double2(y.x, -y.y);
        }
        double s2, c2;
        sincospi(2.0 * (double)k / (double)num, &s2, &c2);                                // e^{2 pi i k / num} = c2 + i s2
        // 1 + i e^{+-i t} = (1 -+ s2) + i c2
        const cplx gp = mcmul(yp, make_double2(1.0 - s2, c2)), gm = mcmul(ym, make_double2(1.0 + s2, c2));
        const cplx q2 = czt_chirp(k, num, +1.0);
        const cplx ap = mcmul(gp, q2), am = mcmul(gm, q2);
        A2[h + k] = make_double2(ap.x * inv_n0, ap.y * inv_n0);
        if (k) A2[h - k] = make_double2(am.x * inv_n0, am.y * inv_n0);
    }
}

// y[2p] + i y[2p+1] = c2[p] C2[p]
__global__ void __launch_bounds__(256) czt_epilogue(const cplx *__restrict__ C2, long long num, long long P, double *__restrict__ out)
{
    for (long long p = blockIdx.x * 256ll + threadIdx.x; p < P; p += (long long)gridDim.x * 256ll) {
        const cplx u = mcmul(czt_chirp(p, num, +1.0), C2[p]);
        out[2 * p] = u.x;
        if (2 * p + 1 < num) out[2 * p + 1] = u.y;
    }
}

struct czt_cache {
    long long M1 = 0, M2 = 0;
    wfx_devbuf b1, b2;
    bool ready = false;
};
static std::map<std::pair<const void *, std::pair<long long, long long>>, czt_cache> g_czt;      // per (context, (n0, num)): one per context

static void czt_release(wfx_ctx *ctx)
{
    std::lock_guard<std::mutex> lock(g_mr_mutex);
    for (auto it = g_czt.begin(); it != g_czt.end();) {
        if (it->first.first == (const void *)ctx) {
            if (it->second.b1.p) (void)hipFree(it->second.b1.p);
            if (it->second.b2.p) (void)hipFree(it->second.b2.p);
            it = g_czt.erase(it);
        } else {
            ++it;
        }
    }
}

// one cyclic convolution on the plan's passes: src (M points, natural order) * table, ping-ponging between A and B; the last inverse
// pass stores the outputs outside (skip_lo, skip_hi) only
static int czt_convolve(wfx_ctx *ctx, mr_plan_cache *pc, const cplx *table, const cplx *src, long long in_len, cplx *A, cplx *B, long long skip_lo,
                        long long skip_hi, cplx **result)
{
    const cplx *tb = (const cplx *)pc->tables.p;
    const int np = pc->h.npass;
    cplx *dst = (src == A) ? B : A;
    for (int i = 0; i < np; ++i) {
        mr_pass_desc d = pc->h.pass[i];
        if (i == 0) d.in_len = in_len;                               // the points behind the input count as zero: never written, never read
        if (i == np - 1) d.gtab = (const double2 *)table;
        WFX_TRY(wfx_mr_launch_pair(ctx, d, tb + pc->h.lo_off[i], 0, i == np - 1 ? 4 : 0, 0, src, dst));
        src = dst;
        dst = dst == A ? B : A;
    }
    return mr_run(ctx, pc, src, A, B, 1, false, nullptr, result, false, skip_lo, skip_hi);
}

static czt_cache *czt_lookup(wfx_ctx *ctx, long long n0, long long num)
{
    const long long L1 = (n0 + 1) / 2, P = (num + 1) / 2, h = (n0 < num ? n0 : num) / 2;
    czt_cache *cz = nullptr;
    {
        std::lock_guard<std::mutex> lock(g_mr_mutex);
        auto key = std::make_pair((const void *)ctx, std::make_pair(n0, num));
        auto it = g_czt.find(key);
        if (it == g_czt.end()) {
            for (auto jt = g_czt.begin(); jt != g_czt.end();) {      // (one pair of kernels per context)
                if (jt->first.first == (const void *)ctx) {
                    (void)hipStreamSynchronize(ctx->stream);
                    if (jt->second.b1.p) (void)hipFree(jt->second.b1.p);
                    if (jt->second.b2.p) (void)hipFree(jt->second.b2.p);
                    jt = g_czt.erase(jt);
                } else {
                    ++jt;
                }
            }
            czt_cache c;
            c.M1 = wfx_mr_padded_length(L1 + 2 * h);
            c.M2 = wfx_mr_padded_length(P + 2 * h);
            it = g_czt.emplace(key, c).first;
        }
        cz = &it->second;
    }
    return cz;
}

// whether wfx_dev_resample_czt takes these lengths (the decode then hands it an int16 capture as it is)
bool wfx_czt_resample_supported(wfx_ctx *ctx, uint64_t n0u, uint64_t numu)
{
    if (n0u < 8192 || numu < 8192 || n0u >= (1ull << 31) || numu >= (1ull << 31) || getenv("WFX_NO_CZT")) return false;
    const czt_cache *cz = czt_lookup(ctx, (long long)n0u, (long long)numu);
    return cz->M1 != 0 && cz->M2 != 0;
}

int wfx_dev_resample_czt(wfx_ctx *ctx, const void *x, bool x_is_i16, uint64_t n0u, uint64_t numu, double *out, int *handled)
{
    *handled = 0;
    if (n0u < 8192 || numu < 8192 || n0u >= (1ull << 31) || numu >= (1ull << 31)) return 0;
    const long long n0 = (long long)n0u, num = (long long)numu;
    const long long L1 = (n0 + 1) / 2, P = (num + 1) / 2, h = (n0 < num ? n0 : num) / 2;
    czt_cache *cz = czt_lookup(ctx, n0, num);
    if (cz->M1 == 0 || cz->M2 == 0) return 0;
    const long long M1 = cz->M1, M2 = cz->M2;
    mr_plan_cache *p1 = nullptr, *p2 = nullptr;
    WFX_TRY(mr_get_plan(ctx, M1, &p1));
    WFX_TRY(mr_get_plan(ctx, M2, &p2));
    for (mr_plan_cache *pc : {p1, p2}) {
        if (pc->h.npass < 2 || !pc->use_mr2) return 0;
        for (int i = 0; i < pc->h.npass; ++i)
            if (pc->h.pass[i].ra <= 0) return 0;
    }
    const size_t cap = (size_t)(M1 > M2 ? M1 : M2) * sizeof(cplx);
    WFX_TRY(wfx_reserve(ctx, ctx->b_work, cap));
    WFX_TRY(wfx_reserve(ctx, ctx->b_work2, cap));
    cplx *A = (cplx *)ctx->b_work.p, *B = (cplx *)ctx->b_work2.p;
    if (!cz->ready) {
        WFX_TRY(wfx_reserve(ctx, cz->b1, (size_t)M1 * sizeof(cplx)));
        WFX_TRY(wfx_reserve(ctx, cz->b2, (size_t)M2 * sizeof(cplx)));
        cplx *res = nullptr;
        WFX_LAUNCH(ctx, K_BS_CHIRP, czt_fill_b1, dim3(wfx_stream_grid((uint64_t)M1, 256)), dim3(256), (cplx *)cz->b1.p, n0, L1, h, M1, 1.0 / (double)M1);
        WFX_TRY(mr_run(ctx, p1, (cplx *)cz->b1.p, A, B, 0, false, (cplx *)cz->b1.p, &res));
        WFX_LAUNCH(ctx, K_BS_CHIRP, czt_fill_b2, dim3(wfx_stream_grid((uint64_t)M2, 256)), dim3(256), (cplx *)cz->b2.p, num, P, h, M2, 1.0 / (double)M2);
        WFX_TRY(mr_run(ctx, p2, (cplx *)cz->b2.p, A, B, 0, false, (cplx *)cz->b2.p, &res));
        cz->ready = true;
    }
    if (x_is_i16)
        WFX_LAUNCH(ctx, K_RESAMPLE_PW, czt_prologue<short>, dim3(wfx_stream_grid((uint64_t)L1, 256)), dim3(256), (const short *)x, n0, L1, A);
    else
        WFX_LAUNCH(ctx, K_RESAMPLE_PW, czt_prologue<double>, dim3(wfx_stream_grid((uint64_t)L1, 256)), dim3(256), (const double *)x, n0, L1, A);
    cplx *C1 = nullptr;
    WFX_TRY(czt_convolve(ctx, p1, (const cplx *)cz->b1.p, A, L1, A, B, h, M1 - h, &C1));      // outputs [0, h] and [M1 - h, M1)
    cplx *A2 = C1 == A ? B : A;
    WFX_LAUNCH(ctx, K_RESAMPLE_PW, czt_glue, dim3(wfx_stream_grid((uint64_t)h + 1, 256)), dim3(256), (const cplx *)C1, n0, num, h, M1, A2);
    cplx *C2 = nullptr;
    WFX_TRY(czt_convolve(ctx, p2, (const cplx *)cz->b2.p, A2, 2 * h + 1, A, B, P - 1, M2, &C2));  // outputs [0, P)
    WFX_LAUNCH(ctx, K_RESAMPLE_PW, czt_epilogue, dim3(wfx_stream_grid((uint64_t)P, 256)), dim3(256), (const cplx *)C2, num, P, out);
    *handled = 1;
    return 0;
}

bool wfx_mr_resample_supported(uint64_t n0, uint64_t num)
{
    if ((n0 & 1) || (num & 1) || n0 < 8 || num < 8 || n0 >= (1ull << 31) || num >= (1ull << 31)) return false;
    return wfx_mr_supported(n0 / 2) && wfx_mr_supported(num / 2);
}

int wfx_dev_resample_mr(wfx_ctx *ctx, const double *x, uint64_t n0, uint64_t num, double *out, bool x_is_i16)
{
    const long long M = (long long)(n0 / 2), K = (long long)(num / 2);
    mr_plan_cache *p1 = nullptr, *p2 = nullptr;
    WFX_TRY(mr_get_plan(ctx, M, &p1));
    const size_t cap = (size_t)(M > K ? M : K) * sizeof(cplx);
    WFX_TRY(wfx_reserve(ctx, ctx->b_work, cap));
    WFX_TRY(wfx_reserve(ctx, ctx->b_work2, cap));
    cplx *A = (cplx *)ctx->b_work.p, *B = (cplx *)ctx->b_work2.p;
    cplx *Z = nullptr;
    // (an int16 capture is read in place by the first pass: no float64 copy of the input)
    // down-sampling reads only the bins [0, num/2] and their mirrors [M - num/2, M) of the forward spectrum: the rest is not stored
    const long long nmin = (long long)(n0 < num ? n0 : num), half = nmin / 2;
    WFX_TRY(mr_run(ctx, p1, (const cplx *)x, A, B, 0, false, nullptr, &Z, x_is_i16, half, M - half));
    cplx *Wb = (Z == A) ? B : A;
    WFX_LAUNCH(ctx, K_RESAMPLE_PW, resample_mr_glue, dim3(wfx_stream_grid((uint64_t)K / 2 + 1, 256)), dim3(256), (const cplx *)Z, (long long)n0,
               (long long)num, Wb, wfx_nt_for(8.0 * (double)nmin));
    WFX_TRY(mr_get_plan(ctx, K, &p2));        // (std::map: p1 stays valid)
    cplx *res = nullptr;
    return mr_run(ctx, p2, Wb, A, B, 1, false, (cplx *)out, &res);
}
