/*
 * ORACLE (test infrastructure only) -- sequential loops of the WEFAX hot path
 * restated in plain C.  Nothing under oracle/ is part of the shipped product:
 * only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load
 * this library, and only as the checker.
 *
 * Built by oracle/Makefile with  gcc -O2 -ffp-contract=off  (no FMA
 * contraction, no fast-math) so that every double operation rounds exactly
 * like the NumPy/SciPy code it restates.
 */
#include <stddef.h>
#include <stdint.h>

/*
 * scipy.signal.lfilter for one biquad (len(b) == len(a) == 3, a[0] == 1) with
 * initial state zi -- the call that scipy.signal.filtfilt makes twice for
 * /root/reference/wefax.py:72.  Transposed direct form II, the same recurrence
 * and operation order as SciPy's lfilter inner loop for doubles:
 *     y   = z0 + b0*x
 *     z0' = z1 + b1*x - a1*y
 *     z1' =      b2*x - a2*y
 */
void wfo_lfilter_biquad(const double *b, const double *a, const double *x,
                        double *y, size_t n, const double *zi)
{
    double z0 = zi[0], z1 = zi[1];
    const double b0 = b[0], b1 = b[1], b2 = b[2], a1 = a[1], a2 = a[2];
    for (size_t i = 0; i < n; ++i) {
        const double xi = x[i];
        const double yi = z0 + b0 * xi;
        z0 = z1 + b1 * xi - a1 * yi;
        z1 = b2 * xi - a2 * yi;
        y[i] = yi;
    }
}

/*
 * Sequential peak picker of /root/reference/wefax.py:226-261 ("pattern_search").
 * corr[i], i in [0, ncorr), is the sliding correlation of wefax.py:236.
 *   peaks = [(0, 0)]
 *   for i: if i - peaks[-1].pos > mindistance: append (i, corr[i])   (wefax.py:238-240)
 *          elif corr[i] > peaks[-1].val:       replace last          (wefax.py:247-249)
 *          if len(peaks) == max_peaks: break                         (wefax.py:251-259)
 * peak_pos receives the final positions, first_pos the index i at which each
 * peak was appended (the reference reports progress with it, wefax.py:245).
 * Returns the number of peaks (>= 1); *hit_limit is 1 when the loop broke.
 */
int wfo_pick_peaks(const int64_t *corr, size_t ncorr, int64_t mindistance,
                   int max_peaks, int64_t *peak_pos, int64_t *first_pos,
                   int *hit_limit)
{
    int np_ = 1;
    int64_t pos = 0, val = 0;
    peak_pos[0] = 0;
    first_pos[0] = 0;
    *hit_limit = 0;
    for (size_t i = 0; i < ncorr; ++i) {
        const int64_t c = corr[i];
        if ((int64_t)i - pos > mindistance) {
            peak_pos[np_ - 1] = pos;
            pos = (int64_t)i;
            val = c;
            first_pos[np_] = pos;
            ++np_;
        } else if (c > val) {
            pos = (int64_t)i;
            val = c;
        }
        if (np_ == max_peaks) {
            *hit_limit = 1;
            break;
        }
    }
    peak_pos[np_ - 1] = pos;
    return np_;
}

/*
 * Vertical pass of Pillow's ImagingResample for an 8-bit image
 * (ImagingResampleVertical_8bpc): out[yy][x] = clip8((2^21 + sum_k in[ymin+k][x]
 * * kk[yy][k]) >> 22).  bounds[2*yy] = ymin, bounds[2*yy+1] = tap count.
 * This is what Image.resize((w, 4*h)) does at /root/reference/wefax.py:325.
 */
void wfo_resize_vertical_8bpc(const uint8_t *in, int w, int h_in, uint8_t *out,
                              int h_out, const int32_t *kk, int ksize,
                              const int32_t *bounds)
{
    (void)h_in;
    for (int yy = 0; yy < h_out; ++yy) {
        const int32_t *k = kk + (size_t)yy * ksize;
        const int ymin = bounds[2 * yy], ymax = bounds[2 * yy + 1];
        for (int x = 0; x < w; ++x) {
            int32_t ss = 1 << 21;
            for (int y = 0; y < ymax; ++y)
                ss += (int32_t)in[(size_t)(y + ymin) * w + x] * k[y];
            int32_t v = ss >> 22;
            out[(size_t)yy * w + x] = (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v));
        }
    }
}
