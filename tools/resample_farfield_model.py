"""Design study (NumPy only, no kernels): `scipy.signal.resample` (wefax.py:384) WITHOUT a transform over the whole capture -- the second
global operator of the path, gated like the Hilbert transform's multipole form (tools/farfield_model.py) before anything is built on it.

scipy's FFT resampler of a real signal x[0..N0) to num samples is y[k] = sum_n x[n] D(k/num - n/N0) with a periodic-sinc kernel.  Its
numerator SEPARATES into a factor of the source and a factor of the target, which leaves the SAME cotangent kernel the Hilbert transform has,
between two grids that do not coincide (u = k/num - n/N0, period 1):

  down-sampling (num < N0), num even (scipy doubles the new Nyquist bin):
      D(u) = [sin(pi num u) cot(pi u) + cos(pi num u)] / N0,   sin(pi num u) = -(-1)^k s_n,  cos(pi num u) = (-1)^k c_n,
      s_n = sin(pi num n / N0), c_n = cos(pi num n / N0)
      =>  y[k] = (-1)^k / N0 * [ C - sum_n (x_n s_n) cot(pi u_kn) ],   C = sum_n x_n c_n          (one number for the whole capture)
  up-sampling (num > N0), N0 even (scipy halves the old Nyquist bin):
      D(u) = sin(pi N0 u) cot(pi u) / N0,   sin(pi N0 u) = (-1)^n t_k,  t_k = sin(pi N0 k / num)
      =>  y[k] = t_k / N0 * sum_n ((-1)^n x_n) cot(pi u_kn)
  pairs with u = 0 exactly (k N0 = n num) are taken out of the sum and given the kernel's limit there: (num + 1) / N0, resp. 1.
  num odd when down-sampling / N0 odd when up-sampling: no Nyquist bin to adjust, the kernel is sin(.) / sin(pi u) -- the same separation
  with a COSECANT in the cotangent's place, anti-periodic round the circle: an interaction that wraps the seam w times carries (-1)^w.

So the resampler is: modulate the sources, a one-dimensional fast multipole sum with the cotangent kernel on the unit circle (near field: the
target's leaf and its two neighbours, directly; far field: P2M, M2M, M2L, L2L, L2P on p Chebyshev nodes -- the tree, its matrices and its
kernels are the Hilbert transform's), multiply by the target factor.  What a rank of a sharded decode would exchange is what the Hilbert
form exchanges (KBs) plus ONE all-reduced number.  On the device the near field is a polyphase filter: for a rational rate (441 / 640 from
16 kHz, 147 / 640 from 48 kHz) the taps of a target depend on k mod 441 (147) only.

    python tools/resample_farfield_model.py            # the gate: max relative error against scipy.signal.resample
"""
from __future__ import annotations

import numpy as np

from farfield_model import _cheb_basis, _cheb_nodes


def _exact_sin_cos(mult: int, idx: np.ndarray, den: int):
    """sin and cos of pi * mult * idx / den, RELATIVELY accurate where they are small: the angle is reduced in integers to r' / den with
    |r'| <= den / 2 (sin) -- a sine taken of an argument near a multiple of pi has lost the digits of its distance from it, and s_n is
    multiplied by cotangents of 1e9 exactly where it is small."""
    prod = mult * idx.astype(np.int64)                                        # (< 2^62 for the sizes of this study)
    r = prod % (2 * den)
    j = (2 * r + den) // (2 * den)                                            # nearest multiple of den
    rs = (r - j * den).astype(np.float64)
    sgn = np.where(j % 2 == 0, 1.0, -1.0)
    s = sgn * np.sin(np.pi * rs / den)
    # cosine: reduce to the nearest odd multiple of den / 2 likewise
    c = sgn * np.cos(np.pi * rs / den)
    return s, c


class CotFMM:
    """S[k] = sum_n w[n] cot(pi (b_k - a_n)) over pairs with b_k != a_n, positions on the unit circle given as exact fractions
    a_n = n / N0 (sources), b_k = k / num (targets)."""

    def __init__(self, n0: int, num: int, p: int = 16, leaf_level: int | None = None, kind: str = "cot"):
        # kind "csc": 1 / sin(pi u) instead -- ANTI-periodic round the circle: an image shifted by w whole circles carries the sign (-1)^w
        self.n0, self.num, self.p, self.kind = n0, num, p, kind
        if leaf_level is None:
            leaf_level = max(3, int(np.floor(np.log2(max(max(n0, num) / 48.0, 8.0)))))
        self.lmax = leaf_level
        c = _cheb_nodes(p)
        self.c = c
        self.m2m = [_cheb_basis((c - 1) / 2, p), _cheb_basis((c + 1) / 2, p)]

    def apply(self, w: np.ndarray):
        n0, num, p, lmax = self.n0, self.num, self.p, self.lmax
        nleaf = 1 << lmax
        src = np.arange(n0, dtype=np.int64)
        tgt = np.arange(num, dtype=np.int64)
        # leaf of a position i / N: floor(i * nleaf / N), in integers
        leaf_s = (src * nleaf) // n0
        leaf_t = (tgt * nleaf) // num
        us = 2.0 * ((src * nleaf - leaf_s * n0) / n0) - 1.0                     # local coordinate in [-1, 1)
        ut = 2.0 * ((tgt * nleaf - leaf_t * num) / num) - 1.0
        W = np.zeros((nleaf, p))
        step = 1 << 18
        for a in range(0, n0, step):
            b = min(n0, a + step)
            np.add.at(W, leaf_s[a:b], _cheb_basis(us[a:b], p) * w[a:b, None])
        ws = {lmax: W}
        for lev in range(lmax - 1, 1, -1):
            ch = ws[lev + 1]
            ws[lev] = ch[0::2] @ self.m2m[0] + ch[1::2] @ self.m2m[1]
        loc = None
        for lev in range(2, lmax + 1):
            nb = 1 << lev
            cur = np.zeros((nb, p))
            if loc is not None:
                cur[0::2] = loc @ self.m2m[0].T
                cur[1::2] = loc @ self.m2m[1].T
            idx = np.arange(nb)
            offsets = {0: (-2, 2, 3), 1: (-3, -2, 2)}
            for par in (0, 1):
                t = idx[par::2]
                seen = [set() for _ in t] if nb <= 8 else None
                for k in offsets[par]:
                    sidx = (t + k) % nb
                    if nb <= 8:
                        keep = []
                        for q, (tb, sb_) in enumerate(zip(t, sidx)):
                            dist = min((sb_ - tb) % nb, (tb - sb_) % nb)
                            ok = dist >= 2 and sb_ not in seen[q]
                            if ok:
                                seen[q].add(int(sb_))
                            keep.append(ok)
                        keep = np.array(keep)
                        if not keep.any():
                            continue
                        tt, ss = t[keep], sidx[keep]
                    else:
                        tt, ss = t, sidx
                    raw = tt - ss
                    for r in np.unique(raw):
                        sel = raw == r
                        z = (r + (self.c[:, None] - self.c[None, :]) / 2) / nb           # difference of the node positions, in circles
                        wz = np.rint(z)
                        z = z - wz
                        g = 1.0 / np.tan(np.pi * z) if self.kind == "cot" else np.where(wz.astype(np.int64) % 2 == 0, 1.0, -1.0) / np.sin(np.pi * z)
                        cur[tt[sel]] += ws[lev][ss[sel]] @ g.T
            loc = cur
        far = np.empty(num)
        for a in range(0, num, step):
            b = min(num, a + step)
            far[a:b] = np.einsum("ij,ij->i", _cheb_basis(ut[a:b], p), loc[leaf_t[a:b]])
        # near field: the target's leaf and its two neighbours; u from integers: (k N0 - n num) / (num N0), reduced to the nearest image
        near = np.zeros(num)
        s_start = np.searchsorted(leaf_s, np.arange(nleaf + 1))
        t_start = np.searchsorted(leaf_t, np.arange(nleaf + 1))
        cnt_s, cnt_t = np.diff(s_start), np.diff(t_start)
        ms, mt = int(cnt_s.max()), int(cnt_t.max())
        big = num * n0
        blk = max(1, (1 << 22) // max(1, mt * 3 * ms))
        for a in range(0, nleaf, blk):
            b = min(nleaf, a + blk)
            leaves = np.arange(a, b)
            ti = t_start[leaves][:, None] + np.arange(mt)[None, :]
            tv = np.arange(mt)[None, :] < cnt_t[leaves][:, None]
            ti = np.where(tv, ti, 0)
            acc = np.zeros((b - a, mt))
            for dk in (-1, 0, 1):
                sl = (leaves + dk) % nleaf
                si = s_start[sl][:, None] + np.arange(ms)[None, :]
                sv = np.arange(ms)[None, :] < cnt_s[sl][:, None]
                si = np.where(sv, si, 0)
                m = ti[:, :, None] * n0 - si[:, None, :] * num                       # exact integers (|m| < 2^62 for the sizes of this study)
                wm = np.rint(m / big).astype(np.int64)
                m = m - big * wm
                ok = sv[:, None, :] & (m != 0)
                with np.errstate(divide="ignore"):
                    arg = np.pi * (np.where(ok, m, 1) / big)
                    kk = np.where(ok, 1.0 / np.tan(arg) if self.kind == "cot" else np.where(wm % 2 == 0, 1.0, -1.0) / np.sin(arg), 0.0)
                acc += np.einsum("lts,ls->lt", kk, np.where(sv, w[si], 0.0))
            near[ti[tv]] += acc[tv]
        return far + near


def resample_fmm(x, num: int, p: int = 16):
    """scipy.signal.resample(x, num) for a real x by the multipole form (any lengths)."""
    x = np.asarray(x, dtype=np.float64)
    n0 = x.shape[0]
    if num == n0:
        return x.copy()
    k = np.arange(num, dtype=np.int64)
    n = np.arange(n0, dtype=np.int64)
    g = int(np.gcd(n0, num))
    pn, pk = num // g, n0 // g                        # coincident pairs: k = j * pn... k N0 = n num  <=>  k = j num/g, n = j N0/g, j = 0 .. g-1
    jn, jk = np.arange(g, dtype=np.int64) * pk, np.arange(g, dtype=np.int64) * pn
    if num < n0:
        s, c = _exact_sin_cos(num, n, n0)
        sign = np.where(k % 2 == 0, 1.0, -1.0)
        if num % 2:
            # odd count: D(u) = sin(pi num u) / (N0 sin(pi u)) = -(-1)^k s_n csc(pi u) / N0, D(0) = num / N0
            y = -sign / n0 * CotFMM(n0, num, p, kind="csc").apply(x * s)
            y[jk] += x[jn] * (num / n0)
            return y
        tot = CotFMM(n0, num, p).apply(x * s)
        big_c = float(np.sum(x * c))
        y = sign / n0 * (big_c - tot)
        # pairs at u = 0: the sum skipped them (s_n is an exact zero there anyway); D(0) = (num + 1) / N0, of which the C term already carries
        # (-1)^k c_n / N0 = 1 / N0 (cos(pi num u) = 1 at u = 0)
        y[jk] += x[jn] * (num / n0)
        return y
    t, _ = _exact_sin_cos(n0, k, num)
    alt = np.where(n % 2 == 0, 1.0, -1.0)
    # even N0: D(u) = sin(pi N0 u) cot(pi u) / N0; odd N0: sin(pi N0 u) csc(pi u) / N0; D(0) = 1 either way (the interpolant goes through the samples)
    y = t / n0 * CotFMM(n0, num, p, kind="cot" if n0 % 2 == 0 else "csc").apply(x * alt)
    y[jk] += x[jn]
    return y


if __name__ == "__main__":
    import os
    import sys
    import time
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from scipy.signal import resample
    from wefax_amd import synth
    kw = dict(start_tone_s=2.0, phasing_lines=20, image_lines=30, stop_tone_s=1.0, black_tail_s=2.0)
    for fs in (48000, 16000, 8000):
        x = synth.synth_capture(float(fs), noise=0.05, seed=3, **kw).astype(np.float64)
        num = int(11025 * (x.shape[0] / fs))
        ref = resample(x, num)
        for p in (12, 16):
            t0 = time.perf_counter()
            got = resample_fmm(x, num, p)
            err = np.max(np.abs(got - ref)) / np.max(np.abs(ref))
            print(f"{fs} Hz -> 11 025 Hz, {x.shape[0]} -> {num} samples, p = {p}: max relative error {err:.3e}   ({time.perf_counter() - t0:.1f} s)", flush=True)
