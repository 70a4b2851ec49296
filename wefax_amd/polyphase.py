"""Time-domain sample-rate front end for oversampled captures (BASELINE configs[3]).

The reference brings every capture to 11 025 Hz with ``scipy.signal.resample``
(/root/reference/wefax.py:375-394): an FFT over the WHOLE capture.  That operator is global,
so it neither shards by sample range nor fits a 60-minute 1.536 MS/s IQ stream (5.5 G pairs).
This module is the halo-local counterpart SURVEY.md section 8e asks for: a chain of FIR DECIMATORS

    [ decimate by a power of two <= 64 ]*  ->  [ decimate by 2, 3, 5 or 6 ]  ->  hand-over rate  ->  (exact FFT resampler)

down to a hand-over rate that divides the capture's rate and lies above 11 025 Hz (1.536 MS/s: /32 -> 48 kHz, /3 -> 16 kHz;
192 kHz: /4, /3; 48 kHz: /3; 44.1 kHz: /3 -> 14 700 Hz), where the exact FFT resampler of the decode path takes the last step --
the reference's own brick wall at 5512.5 Hz.  The kernels live in ``csrc/wfx_polyphase.hip`` and carry NO rounding error of
their own where that is possible: the ingest of an int16 capture is an integer dot product on a 2**-30 tap grid, everything
behind it float64 (round 3).  The fp32 chains of rounds 1-2 (a rational x147/160 stage, hand-overs at 22 050 / 14 700 Hz from
48 kHz, an all-time-domain chain to 11 025 Hz) are gone since round 4: arithmetic narrower than the reference's has no place
next to an exact path.  Filters: linear phase centred on the output sample (zero delay), unit DC gain, pass band to
5512.5 Hz, stop bands wherever something would alias into 0..5512.5 Hz at the stage's output rate.  The stereo / IQ merge of
wefax.py:360-373 is fused into the first stage's loads.

The reference's sampling grid -- output j at input position j * n0 / int(11025 * n0 / fs_in), one period = the whole
capture -- is kept exactly: the chain hands over n0 * out_rate / fs_in samples (lengths for which that is not a whole number
are refused, ``FrontEnd.n_out``) and the exact FFT resampler behind it delivers int(11025 * n0 / fs_in) samples like
wefax.py:384, whole seconds or not.
"""
from __future__ import annotations

import os
from fractions import Fraction

import numpy as np

TARGET_RATE = 11025
# Stop-band attenuation of every stage.  What separates a decode behind this front end from the reference is what aliases into
# 0..5512.5 Hz: measured on a 40-s noisy 1.536 MS/s IQ clip (tools/fe_att_sweep.py, fp32 kernels) -- 90 dB: 1293 of 441 000 stream
# bytes differ by 1 and 9 pixels by 2; 105 dB: 276 / 1; 120 dB: 45 / 1; 135 dB: 19 / 0 (image max |delta| 1); 150 dB: 16 / 0, where
# fp32 accumulation is the floor and the first stage turns compute-bound (+1.9 ms on the 60-minute stream against +0.5 ms at 135).
DEFAULT_ATT_DB = 135.0
LOW_RATE_ATT_DB = 200.0      # Kaiser stages at <= 192 kHz of the float64 chains (FrontEnd._finish)
NYQ = TARGET_RATE / 2.0          # 5512.5 Hz: everything the reference's brick wall keeps


def kaiser_beta(att_db: float) -> float:
    if att_db > 50:
        return 0.1102 * (att_db - 8.7)
    if att_db >= 21:
        return 0.5842 * (att_db - 21) ** 0.4 + 0.07886 * (att_db - 21)
    return 0.0


def kaiser_length(att_db: float, df_norm: float) -> int:
    """Taps for a transition band of df_norm cycles/sample (Kaiser's estimate)."""
    return int(np.ceil((att_db - 7.95) / (2.285 * 2 * np.pi * df_norm))) + 1


def _kaiser_window(u: np.ndarray, half: float, beta: float) -> np.ndarray:
    """Continuous Kaiser window of half-width ``half`` evaluated at offsets u."""
    r = 1.0 - (u / half) ** 2
    return np.where(r > 0, np.i0(beta * np.sqrt(np.maximum(r, 0.0))) / np.i0(beta), 0.0)


_PAIR_CACHE: dict = {}


def _response(h: np.ndarray, fs: float, f) -> np.ndarray:
    """Zero-phase response of the odd-length symmetric FIR h at the frequencies f (Hz)."""
    n = h.shape[0]
    om = 2 * np.pi * np.asarray(f, dtype=np.float64) / fs
    return (np.exp(-1j * np.outer(om, np.arange(n) - (n - 1) // 2)) @ h).real


def ls_fir(n: int, fs: float, bands, desired, weights, grid: int = 400, iters: int = 10) -> np.ndarray:
    """Odd-length linear-phase FIR by weighted least squares on ``bands`` (Hz), everything else a don't-care region; a few
    Lawson reweightings flatten the error towards equiripple.  ``desired`` per band: a number or a function of frequency."""
    k = (n - 1) // 2
    fl, dl, wl = [], [], []
    for (lo, hi), dd, w in zip(bands, desired, weights):
        f = np.linspace(lo, hi, grid)
        fl.append(f)
        dl.append(dd(f) if callable(dd) else np.full_like(f, dd))
        wl.append(np.full_like(f, w))
    f, d, w0 = np.concatenate(fl), np.concatenate(dl), np.concatenate(wl)
    a = np.cos(np.outer(2 * np.pi * f / fs, np.arange(k + 1)))
    a[:, 1:] *= 2
    w = w0.copy()
    for _ in range(iters):
        x = np.linalg.lstsq(a * w[:, None], d * w, rcond=None)[0]
        e = np.abs(a @ x - d) * w0
        w = w * (e / e.max() + 1e-3) ** 0.5
        w /= w.max()
    return np.concatenate([x[:0:-1], x])


MIN_FIX_SHIFT = 26      # coarser grids than 2**-26 are not worth it: the float64 form runs instead
FIX_LB = 12             # csrc/wfx_polyphase.hip PP_FIX_LB: a fixed-point tap is hi * 2**12 + lo, two int16 halves


def fix_shift_for(h: np.ndarray, factor: int | None = None) -> int:
    """The finest grid 2**-s (s <= 30) the integer-exact stencil (csrc/wfx_polyphase.hip MODE 1) takes these taps on: the high half
    of every tap an int16, and -- the kernel moves both halves' int32 sums into an int64 every `flush` polyphase rows -- the high
    halves (and the low halves) of the taps of SOME power-of-two number of rows together below 2**16: times the largest int16
    sample that still fits one int32.  The rows a tap lands in depend on where a slice starts (the kernel aligns windows to 16
    bytes: up to 7 samples of shift), so every shift is tried.  Without ``factor``: all taps in one window (the round-3 first
    form of the kernel; grids to 2**-27).  0: none (the float64 form runs instead)."""
    h = np.asarray(h, dtype=np.float64)
    lim = 0.98 * (1 << 16)
    for s in range(30, 15, -1):
        v = np.rint(np.ldexp(h, s))
        hi = np.floor((v + (1 << (FIX_LB - 1))) / (1 << FIX_LB))
        lo = v - hi * (1 << FIX_LB)
        if np.max(np.abs(v)) >= (1 << 27) - (1 << FIX_LB) - 2:
            continue
        if factor is None:
            if np.max(np.abs(v)) < (1 << 23) - 2 and np.sum(np.abs(hi)) + h.shape[0] < lim:
                return s
            continue
        M = int(factor)
        ok_all = True
        for d in range(8):
            rows_hi, rows_lo = np.zeros(M), np.zeros(M)
            idx = (np.arange(h.shape[0]) + d) % M
            np.add.at(rows_hi, idx, np.abs(hi))
            np.add.at(rows_lo, idx, np.abs(lo))
            per_row = -(-(h.shape[0] + d) // M)
            fits = False
            fr = M
            while fr >= 1 and not fits:
                wh = rows_hi.reshape(-1, fr).sum(axis=1).max() + fr * per_row
                wl = rows_lo.reshape(-1, fr).sum(axis=1).max() + fr * per_row
                fits = wh < lim and wl < lim
                fr //= 2
            ok_all &= fits
        if ok_all:
            return s
    return 0


def quantize_taps(h: np.ndarray, shift: int) -> np.ndarray:
    """Taps rounded to multiples of 2**-shift with first-order error feedback: the rounding error of one tap is carried into
    the next, so the error sequence is a first difference and its spectrum vanishes at DC -- across 0..5512.5 Hz of a 1.536 MS/s
    filter the response moves by 2e-10 instead of the 2e-8 plain rounding costs (the stop bands lose 1 dB at -132).  The sum of
    the taps is kept exactly."""
    out = np.empty(h.shape[0], dtype=np.float64)
    e = 0.0
    sc = float(1 << shift)
    for k, v in enumerate(np.asarray(h, dtype=np.float64)):
        t = v * sc + e
        r = np.rint(t)
        e = t - r
        out[k] = r / sc
    return out


class Decimate:
    """y[k] = sum_j c[j] x[k*M - centre + j]: low-pass + keep every M-th sample."""
    kind = "decimate"

    def __init__(self, fs_in: Fraction, factor: int, pass_hz: float, stop_hz: float, att_db: float):
        self.fs_in, self.fs_out, self.factor = fs_in, fs_in / factor, factor
        fs = float(fs_in)
        n = kaiser_length(att_db, (stop_hz - pass_hz) / fs) | 1           # odd: the centre tap sits on the output sample
        self.centre = (n - 1) // 2
        u = np.arange(n, dtype=np.float64) - self.centre
        fc = (pass_hz + stop_hz) / 2 / fs                                   # cycles per input sample
        h = 2 * fc * np.sinc(2 * fc * u) * _kaiser_window(u, self.centre + 1.0, kaiser_beta(att_db))
        self.coef64 = h / h.sum()
        self.ntaps = n

    fix_shift = 0            # > 0: coef64 lies on the grid 2**-fix_shift (integer-exact ingest)

    def set_taps(self, h: np.ndarray, fix_shift: int = 0):
        """Replace the Kaiser design by another odd-length symmetric one (same stage geometry otherwise)."""
        assert h.shape[0] % 2 == 1
        self.fix_shift = fix_shift
        self.ntaps = int(h.shape[0])
        self.centre = (self.ntaps - 1) // 2
        self.coef64 = np.asarray(h, dtype=np.float64)

    def in_range(self, a: int, b: int):
        """Input index range needed for outputs [a, b)."""
        return a * self.factor - self.centre, (b - 1) * self.factor - self.centre + self.ntaps


LAST_FACTORS = (1, 2, 3, 5, 6)     # what may be left for the last stage once the powers of two are taken out
MIN_HANDOVER = 14000               # Hz: the last filter's transition band (5512.5 .. rate - 5512.5) stays wide


def stage_factors(ratio: int):
    """``ratio`` = fs_in / hand-over rate as a list of stage factors: powers of two (<= 64 each) first, then whatever is left
    (2, 3, 5 or 6); None when the ratio has no such form."""
    out, rem = [], int(ratio)
    while rem > 6 and rem % 2 == 0:
        f = 2
        while f < 64 and rem % (2 * f) == 0:
            f *= 2
        out.append(f)
        rem //= f
    if rem not in LAST_FACTORS:
        return None
    if rem > 1:
        out.append(rem)
    return out or None


class FrontEnd:
    """Stage chain from ``fs_in`` to the hand-over rate and the index bookkeeping around it."""

    def __init__(self, fs_in: int, att_db: float | None = None, stop_rate: int | None = None):
        """``stop_rate``: the hand-over rate -- a divisor of ``fs_in`` above 11 025 Hz whose ratio ``stage_factors`` can split
        (``handover_rate(fs_in)``, the lowest one, by default).  The exact FFT resampler takes the last step from there, i.e. the
        reference's own brick wall at 5512.5 Hz, and only the wide, flat filters of these stages separate the result from it.
        The lower the hand-over rate, the shorter the resampler's forward transform."""
        if att_db is None:
            att_db = float(os.environ.get("WFX_FE_ATT", DEFAULT_ATT_DB))
        self.att_db = att_db
        if int(fs_in) != fs_in or fs_in < 2 * MIN_HANDOVER:
            raise ValueError(f"the time-domain front end needs an integer rate >= {2 * MIN_HANDOVER} Hz, not {fs_in}; use the exact FFT resampler")
        self.fs_in = int(fs_in)
        self.out_rate = int(stop_rate) if stop_rate else self.handover_rate(self.fs_in)
        if self.out_rate <= TARGET_RATE + 1000 or self.fs_in % self.out_rate or stage_factors(self.fs_in // self.out_rate) is None:
            raise ValueError(f"stop_rate {self.out_rate}: not a rate above 11025 Hz that {self.fs_in} Hz reaches by decimations "
                             "(powers of two, then 2, 3, 5 or 6)")
        self.exact_tail = True               # an FFT resample from out_rate to 11 025 Hz follows (kept for callers of rounds 2-3)
        self.f64 = True                      # the chain runs integer-exact + float64 (FrontEndDevice)
        self.stages = []
        self.design = None                   # figures of the least-squares pair, when it replaced the Kaiser designs
        fs = Fraction(self.fs_in)
        for f in stage_factors(self.fs_in // self.out_rate):
            self.stages.append(Decimate(fs, f, NYQ, float(fs / f) - NYQ, att_db))
            fs = fs / f
        self._multiband_pair()
        self._finish()

    def _finish(self):
        """The first stage -- int16 samples, a power-of-two factor >= 8 -- runs as an integer-exact dot product with taps on the grid
        2**-30 (``quantize_taps``; csrc/wfx_polyphase.hip MODE 1), the stages behind it with float64 taps and sums.  What reaches
        the exact path then differs from an ideal filter by the designs' own error alone."""
        if self.design is None:
            # Kaiser designs at the low rates (48 kHz: /3; 192 kHz: /4, /3): their pass-band ripple 10**(-att/20) = 1.8e-7 would be the
            # largest error left (4-9 flipped stream bytes per 48 kHz clip at 135 dB, 1-3 at 160 dB, none at 200 dB);
            # 200 dB costs half as many taps again where taps are cheap
            for k, st in enumerate(self.stages):
                if float(st.fs_in) <= 200e3 and self.att_db < LOW_RATE_ATT_DB:
                    self.stages[k] = Decimate(st.fs_in, st.factor, NYQ, float(st.fs_out) - NYQ, LOW_RATE_ATT_DB)
        for st in self.stages:
            st.f64_chain = True
        s0 = self.stages[0]
        if s0.factor & (s0.factor - 1) == 0 and s0.factor >= 8 and not s0.fix_shift:
            sh = fix_shift_for(s0.coef64, s0.factor)            # (the least-squares pair quantised its first filter before designing the second)
            # a filter with large taps (small factors: /4 at 192 kHz) only gets a coarse grid, and nothing behind it compensates:
            # measured 11-12 flipped stream bytes per 192 kHz clip at 2**-24 -- such stages run in float64 instead (cheap there)
            if sh >= MIN_FIX_SHIFT:
                s0.set_taps(quantize_taps(s0.coef64, sh), sh)

    def _multiband_pair(self):
        """[ /M, /k ] with a large M (the ingest of an oversampled stream): the first filter only has to reject what folds into
        0..5512.5 Hz -- bands of +-5512.5 Hz around the multiples of its output rate -- and may do anything in between, which a
        windowed sinc cannot exploit.  A least-squares design over those bands reaches the stop-band depth in 8 M - 3 taps (8 taps
        per polyphase row: a third less work in the kernel's tap loop than the Kaiser design's 12) at the price of a pass band that
        ripples by 1e-5; the SECOND filter is designed against the inverse of that response, so that the pair is flat to better
        than the Kaiser chain was (measured 9e-8 for 1.536 MS/s, against 1.8e-7).  Kept only if the achieved figures meet att_db."""
        if len(self.stages) != 2 or self.stages[0].kind != "decimate" or self.stages[0].factor < 8:
            return
        s1, s2 = self.stages
        n1 = 8 * s1.factor - 3
        if s1.ntaps <= n1:
            return
        key = (self.fs_in, s1.factor, s2.factor, self.att_db)
        if key in _PAIR_CACHE:                       # (two least-squares solves: 2 s)
            h1, h2, self.design, sh1 = _PAIR_CACHE[key]
            if h1 is not None:
                s1.set_taps(h1, sh1)
                s2.set_taps(h2)
            return
        _PAIR_CACHE[key] = (None, None, None, 0)
        fs1, fo1 = float(s1.fs_in), float(s1.fs_out)
        stops = [(k * fo1 - NYQ, min(k * fo1 + NYQ, fs1 / 2)) for k in range(1, int(fs1 / 2 // fo1) + 1) if k * fo1 - NYQ < fs1 / 2]
        # (stop bands weighted 3000 : 1 -- the pass band may ripple by 1e-4, the second filter follows it anyway: -149 dB instead of
        # -133 dB before the taps are put on their grid; after: -138 dB on the 2**-27 grid of the kernel's first form, -149 dB on
        # 2**-30 now that it flushes the high halves of its sums too)
        h1 = ls_fir(n1, fs1, [(0.0, NYQ)] + stops, [1.0] + [0.0] * len(stops), [1.0] + [3000.0] * len(stops))
        h1 /= h1.sum()
        sh1 = fix_shift_for(h1, s1.factor)
        if sh1 < MIN_FIX_SHIFT:
            sh1 = 0
        if sh1:
            h1 = quantize_taps(h1, sh1)              # what the integer-exact kernel applies; the second filter is designed against THIS response
        fs2, fo2 = float(s2.fs_in), float(s2.fs_out)
        # more taps than the plain low-pass: the pass band now has a shape to follow.  95 taps (WFX_FE_N2_EXTRA=8) make the PAIR flat
        # to 2e-8, 119 taps to 3e-10.  While the first filter sat on the 2**-27 grid (-138 dB of stop band) the longer filter bought
        # nothing; on the 2**-30 grid (-149 dB, the design's own depth) it does -- 80 random 1.536 MS/s clips: 55 -> 62 of 65
        # decodable ones with the identical uint8 stream, 17 -> 5 differing bytes of 22.5 M -- for +1.3 % on the whole decode
        n2 = (s2.ntaps + int(os.environ.get("WFX_FE_N2_EXTRA", "32"))) | 1
        h2 = ls_fir(n2, fs2, [(0.0, NYQ), (fo2 - NYQ, fs2 / 2)], [lambda f: 1.0 / _response(h1, fs1, f), 0.0], [1.0, 1.0], grid=1200, iters=14)
        # verify before adopting
        fp = np.linspace(0.0, NYQ, 3000)
        tol = 10.0 ** (-self.att_db / 20.0)
        stop1 = max(np.abs(_response(h1, fs1, np.linspace(lo, hi, 1500))).max() for lo, hi in stops)
        stop2 = np.abs(_response(h2, fs2, np.linspace(fo2 - NYQ, fs2 / 2, 4000))).max()
        flat = np.max(np.abs(_response(h1, fs1, fp) * _response(h2, fs2, fp) - 1.0))
        if stop1 <= 2.0 * tol and stop2 <= tol and flat <= tol:
            s1.set_taps(h1, sh1)
            s2.set_taps(h2)
            self.design = {"stage1_stop_db": float(20 * np.log10(stop1)), "stage2_stop_db": float(20 * np.log10(stop2)), "pair_flatness": float(flat)}
            _PAIR_CACHE[key] = (h1, h2, self.design, sh1)

    @staticmethod
    def handover_rate(fs_in: int) -> int:
        """The lowest hand-over rate this front end can deliver for ``fs_in``: at least 14 000 Hz, reached by powers of two and a
        last factor of 2, 3, 5 or 6 (1.536 MS/s, 192 kHz, 48 kHz -> 16 000; 44.1 kHz -> 14 700; 96 kHz -> 16 000)."""
        fs_in = int(fs_in)
        for ratio in range(fs_in // MIN_HANDOVER, 0, -1):
            if fs_in % ratio == 0 and stage_factors(ratio) is not None:
                return fs_in // ratio
        raise ValueError(f"no hand-over rate for {fs_in} Hz")

    def n_target(self, n_in: int) -> int:
        """wefax.py:384 ``num = int(11025 * length)`` with ``length = n / sample_rate`` (wefax.py:357), in the reference's own
        float arithmetic: the number of 11 025 Hz samples the decode of an ``n_in``-frame capture must have."""
        return int(TARGET_RATE * (n_in / self.fs_in))

    def granule(self) -> int:
        """Capture lengths this front end reproduces the reference's sampling grid for: whole multiples of this many frames."""
        return Fraction(self.out_rate, self.fs_in).denominator

    def n_out(self, n_in: int) -> int:
        """Samples this front end delivers for an ``n_in``-frame capture: ``n_in * out_rate / fs_in``, which must be a whole
        number.  The reference's FFT resampler treats the capture as ONE period of ``n_in`` frames and puts output j at input
        position ``j * n_in / num`` (wefax.py:384, scipy.signal.resample).  A chain of fixed-ratio stencils keeps that period
        only when its outputs tile it exactly; the exact FFT resampler behind the hand-over then goes from ``n_out`` samples
        to ``n_target(n_in)`` whatever their ratio -- the reference's grid, also for captures that are not whole seconds.
        Any other length is refused (drop ``n_in % granule()`` trailing frames -- less than one hand-over sample -- or use
        the exact path, which takes every length that fits)."""
        prod = Fraction(n_in) * Fraction(self.out_rate, self.fs_in)
        if prod.denominator != 1:
            g = self.granule()
            raise ValueError(f"time-domain front end {self.fs_in} -> {self.out_rate} Hz: a capture of {n_in} frames is not a whole number of "
                             f"hand-over samples (granule {g} frames, {n_in % g} too many): the reference's resampling grid "
                             "(wefax.py:384) cannot be kept; drop the surplus frames or use the exact path")
        return int(prod)

    def chain(self, lo: int, hi: int):
        """Index ranges per stage for outputs [lo, hi) at the hand-over rate: list of (stage, out range,
        in range), first stage first; a stage's output range is the next stage's input range."""
        out = []
        a, b = lo, hi
        for st in reversed(self.stages):
            ia, ib = st.in_range(a, b)
            out.append((st, (a, b), (ia, ib)))
            a, b = ia, ib
        return out[::-1]

    def input_range(self, lo: int, hi: int):
        return self.chain(lo, hi)[0][2]

    def halo(self) -> int:
        """Input samples needed beyond the nominal position of an output, per side (upper bound)."""
        ia, ib = self.input_range(0, 1)
        return max(-ia, ib)

    def describe(self) -> str:
        parts = []
        for st in self.stages:
            parts.append(f"/{st.factor} ({st.ntaps} taps @ {float(st.fs_in):g} Hz{', integer-exact on the 2^-%d grid' % st.fix_shift if st.fix_shift else ', float64'})")
        return " -> ".join(parts)
