"""CPU stand-in for wefax_amd.sharded.HipStages, built from the oracle: it lets the
sharding ORCHESTRATION (ranges, halos, histogram merge, row ownership, gather) be tested
without a GPU.  Test infrastructure only."""
import numpy as np

from oracle import wefax_oracle as wo


def _keys(v):
    u = np.ascontiguousarray(v, dtype=np.float64).view(np.uint64)
    neg = (u >> np.uint64(63)).astype(bool)
    return np.where(neg, ~u, u | np.uint64(1 << 63))


def to_real(raw):
    """int16 [n] -> float64; int16 [n, 2] -> the reference's stereo merge (wefax.py:360-373): int16 wrap, /2."""
    raw = np.asarray(raw)
    if raw.ndim == 2:
        return (raw[:, 0].astype(np.int16) + raw[:, 1].astype(np.int16)).astype(np.int16).astype(np.float64) / 2
    return raw.astype(np.float64)


def decimate_model(x, first, factor, coef, n_out):
    """float64 model of wfx_d_decimate_fir: y[i] = sum_j c[j] x[first + i*factor + j], zeros outside x."""
    x = np.asarray(x, dtype=np.float64)
    c = np.asarray(coef, dtype=np.float64)
    lo, hi = first, first + (n_out - 1) * factor + c.shape[0]
    xp = np.zeros(hi - lo)
    a, b = max(lo, 0), min(hi, x.shape[0])
    if b > a:
        xp[a - lo:b - lo] = x[a:b]
    y = np.zeros(n_out)
    for j in range(c.shape[0]):
        y += c[j] * xp[j:j + (n_out - 1) * factor + 1:factor]
    return y


def rational_model(x, base0, p, q, table, m0, n_out):
    """float64 model of wfx_d_resample_rational."""
    x = np.asarray(x, dtype=np.float64)
    t = np.asarray(table, dtype=np.float64)
    m = m0 + np.arange(n_out, dtype=np.int64)
    pos = (m * p) // q - base0
    ph = (m * p) % q
    y = np.zeros(n_out)
    for j in range(t.shape[1]):
        s = pos + j
        ok = (s >= 0) & (s < x.shape[0])
        y += t[ph, j] * np.where(ok, x[np.clip(s, 0, x.shape[0] - 1)], 0.0)
    return y


def front_end_model(raw, chain):
    """The stage chain of polyphase.FrontEnd.chain on a raw slice, in float64."""
    cur = to_real(raw)
    for st, (a, b), (ia, ib) in chain:
        assert cur.shape[0] == ib - ia
        if st.kind == "decimate":
            cur = decimate_model(cur, 0, st.factor, st.coef, b - a)
        else:
            shift = max(0, -(a // st.q))
            cur = rational_model(cur, ia + st.left + shift * st.p, st.p, st.q, st.table, a + shift * st.q, b - a)
    return cur


class NumpyStages:
    def load_raw(self, raw, in_kind, nl):
        self.raw = np.asarray(raw)
        self.nl = nl
        self.em = np.zeros(nl)
        self.dq = np.zeros(nl, dtype=np.uint8)

    def front_end(self, chain):
        self.xe = front_end_model(self.raw, chain)       # float64 audio of the slice
        assert self.xe.shape[0] == self.nl

    def load_slice(self, xe):
        self.xe = np.asarray(xe)             # int16: filtfilt's odd extension wraps like it does in scipy
        self.nl = self.xe.shape[0]
        self.em = np.zeros(self.nl)
        self.dq = np.zeros(self.nl, dtype=np.uint8)

    def notch_envelope(self, n_global, taps, b, a, med_lo, med_hi, segments):
        # filtfilt per segment: exact where the segment touches a true end, and within ~30 samples
        # of the other ends (halo, never used) it differs from the FIR form -- irrelevant
        af = np.concatenate([wo.filtfilt_biquad(b, a, self.xe[lo:hi]) for lo, hi, _ in segments])
        half = (taps - 1) // 2
        m = np.arange(1, half + 1, 2, dtype=np.float64)
        if n_global % 2 == 0:
            t = (2.0 / n_global) / np.tan(np.pi * m / n_global)
        else:
            t = (1.0 / n_global) / np.tan(np.pi * m / (2.0 * n_global))
        k = np.zeros(2 * half + 1)
        k[half + m.astype(int)] = t          # lag +m
        k[half - m.astype(int)] = -t         # lag -m
        H = np.convolve(af, k, mode="same")
        er = np.hypot(af, H)
        self.em[med_lo:med_hi] = wo.medfilt5(er[med_lo:med_hi])

    def level_hist(self, lo, hi, level, prefixes):
        keys = _keys(self.em[lo:hi])
        shift = 53 - 11 * level if level < 5 else 0
        width = 11 if level < 5 else 9
        out = np.zeros((4, 2048), dtype=np.int64)
        for q in range(4):
            sel = keys if level == 0 else keys[(keys >> np.uint64(shift + width)) == np.uint64(prefixes[q])]
            d = ((sel >> np.uint64(shift)) & np.uint64((1 << width) - 1)).astype(np.int64)
            out[q, : 1 << width] = np.bincount(d, minlength=1 << width)
        return out

    def quantise(self, lo, hi, low, high):
        with np.errstate(invalid="ignore", divide="ignore"):
            d = np.round(255 * (self.em[lo:hi] - low) / (high - low))
        nan = int(np.isnan(d).sum())
        d = np.nan_to_num(np.clip(d, 0, 255))
        self.dq[lo:hi] = d.astype(np.uint8)
        return nan

    def sync_search(self, lo, hi, n_total, n1, n0, mind, frame_samples, width):
        d = self.dq[lo:hi]
        peaks, first, hit = wo.pick_peaks(wo.sync_correlation(d, n1, n0), mind)
        r = {"peaks": peaks, "first": first, "hit_limit": int(hit), "npeaks": len(peaks)}
        try:
            ph = wo.group_peaks(peaks, 11025, frame_samples / 11025)
            r.update(no_group=0, phasing=list(ph), start_frame=(ph[-1] if ph else 0))
        except ValueError:
            r.update(no_group=1, phasing=[], start_frame=0)
        r["height"] = 0 if r["no_group"] else (n_total - r["start_frame"]) // width
        return r

    def image_rows(self, lo, hi, g0, start, width, h_total, y0, rows):
        if rows <= 0:
            return np.zeros((0, width), dtype=np.uint8)
        kk, bounds = wo.pillow_vertical_coeffs(h_total, 4 * h_total)
        out = np.empty((4 * rows, width), dtype=np.uint8)
        d = self.dq[lo:hi]
        for yy in range(4 * y0, 4 * (y0 + rows)):
            ymin, cnt = int(bounds[yy, 0]), int(bounds[yy, 1])
            acc = np.full(width, 1 << 21, dtype=np.int64)
            for k in range(cnt):
                s = start + (ymin + k) * width - g0
                acc += (255 - d[s:s + width].astype(np.int64)) * int(kk[yy, k])
            out[yy - 4 * y0] = np.clip(acc >> 22, 0, 255)
        return out

    def fetch(self, what, lo, hi):
        if what == "audio":
            return np.asarray(self.xe, dtype=np.float64)[lo:hi].copy()
        return (self.em if what == "env" else self.dq)[lo:hi].copy()
