"""Sample-range sharding of ONE long 11 025 Hz capture across GPUs.

The exact path (``wefax.Demodulator``) is global per capture: FFT Hilbert, global
percentiles, a sequential sync search.  This module is the halo-local alternative the
north star describes: every rank owns a contiguous sample range, recomputes a halo on
both sides, and only three small things are exchanged --

  * 6 x one all-reduce of a 4 x 2048 histogram (exact global percentiles, radix select),
  * one broadcast of the sync-search result (start_frame; rank 0 owns the capture's head),
  * ONE gather of the finished image rows to the root.

Operators (all through the C ABI, ``wfx_d_*``): the notch filtfilt in its 49-tap FIR form
with filtfilt's exact edges where a slice touches the capture's true start / end, the
circular FIR Hilbert with the kernel of the WHOLE signal (``taps`` lags), 5-tap median
(zero-padded at the true ends like the reference), quantise, sync search, Pillow-exact
bicubic rows.  The result is bit-identical for any number of ranks AND to the single-GPU
``Demodulator(..., hilbert_mode=WFX_HILBERT_FIR)`` decode (tested); against the exact path
it carries the FIR truncation error of SURVEY.md appendix B.2 (<= 1 LSB on clean captures
at 4095 taps, more on noisy ones), which is why the exact single-GPU path is the default.

``torch.distributed`` is only the transport (``TorchComm``); ``LocalComm`` runs one rank.
"""
from __future__ import annotations

import math

import numpy as np

from . import hostparams as hp

SEL_BITS, SEL_BINS, SEL_LEVELS = 11, 2048, 6


def _sel_shift(level):
    return 53 - 11 * level if level < 5 else 0


def _sel_width(level):
    return 11 if level < 5 else 9


def key_to_f64(key: int) -> float:
    u = (key & 0x7FFFFFFFFFFFFFFF) if (key >> 63) else (~key & 0xFFFFFFFFFFFFFFFF)
    return float(np.array([u], dtype=np.uint64).view(np.float64)[0])


def np_lerp(a: float, b: float, t: float) -> float:
    """numpy/lib/_function_base_impl.py::_lerp on scalars."""
    diff = b - a
    r = a + diff * t
    if t >= 0.5:
        r = b - diff * (1 - t)
    return r


class ShardPlan:
    """Index ranges (global sample indices) of one rank.

    own      [o0, o1)  samples whose envelope this rank contributes to the percentiles
    compute  [c0, c1)  own +- (3 lines + 8): envelope / quantised stream needed for its image rows
    median   [m0, m1)  compute +- 4 (clipped to the capture: the median zero-pads at the TRUE ends only)
    load     [l0, l1)  median +- margin, NOT clipped: indices wrap modulo n (circular operators)
    """

    def __init__(self, n: int, world: int, rank: int, width: int, taps: int):
        self.n, self.world, self.rank = n, world, rank
        self.o0, self.o1 = rank * n // world, (rank + 1) * n // world
        halo = 3 * width + 8
        self.c0, self.c1 = max(0, self.o0 - halo), min(n, self.o1 + halo)
        self.m0, self.m1 = max(0, self.c0 - 4), min(n, self.c1 + 4)
        margin = (taps - 1) // 2 + 32 + 24 + 8
        margin = max(512, (margin + 15) // 16 * 16)        # >= the shortest segment the notch kernel accepts
        # l0 is a multiple of 16: the FIR kernel's fp32 accumulation order depends on the sample's
        # position inside a lane's 16-sample block, so slices are aligned to the global grid and
        # every sample is computed identically for any world size
        self.l0, self.l1 = (self.m0 - margin) // 16 * 16, self.m1 + margin
        if self.l1 - self.l0 > n + 2 * margin:
            raise ValueError("capture too short to shard")

    def rows(self, start: int, width: int, h_total: int):
        """Image lines [y0, y1) whose first sample lies in this rank's own range."""
        y0 = max(0, -((start - self.o0) // width))          # ceil((o0 - start) / w)
        y1 = max(0, -((start - self.o1) // width))
        return min(y0, h_total), min(y1, h_total)


class LocalComm:
    world, rank = 1, 0

    def allreduce_sum(self, a: np.ndarray) -> np.ndarray:
        return a

    def bcast(self, obj, root=0):
        return obj

    def gather(self, obj, root=0):
        return [obj]


class TorchComm:
    """torch.distributed as the transport (gloo on CPU, nccl = RCCL on GPUs)."""

    def __init__(self, dist, torch, device="cpu"):
        self.dist, self.torch, self.device = dist, torch, device
        self.world, self.rank = dist.get_world_size(), dist.get_rank()

    def allreduce_sum(self, a: np.ndarray) -> np.ndarray:
        t = self.torch.from_numpy(np.ascontiguousarray(a, dtype=np.int64)).to(self.device)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM)
        return t.cpu().numpy()

    def bcast(self, obj, root=0):
        box = [obj]
        self.dist.broadcast_object_list(box, src=root)
        return box[0]

    def gather(self, obj, root=0):
        out = [None] * self.world if self.rank == root else None
        self.dist.gather_object(obj, out, dst=root)
        return out


class HipStages:
    """The stage backend used in production: device-resident calls on one native context."""

    def __init__(self, ctx):
        self.ctx = ctx
        self.ptrs = []

    def _alloc(self, nbytes):
        p = self.ctx.dev_malloc(nbytes)
        self.ptrs.append(p)
        return p

    def close(self):
        for p in self.ptrs:
            self.ctx.dev_free(p)
        self.ptrs = []

    def load_slice(self, xe: np.ndarray):
        n = int(xe.shape[0])
        self.nl = n
        self.p_x = self._alloc(2 * n)
        self.p_af, self.p_er, self.p_em = self._alloc(8 * n), self._alloc(8 * n), self._alloc(8 * n)
        self.p_dq = self._alloc(n + 64)
        self.p_hist = self._alloc(4 * SEL_BINS * 4)
        self.ctx.dev_upload(self.p_x, np.ascontiguousarray(xe, dtype=np.int16))

    def load_raw(self, raw, in_kind: int, nl: int, front_end_only: bool = False):
        """Oversampled input of the time-domain front end: ``raw`` is a host array (int16 [n] or [n, 2]) or a
        (device pointer, frames) pair that stays owned by the caller; ``nl`` = samples of the 11 025 Hz slice.
        ``front_end_only``: no buffers for the halo-local stages behind it."""
        self.nl, self.in_kind = nl, in_kind
        if isinstance(raw, tuple):
            self.p_raw, self.n_raw = int(raw[0]), int(raw[1])
        else:
            raw = np.ascontiguousarray(raw, dtype=np.int16)
            self.n_raw = int(raw.shape[0])
            self.p_raw = self._alloc(raw.nbytes)
            self.ctx.dev_upload(self.p_raw, raw)
        self.p_x = self._alloc(8 * nl)                 # float64 audio at 11 025 Hz
        self.x_f64 = True
        self.p_stage = {}
        if front_end_only:
            return
        self.p_af, self.p_er, self.p_em = self._alloc(8 * nl), self._alloc(8 * nl), self._alloc(8 * nl)
        self.p_dq = self._alloc(nl + 64)
        self.p_hist = self._alloc(4 * SEL_BINS * 4)

    def front_end(self, chain):
        """Run the stage chain of ``polyphase.FrontEnd.chain``: raw slice -> float64 audio of the slice."""
        from . import _native as nat
        cur, kind, n_cur = self.p_raw, self.in_kind, self.n_raw
        for k, (st, (a, b), (ia, ib)) in enumerate(chain):
            assert ib - ia == n_cur, (ib - ia, n_cur)
            last = k == len(chain) - 1
            n_out = b - a
            if last:
                out = self.p_x
            else:
                if k not in self.p_stage:
                    self.p_stage[k] = self._alloc(4 * n_out)
                out = self.p_stage[k]
            if st.kind == "decimate":
                self.ctx.d_decimate_fir(cur, kind, n_cur, 0, st.factor, st.coef, out, last, n_out)
            else:
                shift = max(0, -(a // st.q))              # whole phase periods: makes the first output index non-negative
                self.ctx.d_resample_rational(cur, kind, n_cur, ia + st.left + shift * st.p, st.p, st.q, st.table,
                                             a + shift * st.q, out, n_out)
            cur, kind, n_cur = out, nat.WFX_IN_F32_MONO, n_out

    def notch_envelope(self, n_global, taps, b, a, med_lo, med_hi, segments):
        for lo, hi, flags in segments:        # pieces of the slice between the capture's true ends
            if getattr(self, "x_f64", False):
                self.ctx.d_notch_fir_f64(self.p_x + 8 * lo, hi - lo, b, a, self.p_af + 8 * lo, flags)
            else:
                self.ctx.d_notch_fir(self.p_x + 2 * lo, hi - lo, b, a, self.p_af + 8 * lo, flags)
        self.ctx.d_fir_envelope(self.p_af, self.nl, n_global, taps, self.p_er)
        self.ctx.d_median5(self.p_er + 8 * med_lo, med_hi - med_lo, self.p_em + 8 * med_lo)

    def level_hist(self, lo, hi, level, prefixes) -> np.ndarray:
        self.ctx.dev_upload(self.p_hist, np.zeros(4 * SEL_BINS, dtype=np.uint32))
        self.ctx.d_select_hist(self.p_em + 8 * lo, hi - lo, level, prefixes, self.p_hist)
        return self.ctx.dev_download(self.p_hist, (4, SEL_BINS), np.uint32).astype(np.int64)

    def quantise(self, lo, hi, low, high) -> int:
        return self.ctx.d_quantise(self.p_em + 8 * lo, hi - lo, low, high, self.p_dq + lo)

    def sync_search(self, lo, hi, n_total, n1, n0, mind, frame_samples, width):
        info = self.ctx.d_sync_search(self.p_dq + lo, hi - lo, n_total, n1, n0, mind, frame_samples, width)
        return {"start_frame": int(info.start_frame), "height": int(info.height), "no_group": int(info.no_group),
                "npeaks": int(info.npeaks), "hit_limit": int(info.hit_limit),
                "peaks": [int(info.peak_pos[k]) for k in range(info.npeaks)],
                "first": [int(info.first_pos[k]) for k in range(info.npeaks)],
                "phasing": [int(info.phasing[k]) for k in range(info.n_phasing)]}

    def image_rows(self, lo, hi, g0, start, width, h_total, y0, rows) -> np.ndarray:
        if rows <= 0:
            return np.zeros((0, width), dtype=np.uint8)
        p_img = self._alloc(4 * rows * width)
        self.ctx.d_image_rows(self.p_dq + lo, hi - lo, g0, start, width, h_total, y0, rows, p_img)
        return self.ctx.dev_download(p_img, (4 * rows, width), np.uint8)

    def image_rows_dev(self, lo, hi, g0, start, width, h_total, y0, rows):
        """Rows stay on the device: (device pointer, nbytes) for a collective's send buffer."""
        if rows <= 0:
            return 0, 0
        need = 4 * rows * width
        if getattr(self, "img_cap", 0) < need:
            self.p_img, self.img_cap = self._alloc(need), need
        self.ctx.d_image_rows(self.p_dq + lo, hi - lo, g0, start, width, h_total, y0, rows, self.p_img)
        return self.p_img, need

    def fetch(self, what, lo, hi):
        if what == "env":
            return self.ctx.dev_download(self.p_em + 8 * lo, (hi - lo,), np.float64)
        if what == "audio":         # output of the front end (before the notch)
            return self.ctx.dev_download(self.p_x + 8 * lo, (hi - lo,), np.float64)
        return self.ctx.dev_download(self.p_dq + lo, (hi - lo,), np.uint8)


class FrontEndExactDecoder:
    """ONE GPU, oversampled capture: the time-domain front end (polyphase.FrontEnd, halo-local stencils) followed by
    the EXACT rest of the path -- notch filtfilt, FFT Hilbert, global percentiles, sync search, bicubic image: the fused
    decode of ``wefax.DecodeJob`` attached to the front end's output in HBM.  Differs from the reference only by the
    front end's pass band (polyphase.py); faster than the halo-local form, which is what several GPUs need."""

    def __init__(self, ctx, frontend, x, n_in_total=None, in_kind=None, lines_per_minute: int = 120, raw_loader=None):
        from .wefax import DecodeJob
        n_in_total = int(n_in_total if n_in_total is not None else np.asarray(x).shape[0])
        n_fe = frontend.n_out(n_in_total)                   # at 11 025 Hz, or at 22 050 Hz (FrontEnd(stop_at_2x=True))
        rate = 2 * hp.TARGET_RATE if frontend.stop_at_2x else hp.TARGET_RATE
        self.n = n_fe // 2 if frontend.stop_at_2x else n_fe
        self.chain = frontend.chain(0, n_fe)
        ia, ib = self.chain[0][2]
        raw = raw_loader(ia, ib) if raw_loader is not None else np.asarray(x)[np.arange(ia, ib) % n_in_total]
        if in_kind is None:
            in_kind = 1 if (not isinstance(raw, tuple) and raw.ndim == 2) else 0
        self.st = HipStages(ctx)
        self.st.load_raw(raw, in_kind, n_fe, front_end_only=True)
        self.job = DecodeJob.from_device(ctx, self.st.p_x, n_fe, lines_per_minute, sample_rate=rate)
        assert self.job.n == self.n
        self.width = self.job.width

    def run(self):
        """Enqueue front end + fused decode (asynchronous)."""
        self.st.front_end(self.chain)
        self.job.run()

    def result(self):
        return self.job.result()

    def fetch(self, what: str):
        return self.job.fetch(what)

    def close(self):
        self.st.close()


class ShardedDecoder:
    """One rank's part of a sharded decode.  ``run(comm)`` is the multi-process driver;
    ``decode_emulated`` runs every rank of a world in this process (tests, 1-GPU boxes)."""

    def __init__(self, stages, x: np.ndarray, n_total: int, world: int, rank: int, lines_per_minute: int = 120,
                 taps: int = 4095, notch=hp.DEFAULT_NOTCH, slice_loader=None, frontend=None, n_in_total=None,
                 in_kind=None, raw_loader=None):
        """``x``: the capture at 11 025 Hz (int16), or -- with ``frontend`` (polyphase.FrontEnd) -- the oversampled
        capture of ``n_in_total`` frames (int16 [n] / [n, 2]); ``raw_loader(lo, hi)`` then replaces indexing ``x``
        and returns the frames [lo, hi) (indices beyond the capture wrap modulo ``n_in_total``) as a host array or
        a (device pointer, frames) pair.  ``n_total`` is always the sample count at 11 025 Hz."""
        self.st = stages
        self.n = int(n_total)
        self.frontend = frontend
        self.frame_len = 1 / (lines_per_minute / 60)
        self.width = int(self.frame_len * hp.TARGET_RATE)
        self.taps = taps
        self.plan = ShardPlan(self.n, world, rank, self.width, taps)
        self.b, self.a = hp.iirnotch(int(notch[0]), notch[1], hp.TARGET_RATE)
        p = self.plan
        if frontend is not None:
            self.chain = frontend.chain(p.l0, p.l1)
            ia, ib = self.chain[0][2]
            n_in_total = int(n_in_total if n_in_total is not None else np.asarray(x).shape[0])
            if ib - ia > 2 * n_in_total:
                raise ValueError("capture too short to shard")
            raw = raw_loader(ia, ib) if raw_loader is not None else np.asarray(x)[np.arange(ia, ib) % n_in_total]
            if in_kind is None:
                in_kind = 1 if (not isinstance(raw, tuple) and raw.ndim == 2) else 0
            self.st.load_raw(raw, in_kind, p.l1 - p.l0)
        else:
            idx = np.arange(p.l0, p.l1) % self.n
            xe = slice_loader(idx) if slice_loader is not None else np.asarray(x)[idx]
            self.st.load_slice(xe)
        self.ranks4 = None

    def _loc(self, g):                # global sample index -> index into the loaded slice
        return g - self.plan.l0

    # ---- phases -------------------------------------------------------------------
    def segments(self):
        """The loaded slice cut at the capture's true ends (it wraps circularly there):
        (lo, hi, flags) in slice indices, flags bit 0 / 1 = starts / ends at a true end."""
        p = self.plan
        nl = p.l1 - p.l0
        cuts = sorted({k * self.n - p.l0 for k in range(-1, 4) if 0 <= k * self.n - p.l0 <= nl})
        bounds = sorted(set([0, nl] + cuts))
        return [(lo, hi, (1 if lo in cuts else 0) | (2 if hi in cuts else 0)) for lo, hi in zip(bounds[:-1], bounds[1:])]

    def phase_envelope(self):
        p = self.plan
        if self.frontend is not None:
            self.st.front_end(self.chain)
        self.st.notch_envelope(self.n, self.taps, self.b, self.a, self._loc(p.m0), self._loc(p.m1), self.segments())

    def phase_hist(self, level, prefixes):
        p = self.plan
        return self.st.level_hist(self._loc(p.o0), self._loc(p.o1), level, prefixes)

    @staticmethod
    def pick_digits(hist: np.ndarray, ranks, prefixes, level):
        """Host side of the radix select: from the summed histograms, the digit holding each rank."""
        width = _sel_width(level)
        new_p, new_r = [], []
        for q in range(4):
            cum = np.cumsum(hist[q][: 1 << width])
            d = int(np.searchsorted(cum, ranks[q], side="right"))
            before = int(cum[d - 1]) if d > 0 else 0
            new_p.append(((prefixes[q] << width) | d) if level > 0 else d)
            new_r.append(int(ranks[q]) - before)
        return new_p, new_r

    def percentiles(self, reduce_fn):
        lo0, lo1, glo = hp.percentile_plan(self.n, 0.5)
        hi0, hi1, ghi = hp.percentile_plan(self.n, 99.5)
        ranks, prefixes = [lo0, lo1, hi0, hi1], [0, 0, 0, 0]
        for level in range(SEL_LEVELS):
            hist = reduce_fn(self.phase_hist(level, prefixes))
            prefixes, ranks = self.pick_digits(hist, ranks, prefixes, level)
        v = [key_to_f64(k) for k in prefixes]
        return np_lerp(v[0], v[1], glo), np_lerp(v[2], v[3], ghi)

    def phase_quantise(self, low, high):
        p = self.plan
        return self.st.quantise(self._loc(p.c0), self._loc(p.c1), low, high)

    def phase_sync(self):
        """Rank 0 only: its compute range starts at sample 0 of the capture."""
        p = self.plan
        assert p.c0 == 0
        n1, n0, mind = hp.sync_constants(hp.TARGET_RATE, self.frame_len)
        r = self.st.sync_search(self._loc(0), self._loc(p.c1), self.n, n1, n0, mind,
                                self.frame_len * hp.TARGET_RATE, self.width)
        if not r["hit_limit"] and p.c1 < self.n:
            raise RuntimeError("sync search ran past rank 0's shard (fewer than 100 peaks in it): "
                               "capture too short for this world size, decode it on one GPU")
        return r

    def phase_image(self, start, h_total):
        p = self.plan
        y0, y1 = p.rows(start, self.width, h_total)
        rows = self.st.image_rows(self._loc(p.c0), self._loc(p.c1), p.c0, start, self.width, h_total, y0, y1 - y0)
        return y0, rows

    # ---- drivers --------------------------------------------------------------------
    def run(self, comm, exchange=None, keep_on_device=False):
        """All ranks call this; the root gets (image, sync dict, low, high), the others None.
        With an ``ImageExchange`` the rows are gathered device to device (the root then gets a
        list of (first line, uint8 device tensor) instead of one host array).  ``keep_on_device``
        (single rank): the image stays in HBM and (device pointer, bytes) is returned in its place."""
        self.phase_envelope()
        low, high = self.percentiles(comm.allreduce_sum)
        nan = int(comm.allreduce_sum(np.array([self.phase_quantise(low, high)], dtype=np.int64))[0])
        if nan:
            raise ValueError("cannot convert float NaN to integer")
        sync = self.phase_sync() if comm.rank == 0 else None
        sync = comm.bcast(sync, 0)
        if sync["no_group"]:
            max([], key=len)
        if exchange is not None and hasattr(self.st, "image_rows_dev"):
            # rows never leave HBM: device-to-device into the collective's send buffer, one RCCL gather
            p = self.plan
            y0, y1 = p.rows(sync["start_frame"], self.width, sync["height"])
            ptr, nb = self.st.image_rows_dev(self._loc(p.c0), self._loc(p.c1), p.c0, sync["start_frame"], self.width,
                                             sync["height"], y0, y1 - y0)
            if nb:
                self.st.ctx.dev_copy(exchange.payload_ptr, ptr, nb)
            got = exchange.gather(nb, y0)            # the header's second field carries the first line
            if comm.rank != 0:
                return None
            return sorted(((y, buf) for buf, y in got if buf.numel()), key=lambda t: t[0]), sync, low, high
        if keep_on_device and comm.world == 1 and hasattr(self.st, "image_rows_dev"):
            p = self.plan
            y0, y1 = p.rows(sync["start_frame"], self.width, sync["height"])
            return self.st.image_rows_dev(self._loc(p.c0), self._loc(p.c1), p.c0, sync["start_frame"], self.width,
                                          sync["height"], y0, y1 - y0), sync, low, high
        y0, rows = self.phase_image(sync["start_frame"], sync["height"])
        parts = comm.gather((y0, rows), 0)           # the one image collective
        if comm.rank != 0:
            return None
        img = np.concatenate([r for _, r in sorted(parts, key=lambda t: t[0])], axis=0)
        return img, sync, low, high


def decode_emulated(make_stages, x: np.ndarray, world: int, lines_per_minute: int = 120, taps: int = 4095, frontend=None):
    """Run every rank of a ``world``-rank sharded decode in this process, phase by phase."""
    n = int(np.asarray(x).shape[0])
    if frontend is not None:
        n = frontend.n_out(n)
    decs = [ShardedDecoder(make_stages(), x, n, world, r, lines_per_minute, taps, frontend=frontend) for r in range(world)]
    for d in decs:
        d.phase_envelope()
    lo0, lo1, glo = hp.percentile_plan(n, 0.5)
    hi0, hi1, ghi = hp.percentile_plan(n, 99.5)
    ranks, prefixes = [lo0, lo1, hi0, hi1], [0, 0, 0, 0]
    for level in range(SEL_LEVELS):
        hist = sum(d.phase_hist(level, prefixes) for d in decs)
        prefixes, ranks = ShardedDecoder.pick_digits(hist, ranks, prefixes, level)
    v = [key_to_f64(k) for k in prefixes]
    low, high = np_lerp(v[0], v[1], glo), np_lerp(v[2], v[3], ghi)
    if sum(d.phase_quantise(low, high) for d in decs):
        raise ValueError("cannot convert float NaN to integer")
    sync = decs[0].phase_sync()
    if sync["no_group"]:
        max([], key=len)
    parts = [d.phase_image(sync["start_frame"], sync["height"]) for d in decs]
    img = np.concatenate([r for _, r in sorted(parts, key=lambda t: t[0])], axis=0)
    env = np.concatenate([d.st.fetch("env", d._loc(d.plan.o0), d._loc(d.plan.o1)) for d in decs])
    aud = (np.concatenate([d.st.fetch("audio", d._loc(d.plan.o0), d._loc(d.plan.o1)) for d in decs])
           if frontend is not None else None)
    dig = np.concatenate([d.st.fetch("dig", d._loc(d.plan.o0), d._loc(d.plan.o1)) for d in decs])
    for d in decs:
        if hasattr(d.st, "close"):
            d.st.close()
    return {"image": img, "sync": sync, "low": low, "high": high, "envelope": env, "digitalized": dig, "audio": aud}
