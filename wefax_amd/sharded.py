"""ONE capture decoded by several GPUs (include/wefax_hip.h "one capture over several GPUs").

The reference decodes on one host and two of its operators are global: ``scipy.signal.hilbert``
(/root/reference/wefax.py:174) and ``scipy.signal.resample`` (wefax.py:384) are DFTs over the whole capture.
Round 1 replaced them with halo-local FIRs on the sharded path and paid with +-1 LSB differences that move sync
peaks.  Here they stay exact: the native library computes them as DISTRIBUTED transforms (two transposes per
transform over RCCL, ``csrc/wfx_dist.hip``); every other stage runs on a rank's own sample range with a halo.
The result is the single-GPU exact path's and does not depend on the number of ranks.

This module is the thin host side: it cuts the capture the way ``wfx_shard_layout_query`` says, hands each rank its
frames, and drives the phases.  No PyTorch anywhere: the communicator is RCCL bound by the library itself; the
128-byte unique id travels over a loopback TCP socket (``bootstrap_unique_id``).

    * ``ShardedDecoder``      one rank of a real multi-process decode (RCCL), or world 1
    * ``decode_emulated``     every rank of a world in THIS process on one GPU (tests, one-GPU boxes)
    * ``FrontEndDevice`` / ``FrontEndExactDecoder`` / ``FrontEndShardedDecoder``
                              oversampled captures (BASELINE configs[3]): the time-domain front end of
                              ``polyphase.py`` down to the hand-over rate (16 000 Hz) on each rank's slice, then the exact (sharded) path
                              whose FFT resampler takes the last factor of two -- the reference's own brick wall
"""
from __future__ import annotations

import os
import socket
import struct
import time
from fractions import Fraction

import numpy as np

from . import _native as nat
from . import hostparams as hp
from .wefax import build_params

_MAGIC = b"WFXUID01"


# ---- RCCL unique id over a loopback socket ---------------------------------------------------------------------
PORT_SPAN = 16        # rank 0 listens on the first free port of [port, port + PORT_SPAN); the others probe the range


def _nonce16(nonce) -> bytes:
    import hashlib
    return hashlib.sha256(str(nonce).encode()).digest()[:16]


def bootstrap_unique_id(rank: int, world: int, addr: str = "127.0.0.1", port: int = 29611, timeout: float = 120.0,
                        make_id=nat.comm_unique_id, nonce="") -> bytes:
    """Rank 0 creates the RCCL unique id and serves it to the other ``world - 1`` ranks; they fetch it.
    Plain TCP on ``addr`` (stdlib only).  Rank 0 takes the first port of [port, port + 16) it can bind; the peers probe
    that range until something answers with the protocol's magic AND the job's ``nonce`` (any string the ranks of one job
    share: two decodes on one host do not serve each other's peers), so a port that another program -- or another job --
    holds is skipped.  A rank is counted once, whatever it claims to be."""
    tag = _nonce16(nonce)
    if world == 1:
        return make_id()
    if rank == 0:
        uid = make_id()
        srv = None
        for k in range(PORT_SPAN):
            cand = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
            cand.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
            try:
                cand.bind((addr, port + k))
                srv = cand
                break
            except OSError:
                cand.close()
        if srv is None:
            raise nat.NativeError(f"rank 0: no free port in [{port}, {port + PORT_SPAN}) on {addr}")
        srv.listen(world)
        srv.settimeout(timeout)
        served = set()
        try:
            while len(served) < world - 1:
                conn, _ = srv.accept()
                with conn:
                    conn.settimeout(10.0)
                    try:
                        hello = _recv_exact(conn, len(_MAGIC) + 4 + 16)
                    except OSError:
                        continue
                    peer = struct.unpack("<i", hello[len(_MAGIC):len(_MAGIC) + 4])[0]
                    if hello[:len(_MAGIC)] != _MAGIC or hello[len(_MAGIC) + 4:] != tag or not 0 < peer < world:
                        continue                      # not one of ours
                    conn.sendall(_MAGIC + tag + uid)
                    served.add(peer)
        except socket.timeout:
            raise nat.NativeError(f"rank 0: only {len(served)} of {world - 1} ranks fetched the RCCL unique id within {timeout} s")
        finally:
            srv.close()
        return uid
    deadline = time.time() + timeout
    last = None
    while time.time() < deadline:
        for k in range(PORT_SPAN):
            try:
                with socket.create_connection((addr, port + k), timeout=2.0) as conn:
                    conn.settimeout(10.0)
                    conn.sendall(_MAGIC + struct.pack("<i", rank) + tag)
                    blob = _recv_exact(conn, len(_MAGIC) + 16 + nat.WFX_COMM_ID_BYTES)
                    if blob[:len(_MAGIC)] == _MAGIC and blob[len(_MAGIC):len(_MAGIC) + 16] == tag:
                        return blob[len(_MAGIC) + 16:]
            except OSError as e:
                last = e
        time.sleep(0.05)
    raise nat.NativeError(f"rank {rank}: could not fetch the RCCL unique id from {addr}:{port}+ ({last})")


def _recv_exact(conn, n: int) -> bytes:
    buf = b""
    while len(buf) < n:
        part = conn.recv(n - len(buf))
        if not part:
            raise OSError("connection closed")
        buf += part
    return buf


# ---- plain captures ------------------------------------------------------------------------------------------------
def capture_kind(data: np.ndarray) -> int:
    if data.ndim == 2:
        if data.dtype != np.int16:
            raise ValueError("sharded decode: two-channel captures must be int16")
        return nat.WFX_IN_I16_STEREO
    return nat.WFX_IN_I16_MONO if data.dtype == np.int16 else nat.WFX_IN_F64_MONO


class ShardedDecoder:
    """One rank of a sharded exact decode of a capture held on the host (or loaded slice-wise).

    ``loader(lo, hi)`` returns the frames [lo, hi) of the capture; with ``data`` given it is an index expression.
    Ranks other than 0 get ``None`` images; scalars (low, high) are available everywhere."""

    def __init__(self, ctx: nat.Context, comm: nat.Comm, n0: int, sample_rate, lines_per_minute: int = 120, kind: int = nat.WFX_IN_I16_MONO,
                 notch=None, data: np.ndarray | None = None, loader=None, n_out: int | None = None, plan="auto"):
        self.ctx, self.comm = ctx, comm
        self.frame_len = 1 / (lines_per_minute / 60)
        if notch is None:                       # config/config.json like Demodulator (wefax.py:63-66), defaults when there is no file
            notch = hp.load_notch_settings()
        self.params, self.meta = build_params(kind, int(n0), sample_rate, self.frame_len, notch, n_out=n_out, shard_plan=plan_code(plan))
        # uint8 / int32 / float32 captures reach the device as float64, but filtfilt's odd extension (wefax.py:72) is what scipy
        # evaluates in the file's OWN dtype (wraps / float32 rounding): 9 + 9 numbers from the capture's two ends, as DecodeJob
        # hands them over (the ranks holding a true end use them; wfx_shard.hip phase 4)
        if kind == nat.WFX_IN_F64_MONO and not self.meta["resampled"] and int(n0) > 9:
            head = tail = None
            if data is not None and np.asarray(data).dtype not in (np.float64, np.int16):
                head, tail = np.asarray(data)[:10], np.asarray(data)[-10:]
            elif data is None and loader is not None:
                h0 = np.asarray(loader(0, 10))
                if h0.dtype not in (np.float64, np.int16):
                    head, tail = h0, np.asarray(loader(int(n0) - 10, int(n0)))
            if head is not None:
                left, _ = hp.odd_extension(head)
                _, right = hp.odd_extension(tail)
                self.params.has_ext = 1
                self.params.ext_left[:] = [float(v) for v in left]
                self.params.ext_right[:] = [float(v) for v in right]
        self.n = self.meta["n"]
        self.width = self.params.width
        self.shard = nat.Shard(ctx, comm, self.params)
        lay = self.shard.layout
        self.layout = lay
        if data is not None:
            frames = gather_frames(lay, np.asarray(data))
        elif loader is not None:
            frames = gather_frames(lay, None, loader, int(n0))
        else:
            frames = None
        if frames is not None:
            if kind != nat.WFX_IN_I16_MONO and kind != nat.WFX_IN_I16_STEREO:
                frames = np.ascontiguousarray(frames, dtype=np.float64)
            self.shard.upload(frames)

    def attach(self, dev_ptr: int):
        self.shard.attach(dev_ptr)

    def run(self):
        self.shard.run()

    def result(self) -> nat.DecodeInfo:
        return self.shard.result()

    def fetch(self, which: str) -> np.ndarray:
        lay = self.layout
        own = lay.own_samples
        if which == "audio":
            return self.shard.fetch(nat.WFX_BUF_AUDIO, (own,), np.float64)
        if which == "envelope":
            return self.shard.fetch(nat.WFX_BUF_ENVELOPE, (own,), np.float64)
        if which == "digitalized":
            return self.shard.fetch(nat.WFX_BUF_DIGITAL, (own,), np.uint8)
        if which == "stream":           # rank 0: the whole uint8 stream after the gather
            return self.shard.fetch(nat.WFX_BUF_DIGITAL, (self.n,), np.uint8)
        if which == "image":
            info = self.result()
            return self.shard.fetch(nat.WFX_BUF_IMAGE, (4 * info.height, info.width), np.uint8)
        raise KeyError(which)

    def close(self):
        self.shard.close()


PLAN_CODES = {"auto": 0, "dist": 1, "single": 2, "fmm": 3, "rows": 17, "auto-rows": 16}


def plan_code(plan) -> int:
    """``wfx_decode_params.shard_plan``: "auto" (the library's cost model picks distributed or single), "dist" (distributed
    whenever a distributed form exists), "single" (rank 0 alone), "rows" (distributed in the rows layout of rounds 2-3: A/B runs),
    "auto-rows"; "fmm" (round 6: captures at 11 025 Hz cut into contiguous ranges, the Hilbert transform by the fast multipole form of
    csrc/wfx_fmm.hip -- kilobytes on the wire instead of four transposes of the capture); or the integer itself."""
    return int(PLAN_CODES[plan]) if isinstance(plan, str) else int(plan)


def gather_frames(lay, data=None, loader=None, n0: int | None = None) -> np.ndarray:
    """The frames a rank hands over for its layout: one range (rows layout, single plan, one rank), or -- columns layout --
    ``nseg`` segments with ``in_halo`` frames on either side, back to back; frames outside the capture are zeros."""
    if lay.plan == 3:
        # the multipole plan: the own range and `in_halo` frames on either side ROUND THE CIRCLE (rank 0's left halo is the capture's end: the
        # Hilbert transform is cyclic; the kernels apply filtfilt's exact edges there)
        n0 = int(np.asarray(data).shape[0] if data is not None else n0)
        lo, hi, h = int(lay.in_lo), int(lay.in_hi), int(lay.in_halo)
        take = (lambda a, b: np.asarray(data)[a:b]) if data is not None else loader
        parts = []
        if lo - h < 0:
            parts += [take(n0 + lo - h, n0), take(0, lo)] if lo > 0 else [take(n0 - h, n0)]
        else:
            parts.append(take(lo - h, lo))
        parts.append(take(lo, hi))
        if hi + h > n0:
            parts += [take(hi, n0), take(0, hi + h - n0)] if hi < n0 else [take(0, h)]
        else:
            parts.append(take(hi, hi + h))
        return np.concatenate([np.asarray(q) for q in parts])
    if lay.nseg <= 1:
        lo, hi = int(lay.in_lo), int(lay.in_hi)
        return np.asarray(data)[lo:hi] if data is not None else loader(lo, hi)
    h, ln, st, lo = int(lay.in_halo), int(lay.in_seg_len), int(lay.in_seg_stride), int(lay.in_lo)
    n0 = int(data.shape[0] if data is not None else n0)
    parts = []
    for sgm in range(int(lay.nseg)):
        a, b = lo + sgm * st - h, lo + sgm * st + ln + h
        ca, cb = max(a, 0), min(b, n0)                          # (a padded form's last segments lie beyond the capture: all zeros)
        if cb > ca:
            src = np.asarray(data[ca:cb] if data is not None else loader(ca, cb))
            if ca == a and cb == b:
                parts.append(src)
                continue
            part = np.zeros((b - a,) + src.shape[1:], dtype=src.dtype)
            part[ca - a:cb - a] = src
        else:
            ref = np.asarray(data[:1] if data is not None else loader(0, 1))
            part = np.zeros((b - a,) + ref.shape[1:], dtype=ref.dtype)
        parts.append(part)
    return np.concatenate(parts)


def assemble(layouts, blocks, n: int, dtype=None) -> np.ndarray:
    """Per-rank stage buffers (``ShardedDecoder.fetch``) -> the whole capture's array in sample order."""
    out = np.zeros(int(n), dtype=dtype if dtype is not None else np.asarray(blocks[0]).dtype)
    for lay, blk in zip(layouts, blocks):
        if lay.own_samples:
            idx = lay.own_index()
            if idx.size and idx[-1] >= out.shape[0]:        # a padded form's segments reach past the capture's end: those slots hold nothing
                keep = idx < out.shape[0]
                out[idx[keep]] = np.asarray(blk)[keep]
            else:
                out[idx] = blk
    return out


def layout_supported(n0: int, sample_rate, world: int, lines_per_minute: int = 120, kind: int = nat.WFX_IN_F64_MONO, n_out: int | None = None) -> bool:
    """Whether the sharded path takes a capture of this description over ``world`` ranks (``wfx_shard_layout_query`` answers
    without a GPU).  Since round 3 that is every valid capture: one with no distributed form gets the single plan."""
    p, _ = build_params(kind, int(n0), sample_rate, 1 / (lines_per_minute / 60), hp.DEFAULT_NOTCH, n_out=n_out)
    try:
        nat.shard_layout(p, world, 0)
        return True
    except nat.NativeError:
        return False


def layout_distributed(n0: int, sample_rate, world: int, lines_per_minute: int = 120, kind: int = nat.WFX_IN_F64_MONO, n_out: int | None = None) -> bool:
    """Whether the capture's transforms are distributed over the ranks (False: the single plan -- rank 0 decodes it alone)."""
    p, _ = build_params(kind, int(n0), sample_rate, 1 / (lines_per_minute / 60), hp.DEFAULT_NOTCH, n_out=n_out, shard_plan=plan_code("dist"))
    lay = nat.shard_layout(p, world, 0)
    return not (lay.first_radix[0] == 0 and lay.first_radix[1] == 0)


def _info_dict(info: nat.DecodeInfo) -> dict:
    return {"start_frame": int(info.start_frame), "height": int(info.height), "no_group": int(info.no_group),
            "npeaks": int(info.npeaks), "hit_limit": int(info.hit_limit), "nan_count": int(info.nan_count),
            "peaks": [int(info.peak_pos[k]) for k in range(info.npeaks)],
            "first": [int(info.first_pos[k]) for k in range(info.npeaks)],
            "phasing": [int(info.phasing[k]) for k in range(info.n_phasing)]}


def decode_emulated(data: np.ndarray, sample_rate, world: int, lines_per_minute: int = 120, device: int = 0, notch=None,
                    want=("image", "stream", "envelope", "audio"), make_decoder=None, free_after=None, repeat: int = 1, plan="dist"):
    """Every rank of a ``world``-rank sharded decode in this process, on one GPU, phase by phase (local communicator:
    a collective completes when the last rank has posted its part).  Returns the root's results plus the per-rank
    blocks put in sample order, for comparison with the single-GPU path (tests).  ``plan``: "dist" by default -- the emulation
    exists to exercise the distributed form at every world size, whatever the cost model would pick on real links."""
    data = np.asarray(data)
    comms = nat.Comm.local(world)
    ctxs = [nat.Context(device) for _ in range(world)]
    decs = []
    try:
        for r in range(world):
            if make_decoder is not None:
                decs.append(make_decoder(ctxs[r], comms[r]))
            else:
                decs.append(ShardedDecoder(ctxs[r], comms[r], data.shape[0], sample_rate, lines_per_minute, capture_kind(data), notch, data=data, plan=plan))
        for d in decs:
            if hasattr(d, "front_end"):
                d.front_end()
        for rep in range(max(1, repeat) - 1):       # earlier decodes of the same shards (buffers and plans are reused by the last one)
            for ph in range(decs[0].shard.phases):
                for d in decs:
                    d.shard.phase(ph)
            for d in decs:
                d.result()
        for attempt in range(8):
            nph = decs[0].shard.phases
            for ph in range(nph):
                for d in decs:
                    d.shard.phase(ph)
            try:
                infos = [d.result() for d in decs]
                break
            except nat.NativeError as e:
                # the percentile select's candidate lists overflowed (every rank reports it alike and has raised its capacity):
                # the ranks of an in-process world are driven from here, so the repeat is ours
                if "overflow" not in str(e) or attempt == 7:
                    raise
                for d in decs[1:]:
                    try:
                        d.result()
                    except nat.NativeError:
                        pass
        out = {"sync": _info_dict(infos[0]), "low": infos[0].low, "high": infos[0].high,
               "lows": [i.low for i in infos], "highs": [i.high for i in infos], "n": decs[0].n, "width": decs[0].width,
               "layouts": [(int(d.layout.own_lo), int(d.layout.own_hi), int(d.layout.in_lo), int(d.layout.in_hi)) for d in decs],
               "own": [d.layout.own_samples for d in decs], "plan": int(decs[0].layout.plan),
               "first_radix": tuple(decs[0].layout.first_radix), "wire": [c.wire_stats() for c in comms]}
        lays = [d.layout for d in decs]
        if "stream" in want:
            out["digitalized"] = decs[0].fetch("stream")
            out["digitalized_blocks"] = assemble(lays, [d.fetch("digitalized") for d in decs], decs[0].n, np.uint8)
        if "envelope" in want:
            out["envelope"] = assemble(lays, [d.fetch("envelope") for d in decs], decs[0].n, np.float64)
        if "audio" in want:
            out["audio"] = assemble(lays, [d.fetch("audio") for d in decs], decs[0].n, np.float64)
        if "image" in want and not infos[0].no_group and not infos[0].nan_count and infos[0].height > 0:
            out["image"] = decs[0].fetch("image")
        return out
    finally:
        for d in decs:
            d.close()
        for c, ptr in (free_after or []):          # device memory the decoders' loaders allocated on the ranks' contexts
            c.dev_free(ptr)
        for c in comms:
            c.close()
        for c in ctxs:
            c.close()


# ---- oversampled captures: the time-domain front end on the device -------------------------------------------
class FrontEndDevice:
    """The stage chain of ``polyphase.FrontEnd`` on one context: raw frames (int16 mono / IQ) of a slice -> float64 audio.

    ``raw``: host array (int16 [n] or [n, 2]) or a (device pointer, frames) pair that stays owned by the caller."""

    def __init__(self, ctx: nat.Context, chain, raw, in_kind: int, nbatch: int = 1, raw_stride: int = 0, out_ptr: int | None = None):
        """``nbatch`` > 1 (the segments a rank owns in the columns layout of the sharded decode): ``raw`` holds that many equally
        long slices, ``raw_stride`` frames apart (a multiple of 16 bytes); every stage runs them in ONE launch and the result is
        ``nbatch`` rows of ``n_out`` samples back to back.  Float64 chains only."""
        self.ctx, self.chain, self.in_kind = ctx, chain, in_kind
        self.nbatch = int(nbatch)
        self.ptrs = []
        if isinstance(raw, tuple):
            self.p_raw, self.n_raw = int(raw[0]), int(raw[1])
        else:
            raw = np.ascontiguousarray(raw, dtype=np.int16)
            self.n_raw = int(raw.shape[0])
            self.p_raw, self.placement = ctx.dev_malloc_placed(raw.nbytes)      # (WFX_PLACE_TRIES > 1: the best of a few allocations)
            self.ptrs.append(self.p_raw)
            ctx.dev_upload(self.p_raw, raw)
        self.raw_stride = int(raw_stride) if self.nbatch > 1 else 0
        a, b = chain[-1][1]
        self.n_out = b - a
        self.p_out = int(out_ptr) if out_ptr else None          # (given: the caller's buffer, e.g. a slot of a larger one -- not freed here)
        probe_out = ((self.n_raw // 32 - 8) - 119) // 3 + 1            # outputs wfx_d_stream_rate writes for a capture of n_raw frames
        if (self.p_out is None and self.nbatch == 1 and self.n_raw * 4 >= (1 << 30) and in_kind == nat.WFX_IN_I16_STEREO and len(chain) == 2 and chain[0][0].factor == 32
                and chain[1][0].factor == 3 and probe_out <= self.n_out):
            # the 1.536 MS/s ingest writes its output in small bursts under 22 GB of reads: where THAT buffer lies counts as much as where
            # the capture does (WFX_PLACE_TRIES > 1: the best of a few allocations, timed with the capture in place)
            self.p_out, self.placement_out = ctx.dev_malloc_placed(8 * self.n_out, probe=lambda q: ctx.d_stream_rate(self.p_raw, self.n_raw * 4, out_ptr=q))
            self.ptrs.append(self.p_out)
        if self.p_out is None:
            self.p_out = self._alloc(8 * self.n_out * self.nbatch)
        self.p_stage = {}
        # a chain of decimations: integer-exact ingest where the first stage qualifies, float64 behind it (polyphase.FrontEnd._finish)
        self.f64 = True
        if any(st.kind != "decimate" for st, _, _ in chain):
            raise ValueError("front end: chains of decimations only (the fp32 rational stage was removed in round 4)")
        self.stage_stride = {}
        for k, (st, (a, b), _) in enumerate(chain[:-1]):
            stride = (b - a) + ((b - a) & 1)                    # members of a batch start a multiple of 16 bytes apart
            self.stage_stride[k] = stride           # (buffers on first use: the fused ingest never needs stage 0's)
        self.exact_ingest = None
        self.fused_ingest = False           # run() took the one-kernel form of the first two stages (csrc/wfx_ingest.hip)

    def _alloc(self, nbytes):
        p = self.ctx.dev_malloc(nbytes)
        self.ptrs.append(p)
        return p

    def _stage(self, k):
        if k not in self.p_stage:
            self.p_stage[k] = self._alloc(8 * self.stage_stride[k] * self.nbatch)
        return self.p_stage[k]

    def run(self):
        """Enqueue the chain (asynchronous); the float64 audio of the slice ends at ``p_out``."""
        cur, kind, n_cur = self.p_raw, self.in_kind, self.n_raw
        first_stage = 0
        if len(self.chain) >= 2 and self.chain[0][0].fix_shift and not os.environ.get("WFX_FE_UNFUSED"):
            # the ingest and the stage behind it in ONE streaming kernel (csrc/wfx_ingest.hip): the intermediate rate never reaches
            # memory; bit-identical to the two launches below, which run when the shapes are not that kernel's
            (s1, _, (ia, ib)), (s2, (a, b), _) = self.chain[0], self.chain[1]
            assert ib - ia == n_cur, (ib - ia, n_cur)
            last = len(self.chain) == 2
            out = self.p_out if last else self._stage(1)
            if self.ctx.d_ingest_chain(cur, kind, n_cur, s1.factor, s1.coef64, s1.fix_shift, s2.factor, s2.coef64, out, b - a, nbatch=self.nbatch,
                                       in_stride=self.raw_stride, out_stride=(b - a) if last else self.stage_stride[1]):
                self.exact_ingest = True
                self.fused_ingest = True
                cur, kind, n_cur, first_stage = out, nat.WFX_IN_F64_MONO, b - a, 2
        for k, (st, (a, b), (ia, ib)) in enumerate(self.chain):
            if k < first_stage:
                continue
            assert ib - ia == n_cur, (ib - ia, n_cur)
            last = k == len(self.chain) - 1
            n_out = b - a
            out = self.p_out if last else self._stage(k)
            in_stride = self.raw_stride if k == 0 else self.stage_stride[k - 1]
            ex = self.ctx.d_decimate_fir64(cur, kind, n_cur, 0, st.factor, st.coef64, out, n_out, st.fix_shift if k == 0 else 0,
                                           nbatch=self.nbatch, in_stride=in_stride, out_stride=n_out if last else self.stage_stride[k])
            if k == 0:
                self.exact_ingest = ex
            cur, kind, n_cur = out, nat.WFX_IN_F64_MONO, n_out

    def fetch(self) -> np.ndarray:
        return self.ctx.dev_download(self.p_out, (self.n_out * self.nbatch,), np.float64)

    def close(self):
        for p in self.ptrs:
            self.ctx.dev_free(p)
        self.ptrs = []


def _raw_slice(x, raw_loader, ia: int, ib: int, n_in_total: int):
    if raw_loader is not None:
        return raw_loader(ia, ib)
    return np.asarray(x)[np.arange(ia, ib) % n_in_total]


class FrontEndExactDecoder:
    """ONE GPU, oversampled capture: the time-domain front end (polyphase.FrontEnd(stop_rate=...), halo-local stencils)
    down to the hand-over rate, then the fused exact decode of ``wefax.DecodeJob`` attached to its output in HBM -- notch
    filtfilt, FFT resample by 2, FFT Hilbert, global percentiles, sync search, bicubic image.  Differs from the reference
    only by the front end's pass band (polyphase.py)."""

    def __init__(self, ctx, frontend, x, n_in_total=None, in_kind=None, lines_per_minute: int = 120, raw_loader=None, notch=None,
                 hilbert_mode=None):
        from .wefax import DecodeJob, DEFAULT_HILBERT_MODE
        n_in_total = int(n_in_total if n_in_total is not None else np.asarray(x).shape[0])
        n_fe = frontend.n_out(n_in_total)                   # at 11 025 Hz, or at the hand-over rate (FrontEnd(stop_rate=...))
        rate = frontend.out_rate
        self.n = frontend.n_target(n_in_total)              # wefax.py:384 on the ORIGINAL capture
        self.chain = frontend.chain(0, n_fe)
        ia, ib = self.chain[0][2]
        raw = _raw_slice(x, raw_loader, ia, ib, n_in_total)
        if in_kind is None:
            in_kind = 1 if (not isinstance(raw, tuple) and raw.ndim == 2) else 0
        self.fe = FrontEndDevice(ctx, self.chain, raw, in_kind)
        self.job = DecodeJob.from_device(ctx, self.fe.p_out, n_fe, lines_per_minute, notch=notch if notch is not None else hp.load_notch_settings(),
                                         hilbert_mode=DEFAULT_HILBERT_MODE if hilbert_mode is None else hilbert_mode, sample_rate=rate,
                                         n_out=self.n if frontend.exact_tail else None)
        assert self.job.n == self.n
        self.width = self.job.width

    def run(self):
        """Enqueue front end + fused decode (asynchronous)."""
        self.fe.run()
        self.job.run()

    def result(self):
        return self.job.result()

    def fetch(self, what: str):
        return self.job.fetch(what)

    def close(self):
        self.fe.close()


INGEST_BYTES_PER_S = 6.0e12          # the streaming ingest reads a capture at 0.75-0.8 of the 8 TB/s HBM peak


def choose_plan_with_front_end(n_fe: int, rate, lines_per_minute: int, notch, n_out: int, world: int, raw_bytes: int):
    """Which plan an oversampled capture takes on `world` ranks when the caller leaves it open: the library's model of every candidate (its own
    choice, the chunk-local plan, the transposing plan) for the hand-over-rate signal PLUS the front end's time -- the whole stream on rank 0
    under the single plan, 1 / world of it under a sharded one.  Host only; every rank computes the same answer.  Returns (plan name, figures)."""
    frame_len = 1 / (lines_per_minute / 60)
    if notch is None:
        notch = hp.load_notch_settings()
    fe_s = raw_bytes / INGEST_BYTES_PER_S
    figures = {}
    for name in ("auto", "fmm", "dist"):
        try:
            p, _ = build_params(nat.WFX_IN_F64_MONO, int(n_fe), rate, frame_len, notch, n_out=n_out, shard_plan=plan_code(name))
            lay = nat.shard_layout(p, world, 0)
        except nat.NativeError:
            continue
        if lay.plan == 0:
            figures[name] = {"plan": 0, "model_s": lay.model_single_s + fe_s}
        else:
            figures[name] = {"plan": int(lay.plan), "model_s": lay.model_dist_compute_s + lay.model_dist_wire_s + fe_s / world}
    best = min(figures, key=lambda k: (figures[k]["model_s"], k != "auto"))
    return best, {"front_end_s": fe_s, "candidates": figures, "chosen": best}


class FrontEndShardedDecoder:
    """One rank of the sharded decode of an oversampled capture (BASELINE configs[3]): the rank runs the front end over the raw
    frames its rows of the hand-over-rate signal need (halo of the FIR chain included: ``raw_loader(lo, hi)`` with indices
    wrapping modulo the capture), then the sharded exact path takes over -- its distributed FFT resampler brings the
    hand-over-rate signal to 11 025 Hz exactly as the one-GPU form does.  The raw stream is split ``world`` ways and never moves
    between GPUs; what is exchanged is the transposes of the transforms at the hand-over rate and at 11 025 Hz."""

    def __init__(self, ctx, comm, frontend, x, n_in_total=None, in_kind=None, lines_per_minute: int = 120, raw_loader=None,
                 notch=None, plan="auto"):
        n_in_total = int(n_in_total if n_in_total is not None else np.asarray(x).shape[0])
        n_fe = frontend.n_out(n_in_total)
        self.plan_choice = None
        if plan == "auto" and comm.world > 1:
            # the library's cost model sees the capture at the hand-over rate only; the front end in front of it is most of an oversampled
            # decode, and only a sharded plan divides it by the world size -- the choice is made here, with the front end's time added
            plan, self.plan_choice = choose_plan_with_front_end(n_fe, frontend.out_rate, lines_per_minute, notch, frontend.n_target(n_in_total), comm.world,
                                                                n_in_total * (4 if (in_kind == nat.WFX_IN_I16_STEREO or (x is not None and np.asarray(x).ndim == 2)) else 2))
        self.dec = ShardedDecoder(ctx, comm, n_fe, frontend.out_rate, lines_per_minute, nat.WFX_IN_F64_MONO, notch,
                                  n_out=frontend.n_target(n_in_total), plan=plan)
        lay = self.dec.layout
        self.shard, self.layout, self.n, self.width = self.dec.shard, lay, self.dec.n, self.dec.width
        if lay.in_hi == lay.in_lo:
            # a rank that owns nothing (the SINGLE plan of a capture with no distributed form: rank 0 decodes alone) has no front
            # end to run; it takes part in the phases with a placeholder input
            self.chain, self.fe, self.raw_range, self.raw_frames = [], None, (0, 0), 0
            self._none = ctx.dev_malloc(64)
            self.dec.attach(self._none)
            return
        if lay.nseg > 1:
            # COLUMNS layout: the rank owns nseg equally long, equally spaced segments of the hand-over-rate signal -- the front end
            # runs over the raw frames of every one of them (its own halo around each: a few thousand frames, 0.3 % of a segment
            # of the 60-minute stream) in one batched launch per stage and delivers the rows the resampler's first pass reads
            assert lay.in_halo == 0
            ratio = Fraction(frontend.fs_in, frontend.out_rate)
            assert ratio.denominator == 1, "a chain of decimations only"
            self.chain = frontend.chain(int(lay.in_lo), int(lay.in_lo + lay.in_seg_len))
            ia, ib = self.chain[0][2]
            seg_raw = int(lay.in_seg_stride) * int(ratio)                       # raw frames between the starts of two segments
            if in_kind is None:
                in_kind = 1 if (x is not None and np.asarray(x).ndim == 2) else 0
            fb = 4 if in_kind == nat.WFX_IN_I16_STEREO else 2
            per16 = 16 // fb
            rstride = -(-(ib - ia) // per16) * per16
            nseg = int(lay.nseg)
            p_raw = ctx.dev_malloc(nseg * rstride * fb + 64)
            self._raw_ptr = p_raw
            if raw_loader is not None and hasattr(raw_loader, "into"):
                for sgm in range(nseg):                                         # (a loader that writes where it is told: no copy)
                    raw_loader.into(p_raw + sgm * rstride * fb, ia + sgm * seg_raw, ib + sgm * seg_raw)
            elif raw_loader is not None:
                for sgm in range(nseg):
                    got = raw_loader(ia + sgm * seg_raw, ib + sgm * seg_raw)
                    if isinstance(got, tuple):
                        ctx.dev_copy(p_raw + sgm * rstride * fb, int(got[0]), (ib - ia) * fb)
                    else:
                        ctx.dev_upload(p_raw + sgm * rstride * fb, np.ascontiguousarray(got, dtype=np.int16))
            else:
                xa = np.asarray(x)
                idx = (ia + np.arange(nseg, dtype=np.int64)[:, None] * seg_raw + np.arange(rstride, dtype=np.int64)[None, :]) % n_in_total
                ctx.dev_upload(p_raw, np.ascontiguousarray(xa[idx.reshape(-1)], dtype=np.int16))
            self.fe = FrontEndDevice(ctx, self.chain, (p_raw, ib - ia), in_kind, nbatch=nseg, raw_stride=rstride)
            self.dec.attach(self.fe.p_out)
            self.raw_range = (ia, ia + (nseg - 1) * seg_raw + (ib - ia))
            self.raw_frames = nseg * (ib - ia)
            return
        self.extra = []
        if lay.plan == 3:
            # the multipole plan: the rank's range of the hand-over-rate signal with `in_halo` samples on either side ROUND THE CIRCLE -- the
            # resampler is cyclic in that signal, so what lies beyond its ends is its other end (computed from the raw frames THERE, by a
            # second small chain), not what the FIR chain would make of raw frames wrapped round the capture
            lo, hi, h = int(lay.in_lo), int(lay.in_hi), int(lay.in_halo)
            m_lo, m_hi = max(lo - h, 0), min(hi + h, n_fe)
            self._comb = ctx.dev_malloc(8 * (hi - lo + 2 * h) + 64)
            pieces = [(m_lo, m_hi, m_lo - (lo - h))]
            if lo - h < 0:
                pieces.append((n_fe + lo - h, n_fe, 0))
            if hi + h > n_fe:
                pieces.append((0, hi + h - n_fe, n_fe - (lo - h)))
            fes = []
            self.raw_frames = 0
            for a, b, off in pieces:
                ch = frontend.chain(a, b)
                ia, ib = ch[0][2]
                raw = _raw_slice(x, raw_loader, ia, ib, n_in_total)
                if in_kind is None:
                    in_kind = 1 if (not isinstance(raw, tuple) and raw.ndim == 2) else 0
                fes.append(FrontEndDevice(ctx, ch, raw, in_kind, out_ptr=self._comb + 8 * off))
                self.raw_frames += ib - ia
            self.chain = fes[0].chain
            self.fe, self.extra = fes[0], fes[1:]
            self.raw_range = tuple(self.chain[0][2])
            self.dec.attach(self._comb)
            return
        self.chain = frontend.chain(int(lay.in_lo), int(lay.in_hi))
        ia, ib = self.chain[0][2]
        raw = _raw_slice(x, raw_loader, ia, ib, n_in_total)
        if in_kind is None:
            in_kind = 1 if (not isinstance(raw, tuple) and raw.ndim == 2) else 0
        self.fe = FrontEndDevice(ctx, self.chain, raw, in_kind)
        self.dec.attach(self.fe.p_out)
        self.raw_range = (ia, ib)
        self.raw_frames = ib - ia

    def run(self):
        self.front_end()
        self.dec.run()

    def front_end(self):
        if self.fe is not None:
            self.fe.run()
        for fe in getattr(self, "extra", []):
            fe.run()

    def result(self):
        return self.dec.result()

    def fetch(self, which: str):
        return self.dec.fetch(which)

    def close(self):
        self.dec.close()
        if getattr(self, "_raw_ptr", None):
            self.dec.ctx.dev_free(self._raw_ptr)
            self._raw_ptr = None
        for fe in getattr(self, "extra", []):
            fe.close()
        if getattr(self, "_comb", None):
            self.dec.ctx.dev_free(self._comb)
            self._comb = None
        if self.fe is not None:
            self.fe.close()
        elif getattr(self, "_none", None):
            self.dec.ctx.dev_free(self._none)
            self._none = None
