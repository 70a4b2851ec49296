"""Device-side synthesis of the oversampled IQ workload (BASELINE configs[3]).

A 60-minute 1.536 MS/s int16 IQ capture is 5.53 G frames = 22 GB: it cannot be a RIFF wav and is too slow to build on
the host, so the benchmark synthesises it straight into HBM with the library's own test-signal kernels
(``csrc/wfx_synth.hip``, C ABI ``wfx_synth_capture``): same recipe as ``synth.synth_capture``
(``wefax_frequency_track`` -> phase-continuous FM -> additive white noise -> int16), for any range of global frame
indices; indices outside the capture wrap modulo its length (the halo of the FIR front end at the capture's ends), and
the noise of a frame depends only on (seed, frame index) so overlapping slices of different ranks agree bit for bit.
Nothing here is on the decode path.
"""
from __future__ import annotations

from . import _native as nat


def synth_params(fs: float, noise: float = 0.0, seed: int = 0, amplitude: float = 0.5, lpm: int = 120, ioc: int = 576,
                 start_tone_s: float = 5.0, phasing_lines: int = 60, image_lines: int = 1200, stop_tone_s: float = 5.0,
                 black_tail_s: float = 10.0, iq: bool = True) -> nat.SynthParams:
    p = nat.SynthParams()
    p.sample_rate, p.lines_per_minute, p.ioc = float(fs), int(lpm), int(ioc)
    p.start_tone_s, p.phasing_lines, p.image_lines = float(start_tone_s), int(phasing_lines), int(image_lines)
    p.stop_tone_s, p.black_tail_s = float(stop_tone_s), float(black_tail_s)
    p.amplitude, p.noise, p.seed, p.iq = float(amplitude), float(noise), int(seed), int(bool(iq))
    return p


def capture_frames(fs: float, lpm: int = 120, start_tone_s: float = 5.0, phasing_lines: int = 60,
                   image_lines: int = 1200, stop_tone_s: float = 5.0, black_tail_s: float = 10.0) -> int:
    dur = start_tone_s + (phasing_lines + image_lines) * (60.0 / lpm) + stop_tone_s + black_tail_s
    return int(round(dur * fs))


def synth_into(ctx: nat.Context, params: nat.SynthParams, ptr: int, lo: int, hi: int):
    """Frames lo..hi-1 of the capture (global indices wrapping modulo its length) into device memory at ``ptr``."""
    n0 = int(ctx.lib.wfx_synth_frames(params))
    fb = 4 if params.iq else 2
    g = lo
    while g < hi:
        s = g % n0
        c = min(hi - g, n0 - s)
        ctx.synth_capture(params, s, s + c, ptr + (g - lo) * fb)
        g += c


PLACEMENTS = []        # stream rates (GB/s) of the candidates of every placed allocation (Context.dev_malloc_placed), for bench.py's report


def synth_slice(ctx: nat.Context, params: nat.SynthParams, lo: int, hi: int) -> int:
    """Device pointer (owned by the caller: ``ctx.dev_free``) to frames lo..hi-1 of the capture, global indices wrapping
    modulo its length.  int16 [hi - lo] or, with ``params.iq``, interleaved [hi - lo, 2]."""
    fb = 4 if params.iq else 2
    ptr, rates = ctx.dev_malloc_placed((hi - lo) * fb)
    if rates:
        PLACEMENTS.append(rates)
    synth_into(ctx, params, ptr, lo, hi)
    return ptr


class SliceLoader:
    """``raw_loader`` for the front-end decoders of ``sharded.py``: slices of a synthesised capture, each in device memory of
    its own (``loader(lo, hi)``; freed by ``close``) or written where the decoder wants them (``loader.into(ptr, lo, hi)``: the
    segments of the columns layout go straight into the decoder's batch buffer)."""

    def __init__(self, ctx: nat.Context, params: nat.SynthParams):
        self.ctx, self.params, self.keep, self.calls = ctx, params, [], []

    def __call__(self, lo: int, hi: int):
        ptr = synth_slice(self.ctx, self.params, lo, hi)
        self.keep.append(ptr)
        self.calls.append((lo, hi))
        return ptr, hi - lo

    def into(self, ptr: int, lo: int, hi: int):
        synth_into(self.ctx, self.params, ptr, lo, hi)
        self.calls.append((lo, hi))

    def close(self):
        for p in self.keep:
            self.ctx.dev_free(p)
        self.keep = []
