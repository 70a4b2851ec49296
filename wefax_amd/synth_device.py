"""Device-side synthesis of the oversampled IQ workload (BASELINE configs[3]).

A 60-minute 1.536 MS/s int16 IQ capture is 5.53 G frames = 22 GB: it cannot be a RIFF wav and
is too slow to build on the host, so the benchmark synthesises it straight into HBM with a
few torch kernels (torch is plumbing here: device memory and elementwise math for the TEST
SIGNAL; nothing in this file is on the decode path).  Same recipe as ``synth.synth_capture``
(``wefax_frequency_track`` -> phase-continuous FM -> additive white noise -> int16), evaluated
chunk by chunk with a running phase, for any range of global frame indices; indices outside
the capture wrap modulo its length (the halo of a circular operator), and the noise of a frame
depends only on (seed, chunk) so overlapping slices of different ranks agree bit for bit.
"""
from __future__ import annotations

import math

from .synth import BLACK_HZ, WHITE_HZ

CHUNK = 1 << 25          # frames per synthesis step


def capture_frames(fs: float, lpm: int = 120, start_tone_s: float = 5.0, phasing_lines: int = 60,
                   image_lines: int = 1200, stop_tone_s: float = 5.0, black_tail_s: float = 10.0) -> int:
    dur = start_tone_s + (phasing_lines + image_lines) * (60.0 / lpm) + stop_tone_s + black_tail_s
    return int(round(dur * fs))


def _freq(torch, t, lpm, ioc, start_tone_s, phasing_lines, image_lines, stop_tone_s):
    """synth.wefax_frequency_track on a tensor of times (float64 seconds)."""
    t_line = 60.0 / lpm
    start_hz = 300.0 if ioc == 576 else 675.0
    f = torch.full_like(t, BLACK_HZ)
    white = torch.full_like(t, WHITE_HZ)
    t0 = 0.0
    seg = (t >= t0) & (t < t0 + start_tone_s)
    ph = torch.remainder((t - t0) * start_hz, 1.0)
    f = torch.where(seg & (ph < 0.5), white, f)
    t0 += start_tone_s
    t1 = t0 + phasing_lines * t_line
    seg = (t >= t0) & (t < t1)
    frac = torch.remainder((t - t0) / t_line, 1.0)
    f = torch.where(seg & (frac < 0.05), white, f)
    t0 = t1
    t1 = t0 + image_lines * t_line
    seg = (t >= t0) & (t < t1)
    frac = torch.remainder((t - t0) / t_line, 1.0)
    ramp = BLACK_HZ + (WHITE_HZ - BLACK_HZ) * (frac - 0.05) / 0.95
    f = torch.where(seg, torch.where(frac < 0.05, white, ramp), f)
    t0 = t1
    seg = (t >= t0) & (t < t0 + stop_tone_s)
    ph = torch.remainder((t - t0) * 450.0, 1.0)
    f = torch.where(seg & (ph < 0.5), white, f)
    return f


def synth_iq_slice(torch, device, lo: int, hi: int, fs: float, noise: float = 0.0, seed: int = 0, amplitude: float = 0.5,
                   lpm: int = 120, ioc: int = 576, start_tone_s: float = 5.0, phasing_lines: int = 60,
                   image_lines: int = 1200, stop_tone_s: float = 5.0, black_tail_s: float = 10.0):
    """int16 tensor [hi - lo, 2] on ``device``: frames lo..hi-1 (global indices, wrapping modulo the capture)."""
    n0 = capture_frames(fs, lpm, start_tone_s, phasing_lines, image_lines, stop_tone_s, black_tail_s)
    out = torch.empty((hi - lo, 2), dtype=torch.int16, device=device)
    # pieces of [lo, hi) per period of the capture: (dst offset, source start, count)
    pieces = []
    g = lo
    while g < hi:
        s = g % n0
        c = min(hi - g, n0 - s)
        pieces.append((g - lo, s, c))
        g += c
    carry = 0.0
    two_pi_fs = 2.0 * math.pi / fs
    last_needed = max(s + c for _, s, c in pieces)
    for c0 in range(0, n0, CHUNK):
        c1 = min(c0 + CHUNK, n0)
        if c0 >= last_needed:
            break
        idx = torch.arange(c0, c1, dtype=torch.float64, device=device)
        f = _freq(torch, idx / fs, lpm, ioc, start_tone_s, phasing_lines, image_lines, stop_tone_s)
        cs = torch.cumsum(f, 0)
        total = float(cs[-1].item())
        need = [(d, s, c) for d, s, c in pieces if s < c1 and s + c > c0]
        if need:
            phi = (cs + carry) * two_pi_fs
            xi, xq = amplitude * torch.cos(phi), amplitude * torch.sin(phi)
            if noise > 0:
                gen = torch.Generator(device=device)
                gen.manual_seed(seed * 1000003 + c0 // CHUNK)
                nz = torch.randn((2, c1 - c0), dtype=torch.float32, device=device, generator=gen)
                xi = xi + noise * nz[0].double()
                xq = xq + noise * nz[1].double()
            iq = torch.stack([xi, xq], dim=1)
            iq = torch.clamp(torch.round(iq * 32767.0), -32768, 32767).to(torch.int16)
            for d, s, c in need:
                a, b = max(s, c0), min(s + c, c1)
                out[d + (a - s):d + (b - s)] = iq[a - c0:b - c0]
            del phi, xi, xq, iq
        carry += total
        del idx, f, cs
    return out
