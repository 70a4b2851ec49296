"""Synthetic WEFAX (HF radiofax) captures for tests and benchmarks.

The reference ships no generator and its only full-length recording is a
missing blob (/root/reference/.MISSING_LARGE_BLOBS:1), so every workload in
BASELINE.json's ``configs`` is built here from the transmission format the
reference documents in /root/reference/README.md:27-97 (black 1500 Hz, white
2300 Hz, start tone 300 Hz / 675 Hz toggle, phasing lines 5 % white + 95 %
black, stop tone 450 Hz toggle, then black).

Everything is plain numpy on the host; nothing here is on the decode path.
"""
from __future__ import annotations

import struct

import numpy as np

BLACK_HZ = 1500.0
WHITE_HZ = 2300.0


def wefax_frequency_track(fs: float,
                          lpm: int = 120,
                          ioc: int = 576,
                          start_tone_s: float = 5.0,
                          phasing_lines: int = 60,
                          image_lines: int = 1200,
                          stop_tone_s: float = 5.0,
                          black_tail_s: float = 10.0,
                          lead_silence_s: float = 0.0) -> np.ndarray:
    """Instantaneous frequency (Hz) per sample of one WEFAX transmission.

    Line timing sits on the exact time grid (60/lpm seconds per line, i.e.
    5512.5 samples at 11 025 Hz / 120 LPM), so the image slants exactly the way
    the reference's truncated ``frame_width`` makes it (wefax.py:298).
    """
    t_line = 60.0 / lpm
    start_hz = 300.0 if ioc == 576 else 675.0
    dur = (lead_silence_s + start_tone_s + (phasing_lines + image_lines) * t_line
           + stop_tone_s + black_tail_s)
    n = int(round(dur * fs))
    t = np.arange(n, dtype=np.float64) / fs
    f = np.full(n, BLACK_HZ, dtype=np.float64)

    t0 = lead_silence_s
    # start tone: square toggle white/black
    seg = (t >= t0) & (t < t0 + start_tone_s)
    ph = np.mod((t[seg] - t0) * start_hz, 1.0)
    f[seg] = np.where(ph < 0.5, WHITE_HZ, BLACK_HZ)
    t0 += start_tone_s

    # phasing lines: 5 % white then 95 % black
    t1 = t0 + phasing_lines * t_line
    seg = (t >= t0) & (t < t1)
    frac = np.mod((t[seg] - t0) / t_line, 1.0)
    f[seg] = np.where(frac < 0.05, WHITE_HZ, BLACK_HZ)
    t0 = t1

    # image lines: 5 % white sync + linear black->white ramp
    t1 = t0 + image_lines * t_line
    seg = (t >= t0) & (t < t1)
    frac = np.mod((t[seg] - t0) / t_line, 1.0)
    ramp = BLACK_HZ + (WHITE_HZ - BLACK_HZ) * (frac - 0.05) / 0.95
    f[seg] = np.where(frac < 0.05, WHITE_HZ, ramp)
    t0 = t1

    # stop tone: 450 Hz toggle
    seg = (t >= t0) & (t < t0 + stop_tone_s)
    ph = np.mod((t[seg] - t0) * 450.0, 1.0)
    f[seg] = np.where(ph < 0.5, WHITE_HZ, BLACK_HZ)
    # rest: black (already)
    if lead_silence_s > 0:
        f[t < lead_silence_s] = 0.0
    return f


def _phase(fs: float, f: np.ndarray) -> np.ndarray:
    return 2.0 * np.pi * np.cumsum(f) / fs


def _to_int16(x: np.ndarray) -> np.ndarray:
    return np.clip(np.rint(x * 32767.0), -32768, 32767).astype(np.int16)


def synth_capture(fs: float = 11025.0, noise: float = 0.0, seed: int = 0,
                  amplitude: float = 0.5, iq: bool = False, **kw) -> np.ndarray:
    """int16 capture: mono ``[n]``, or interleaved I/Q ``[n, 2]`` when ``iq``.

    ``noise`` is the standard deviation of additive white Gaussian noise in
    units of full scale.
    """
    ugly = kw.pop("ugly", None)
    f = wefax_frequency_track(fs, **kw)
    rng = np.random.default_rng(seed)
    n = f.shape[0]
    env = 1.0
    dc = 0.0
    if ugly:
        # a short-wave channel instead of a laboratory one (round 6): ``ugly`` = dict(fade_depth 0..1, fade_hz, drift_hz, impulses_per_s,
        # impulse_fs, clip, dc): slow selective fading of the amplitude, a carrier that drifts by up to drift_hz, impulsive noise, clipping at
        # full scale (clip = gain in front of the converter), a constant offset
        t = np.arange(n) / fs
        env = 1.0 - float(ugly.get("fade_depth", 0.0)) * 0.5 * (1.0 + np.sin(2 * np.pi * float(ugly.get("fade_hz", 0.2)) * t + rng.uniform(0, 2 * np.pi)))
        f = np.where(f > 0, f + float(ugly.get("drift_hz", 0.0)) * np.sin(2 * np.pi * t / max(t[-1], 1.0) * float(ugly.get("drift_cycles", 1.5))), f)
        dc = float(ugly.get("dc", 0.0))
    phi = _phase(fs, f)

    def dirt(x):
        if not ugly:
            return x
        rate = float(ugly.get("impulses_per_s", 0.0))
        if rate > 0:
            k = rng.poisson(rate * n / fs)
            pos = rng.integers(0, n, k)
            width = max(1, int(fs * 2e-4))
            amp = float(ugly.get("impulse_fs", 0.8)) * rng.choice([-1.0, 1.0], k)
            for p_, a_ in zip(pos, amp):
                x[p_:p_ + width] += a_
        return np.clip((x + dc) * float(ugly.get("clip", 1.0)), -1.0, 32767.0 / 32768.0)

    if not iq:
        x = amplitude * env * np.sin(phi)
        if kw.get("lead_silence_s", 0.0) > 0:
            x[f == 0.0] = 0.0
        if noise > 0:
            x = x + noise * rng.standard_normal(x.shape[0])
        return _to_int16(dirt(x))
    xi = amplitude * env * np.cos(phi)
    xq = amplitude * env * np.sin(phi)
    if noise > 0:
        xi = xi + noise * rng.standard_normal(xi.shape[0])
        xq = xq + noise * rng.standard_normal(xq.shape[0])
    return np.stack([_to_int16(dirt(xi)), _to_int16(dirt(xq))], axis=1)


# The workloads BASELINE.json names.
def config_c2(noise: float = 0.0, seed: int = 0) -> np.ndarray:
    """10-minute 11 025 Hz capture, 120 LPM / IOC576: N = 7 166 250 samples."""
    return synth_capture(11025.0, noise=noise, seed=seed)


def config_c3(noise: float = 0.0, seed: int = 0) -> np.ndarray:
    """60-minute 48 kHz capture: N0 = 172 800 000 samples."""
    return synth_capture(48000.0, noise=noise, seed=seed, image_lines=7110,
                         black_tail_s=5.0)


def config_c5_member(i: int, noise: float = 0.0) -> tuple[np.ndarray, int]:
    """Capture ``i`` of the 64-capture batch: (samples, lines_per_minute)."""
    lpm = 120 if i % 2 == 0 else 240
    ioc = 576 if (i // 2) % 2 == 0 else 288
    lines = 1200 if ioc == 576 else 600
    if lpm == 240:
        lines *= 2          # same duration as the 120 LPM member
    x = synth_capture(11025.0, noise=noise, seed=i, lpm=lpm, ioc=ioc,
                      image_lines=lines,
                      phasing_lines=60 if lpm == 120 else 120)
    return x, lpm


class Pcm24(np.ndarray):
    """Marker: an int32 array whose values are 24-bit samples (-2**23 .. 2**23 - 1); ``write_wav`` stores three bytes each."""


def write_wav(path: str, fs: int, data: np.ndarray) -> None:
    """Minimal PCM RIFF writer (int16 mono / multi-channel, uint8, int32, 24-bit via ``Pcm24``, float32, float64)."""
    pcm24 = isinstance(data, Pcm24)
    data = np.ascontiguousarray(data)
    ch = 1 if data.ndim == 1 else data.shape[1]
    if data.dtype == np.float32 or data.dtype == np.float64:
        fmt_tag = 3
    else:
        fmt_tag = 1
    bits = data.dtype.itemsize * 8
    raw = data.astype(data.dtype.newbyteorder("<"), copy=False).tobytes()
    if pcm24:
        bits = 24
        raw = np.frombuffer(raw, dtype=np.uint8).reshape(-1, 4)[:, :3].tobytes()
    hdr = struct.pack("<4sI4s4sIHHIIHH4sI", b"RIFF", 36 + len(raw), b"WAVE",
                      b"fmt ", 16, fmt_tag, ch, int(fs),
                      int(fs) * ch * bits // 8, ch * bits // 8, bits,
                      b"data", len(raw))
    with open(path, "wb") as fh:
        fh.write(hdr)
        fh.write(raw)
