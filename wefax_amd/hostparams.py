"""Host-side scalar arithmetic of the decode: everything the reference computes
in Python floats / NumPy scalars before or between its array operations.  These
are evaluated here with the same expressions (so the same IEEE doubles come out)
and handed to the native library as plain numbers.
"""
from __future__ import annotations

import json
import math
import os
import struct

import numpy as np

TARGET_RATE = 11025                 # wefax.py:60
DEFAULT_NOTCH = (2600, 1)           # config/config.json:16-17


def load_notch_settings(config_path: str = "config/config.json"):
    """The two constants the hot path reads from the reference's config file
    (wefax.py:63-64; config.py:11 opens the file relative to cwd).  Defaults only when
    the file is absent; a malformed file or a missing key raises, as it does in the reference."""
    try:
        fh = open(config_path)
    except OSError:
        return DEFAULT_NOTCH
    with fh:
        s = json.load(fh)["notch_filter_settings"]
    return int(s["notch_filter_frequency"]), s["notch_filter_quality_factor"]


def load_detector_settings(config_path: str = "config/config.json"):
    """``tones_settings`` and ``sync_pulse_settings`` of the reference's config file as the live-path detectors read them
    (data_packet.py:24-42), in the shape of ``detect.TONES`` / ``detect.SYNC_PULSE``.  (None, None) when the file is absent --
    the caller then keeps the reference's shipped values; a malformed file or a missing key raises, as in the reference."""
    try:
        fh = open(config_path)
    except OSError:
        return None, None
    with fh:
        cfg = json.load(fh)
    t, p = cfg["tones_settings"], cfg["sync_pulse_settings"]
    tones = dict(start_distance=t["start_tone_peaks_minimum_distance"], stop_distance=t["stop_tone_peaks_minimum_distance"],
                 height=t["peaks_minimum_height"], prominence=t["peaks_minimum_prominence"], fmin=t["peaks_minimum_frequency"],
                 fmax=t["peaks_maximum_frequency"], amount_min=t["peaks_minimum_amount"], amount_max=t["peaks_maximum_amount"])
    pulse = dict(height=p["peaks_minimum_height"], prominence=p["peaks_minimum_prominence"], fmin=p["peaks_minimum_frequency"],
                 fmax=p["peaks_maximum_frequency"])
    return tones, pulse


def iirnotch(w0: float, q: float, fs: float):
    """scipy.signal.iirnotch (wefax.py:68): second-order notch, -3 dB bandwidth w0/Q."""
    w0 = 2 * float(w0) / fs
    if w0 > 1.0 or w0 < 0.0:
        raise ValueError("w0 should be such that 0 < w0 < 1")
    bw = w0 / float(q) * np.pi
    w0 = w0 * np.pi
    beta = np.tan(bw / 2.0)
    gain = 1.0 / (1.0 + beta)
    b = gain * np.array([1.0, -2.0 * np.cos(w0), 1.0])
    a = np.array([1.0, -2.0 * gain * np.cos(w0), (2.0 * gain - 1.0)])
    return b, a


def odd_extension(x: np.ndarray, edge: int = 9):
    """The samples scipy.signal.filtfilt (wefax.py:72) puts before and after ``x`` (scipy.signal._arraytools.odd_ext:
    ``2 x[0] - x[edge:0:-1]`` and ``2 x[-1] - x[-2:-(edge+2):-1]``), evaluated -- as scipy does -- in x's OWN dtype: uint8 and
    int32 captures wrap, float32 ones round to float32.  Returned as two float64 arrays (left in time order, right in time
    order) for the native notch, which otherwise would form the extension from the float64 copy it is handed."""
    x = np.asarray(x)
    with np.errstate(over="ignore"):
        left = 2 * x[0:1] - x[edge:0:-1]
        right = 2 * x[-1:] - x[-2:-(edge + 2):-1]
    return left.astype(np.float64), right.astype(np.float64)


def percentile_plan(n: int, q_percent: float):
    """np.percentile(., q) 'linear' (wefax.py:196): the two order-statistic ranks and
    the lerp weight, as numpy/lib/_function_base_impl.py::_quantile derives them."""
    q = np.true_divide(q_percent, 100)
    virtual = (n - 1) * q
    prev = math.floor(virtual)
    nxt = prev + 1
    gamma = float(virtual - prev)
    if virtual >= n - 1:
        prev = nxt = n - 1
    if virtual < 0:
        prev = nxt = 0
    return int(prev), int(nxt), gamma


def sync_constants(sample_rate: int, frame_len: float):
    """wefax.py:223-229: run lengths of the sync pattern and the peak spacing."""
    samples = lambda x: int(x * frame_len * sample_rate)  # noqa: E731
    return samples(0.005), samples(0.001), int(frame_len * sample_rate * 0.8)


def read_wav(path: str):
    """(sample_rate, ndarray) from a RIFF/WAVE file, PCM or IEEE float, in the
    dtypes scipy.io.wavfile.read (wefax.py:349) returns: uint8, int16, int32
    (24-bit left-justified), float32, float64; [n] or [n, channels]."""
    with open(path, "rb") as fh:
        blob = memoryview(fh.read())           # slices below are views: the samples are copied once, by the upload
    if len(blob) < 12 or blob[:4] != b"RIFF" or blob[8:12] != b"WAVE":
        raise ValueError("File format not understood. Only 'RIFF' and 'WAVE' supported.")
    pos, fmt, payload = 12, None, None
    while pos + 8 <= len(blob):
        cid, size = bytes(blob[pos:pos + 4]), struct.unpack_from("<I", blob, pos + 4)[0]
        body = pos + 8
        if cid == b"fmt ":
            tag, ch, rate, _bps, _align, bits = struct.unpack_from("<HHIIHH", blob, body)
            if tag == 0xFFFE and size >= 40:
                tag = struct.unpack_from("<H", blob, body + 24)[0]
            fmt = (tag, ch, rate, bits)
        elif cid == b"data":
            payload = blob[body:body + size]
            break
        pos = body + size + (size & 1)
    if fmt is None or payload is None:
        raise ValueError("Incomplete wav file: missing fmt or data chunk")
    tag, ch, rate, bits = fmt
    if tag == 1 and bits == 24:
        raw = np.frombuffer(payload[:len(payload) // 3 * 3], dtype=np.uint8).reshape(-1, 3)
        wide = np.zeros((raw.shape[0], 4), dtype=np.uint8)
        wide[:, 1:] = raw
        a = wide.view("<i4").reshape(-1)
    else:
        table = {(1, 8): np.uint8, (1, 16): "<i2", (1, 32): "<i4", (3, 32): "<f4", (3, 64): "<f8"}
        if (tag, bits) not in table:
            raise ValueError(f"Unsupported wav format tag {tag:#x} with {bits} bits")
        dt = np.dtype(table[(tag, bits)])
        a = np.frombuffer(payload[:len(payload) // dt.itemsize * dt.itemsize], dtype=dt)
    frames = a.shape[0] // ch
    a = a[:frames * ch]
    if ch > 1:
        a = a.reshape(frames, ch)
    return int(rate), np.array(a)
