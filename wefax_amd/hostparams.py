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


_READ_POOL = None
_READ_SLICE = 2 << 20


def _read_into(fd: int, dest: np.ndarray, offset: int):
    """File bytes [offset, offset + dest.nbytes) into ``dest`` (uint8): slices of 2 MiB by a few threads -- copying a file out of the
    page cache is memcpy-bound per thread (13 MB: 2 ms on one), and os.preadv releases the GIL."""
    global _READ_POOL
    mv = memoryview(dest)
    n = dest.nbytes

    def part(lo):
        hi, got = min(n, lo + _READ_SLICE), lo
        while got < hi:
            k = os.preadv(fd, [mv[got:hi]], offset + got)
            if k <= 0:
                raise ValueError("Incomplete wav file: data chunk is shorter than its header says")
            got += k

    starts = range(0, n, _READ_SLICE)
    if len(starts) <= 1:
        for lo in starts:
            part(lo)
        return
    if _READ_POOL is None:
        from concurrent.futures import ThreadPoolExecutor
        _READ_POOL = ThreadPoolExecutor(max_workers=min(12, os.cpu_count() or 1), thread_name_prefix="wfx-read")
    list(_READ_POOL.map(part, starts))


def _wav_header(fh):
    """(format tag, channels, rate, bits, offset of the samples, their byte count) of an open RIFF/WAVE file."""
    head = fh.read(12)
    if len(head) < 12 or head[:4] != b"RIFF" or head[8:12] != b"WAVE":
        raise ValueError("File format not understood. Only 'RIFF' and 'WAVE' supported.")
    file_size = os.fstat(fh.fileno()).st_size
    pos, fmt, payload = 12, None, None
    while pos + 8 <= file_size:
        fh.seek(pos)
        hdr = fh.read(8)
        if len(hdr) < 8:
            break
        cid, size = hdr[:4], struct.unpack("<I", hdr[4:])[0]
        body = pos + 8
        if cid == b"fmt ":
            blob = fh.read(min(size, 64))
            tag, ch, rate, _bps, _align, bits = struct.unpack_from("<HHIIHH", blob, 0)
            if tag == 0xFFFE and size >= 40:
                tag = struct.unpack_from("<H", blob, 24)[0]
            fmt = (tag, ch, rate, bits)
        elif cid == b"data":
            payload = (body, max(0, min(size, file_size - body)))
            break
        pos = body + size + (size & 1)
    if fmt is None or payload is None:
        raise ValueError("Incomplete wav file: missing fmt or data chunk")
    return (*fmt[:2], fmt[2], fmt[3], *payload)


_WAV_ITEM = {(1, 8): 1, (1, 16): 2, (1, 24): 3, (1, 32): 4, (3, 32): 4, (3, 64): 8}


def wav_info(path: str):
    """(sample_rate, frames, channels) from the headers alone -- what file_info (wefax.py:342-346) reports, without the samples."""
    with open(path, "rb") as fh:
        tag, ch, rate, bits, _body, nbytes = _wav_header(fh)
    if (tag, bits) not in _WAV_ITEM:
        raise ValueError(f"Unsupported wav format tag {tag:#x} with {bits} bits")
    return int(rate), (nbytes // (_WAV_ITEM[(tag, bits)] * ch) if ch else 0), int(ch)


def wav_pcm16_layout(path: str):
    """(sample_rate, channels, offset of the samples in the file, frames) of a 16-bit PCM wav with one or two channels -- the
    captures that reach the device exactly as they lie in the file -- else None (scipy.io.wavfile.read's other dtypes, more
    channels: ``read_wav`` handles those)."""
    with open(path, "rb") as fh:
        tag, ch, rate, bits, body, nbytes = _wav_header(fh)
    if tag != 1 or bits != 16 or ch not in (1, 2):
        return None
    return int(rate), int(ch), int(body), int(nbytes // (2 * ch))


def read_wav(path: str, alloc=None):
    """(sample_rate, ndarray) from a RIFF/WAVE file, PCM or IEEE float, in the
    dtypes scipy.io.wavfile.read (wefax.py:349) returns: uint8, int16, int32
    (24-bit left-justified), float32, float64; [n] or [n, channels].

    The samples are read straight into the array that is returned -- one copy out of the page cache, by a few threads.  ``alloc``
    (bytes -> uint8 array of at least that size) lets the caller provide the memory: the decoder passes its context's page-locked
    staging buffer, from which the upload is a DMA (the array is then only valid until that context reads its next file)."""
    with open(path, "rb") as fh:
        tag, ch, rate, bits, body, nbytes = _wav_header(fh)
        if tag == 1 and bits == 24:
            raw = np.empty(nbytes // 3 * 3, dtype=np.uint8)
            _read_into(fh.fileno(), raw, body)
            raw = raw.reshape(-1, 3)
            wide = np.zeros((raw.shape[0], 4), dtype=np.uint8)
            wide[:, 1:] = raw
            a = wide.view("<i4").reshape(-1)
        else:
            table = {(1, 8): np.uint8, (1, 16): "<i2", (1, 32): "<i4", (3, 32): "<f4", (3, 64): "<f8"}
            if (tag, bits) not in table:
                raise ValueError(f"Unsupported wav format tag {tag:#x} with {bits} bits")
            dt = np.dtype(table[(tag, bits)])
            nbytes = nbytes // (dt.itemsize * ch) * (dt.itemsize * ch) if ch > 0 else 0
            store = None
            if alloc is not None and nbytes:
                store = alloc(nbytes)
            if store is None:
                store = np.empty(nbytes, dtype=np.uint8)
            store = store[:nbytes]
            _read_into(fh.fileno(), store, body)
            a = store.view(dt)
    frames = a.shape[0] // ch
    a = a[:frames * ch]
    if ch > 1:
        a = a.reshape(frames, ch)
    return int(rate), a
