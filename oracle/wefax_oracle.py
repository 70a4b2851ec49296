"""ORACLE -- CPU restatement of wojlin/WEFAX ``wefax.py: Demodulator.process()``.

TEST INFRASTRUCTURE ONLY.  Nothing in the shipped package (``wefax_amd/``) may
import this module; only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` use it, and only as the checker / the CPU
baseline -- never as the thing measured or shipped.

Parity status: PINNED.  The reference's own tests pin no numbers for this path
(/root/reference/tests/tests.py never calls ``process()``) and its only
full-length recording is a missing blob, so the oracle is pinned against the
reference itself: ``tests/golden/make_golden.py`` imports
/root/reference/wefax.py in the build container, runs it on the inputs under
``tests/golden/inputs/`` and stores every stage; ``tests/test_oracle_golden.py``
checks this restatement against those vectors (uint8 stream, sync peaks,
``start_frame``, final image, exceptions and progress messages bit-exact;
float stages to 1e-9 relative).

Third-party code the reference calls on this path is not vendored in
/root/reference (requirements.txt pins scipy==1.10.0, numpy==1.24.2,
Pillow==9.4.0); the goldens were produced with scipy 1.15.3 / numpy 2.2.6 /
Pillow 12.2.0.  Their published algorithms are restated here with numpy only:

  scipy.io.wavfile.read      -> read_wav
  scipy.signal.resample      -> resample_fft        (wefax.py:384)
  scipy.signal.iirnotch      -> iirnotch            (wefax.py:68)
  scipy.signal.filtfilt      -> filtfilt_biquad     (wefax.py:72)
  scipy.signal.hilbert       -> hilbert_fft         (wefax.py:174)
  scipy.signal.medfilt(.,5)  -> medfilt5            (wefax.py:175)
  PIL.Image.resize((w,4h))   -> resize_rows_bicubic (wefax.py:325)

Sequential loops live in oracle_kernels.c (built by oracle/Makefile); pure
Python fall-backs are used when the library is missing (small inputs only).
"""
from __future__ import annotations

import ctypes
import math
import os
import struct
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "_build", "liboracle.so")
_lib = None

NOTCH_F0 = 2600        # /root/reference/config/config.json:16, cast int() at wefax.py:63
NOTCH_Q = 1            # config.json:17
TARGET_RATE = 11025    # wefax.py:60


def _load_lib():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(_LIB_PATH):
        try:
            subprocess.run(["make", "-C", _HERE, "-s"], check=True,
                           stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        except Exception:
            _lib = False
            return _lib
    try:
        lib = ctypes.CDLL(_LIB_PATH)
    except OSError:
        _lib = False
        return _lib
    dp = ctypes.POINTER(ctypes.c_double)
    lib.wfo_lfilter_biquad.argtypes = [dp, dp, dp, dp, ctypes.c_size_t, dp]
    lib.wfo_lfilter_biquad.restype = None
    i64p = ctypes.POINTER(ctypes.c_int64)
    lib.wfo_pick_peaks.argtypes = [i64p, ctypes.c_size_t, ctypes.c_int64, ctypes.c_int,
                                   i64p, i64p, ctypes.POINTER(ctypes.c_int)]
    lib.wfo_pick_peaks.restype = ctypes.c_int
    u8p = ctypes.POINTER(ctypes.c_uint8)
    i32p = ctypes.POINTER(ctypes.c_int32)
    lib.wfo_resize_vertical_8bpc.argtypes = [u8p, ctypes.c_int, ctypes.c_int, u8p,
                                             ctypes.c_int, i32p, ctypes.c_int, i32p]
    lib.wfo_resize_vertical_8bpc.restype = None
    _lib = lib
    return _lib


def _ptr(a, t):
    return a.ctypes.data_as(ctypes.POINTER(t))


# --------------------------------------------------------------------------
# a3: wav ingest  (wefax.py:348-358 -> scipy.io.wavfile.read)
# --------------------------------------------------------------------------
def read_wav(path: str):
    """(sample_rate, data) like scipy.io.wavfile.read for PCM / IEEE-float RIFF.

    uint8 for 8-bit, int16, int32 (24-bit is left-justified into int32, as scipy
    >= 1.6 does), float32/float64; shape [n] for one channel else [n, channels].
    """
    with open(path, "rb") as fh:
        blob = fh.read()
    if blob[:4] != b"RIFF" or blob[8:12] != b"WAVE":
        raise ValueError("File format not understood. Only 'RIFF' and 'WAVE' supported.")
    pos = 12
    fmt = None
    data = None
    while pos + 8 <= len(blob):
        cid = blob[pos:pos + 4]
        size = struct.unpack_from("<I", blob, pos + 4)[0]
        body = pos + 8
        if cid == b"fmt ":
            tag, ch, rate, _br, align, bits = struct.unpack_from("<HHIIHH", blob, body)
            if tag == 0xFFFE and size >= 40:   # WAVE_FORMAT_EXTENSIBLE
                tag = struct.unpack_from("<H", blob, body + 24)[0]
            fmt = (tag, ch, rate, align, bits)
        elif cid == b"data":
            if fmt is None:
                raise ValueError("No fmt chunk before data")
            data = blob[body:body + size]
            break
        pos = body + size + (size & 1)
    if fmt is None or data is None:
        raise ValueError("No data chunk")
    tag, ch, rate, align, bits = fmt
    if tag == 1:
        if bits == 8:
            a = np.frombuffer(data, dtype=np.uint8)
        elif bits == 16:
            a = np.frombuffer(data, dtype="<i2")
        elif bits == 32:
            a = np.frombuffer(data, dtype="<i4")
        elif bits == 24:
            raw = np.frombuffer(data[:len(data) // 3 * 3], dtype=np.uint8).reshape(-1, 3)
            a = np.zeros((raw.shape[0], 4), dtype=np.uint8)
            a[:, 1:] = raw
            a = a.view("<i4").reshape(-1)
        else:
            raise ValueError(f"Unsupported bit depth: {bits}")
    elif tag == 3:
        a = np.frombuffer(data, dtype="<f4" if bits == 32 else "<f8")
    else:
        raise ValueError(f"Unknown wave file format: {tag:#x}")
    n = a.shape[0] // ch
    a = a[:n * ch]
    if ch > 1:
        a = a.reshape(n, ch)
    return int(rate), a.copy()


# --------------------------------------------------------------------------
# a4: stereo merge  (wefax.py:360-373)
# --------------------------------------------------------------------------
def merge_channels(data: np.ndarray) -> np.ndarray:
    """np.divide(np.add(L, R), 2) on numpy scalars: the add happens in the
    sample dtype (int16 / uint8 / int32 wrap), the divide gives float64 for the
    integer formats -- and float32 for a float32 wav, so the list the reference
    hands to filtfilt (wefax.py:72) becomes a float32 array there and the odd
    extension at its two ends is evaluated in float32 (pinned by the golden
    stereo_f32_240: the first audio sample differs by 1.7e-10 otherwise)."""
    l, r = data[:, 0], data[:, 1]
    with np.errstate(over="ignore"):
        s = np.add(l, r)            # stays in the input dtype -> wraps
    out = np.divide(s, 2)
    return out if out.dtype == np.float32 else out.astype(np.float64)


# --------------------------------------------------------------------------
# a5: FFT resample  (wefax.py:384 -> scipy.signal.resample, real input)
# --------------------------------------------------------------------------
def resample_fft(x, num: int) -> np.ndarray:
    x = np.asarray(x)
    if x.dtype.kind != "f" or x.dtype.itemsize < 8:
        x = x.astype(np.float64)
    nx = x.shape[0]
    X = np.fft.rfft(x)
    Y = np.zeros(num // 2 + 1, dtype=X.dtype)
    n = min(num, nx)
    nyq = n // 2 + 1
    Y[:nyq] = X[:nyq]
    if n % 2 == 0:
        if num < nx:
            Y[n // 2] *= 2.0
        elif nx < num:
            Y[n // 2] *= 0.5
    y = np.fft.irfft(Y, num)
    y *= (float(num) / float(nx))
    return y


# --------------------------------------------------------------------------
# a6: notch / slope filter  (wefax.py:63-72)
# --------------------------------------------------------------------------
def iirnotch(w0: float, q: float, fs: float):
    w0 = float(w0)
    q = float(q)
    w0 = 2 * w0 / fs
    bw = w0 / q
    bw = bw * np.pi
    w0 = w0 * np.pi
    beta = np.tan(bw / 2.0)
    gain = 1.0 / (1.0 + beta)
    b = gain * np.array([1.0, -2.0 * np.cos(w0), 1.0])
    a = np.array([1.0, -2.0 * gain * np.cos(w0), (2.0 * gain - 1.0)])
    return b, a


def lfilter_zi(b, a):
    """Steady-state step-response state (scipy.signal.lfilter_zi, a[0] == 1)."""
    iminus_a = np.array([[1.0 + a[1], -1.0], [a[2], 1.0]])
    bb = b[1:] - a[1:] * b[0]
    return np.linalg.solve(iminus_a, bb)


def lfilter_biquad(b, a, x, zi):
    x = np.ascontiguousarray(x, dtype=np.float64)
    y = np.empty_like(x)
    lib = _load_lib()
    b = np.ascontiguousarray(b, dtype=np.float64)
    a = np.ascontiguousarray(a, dtype=np.float64)
    zi = np.ascontiguousarray(zi, dtype=np.float64)
    if lib:
        d = ctypes.c_double
        lib.wfo_lfilter_biquad(_ptr(b, d), _ptr(a, d), _ptr(x, d), _ptr(y, d),
                               x.shape[0], _ptr(zi, d))
        return y
    z0, z1 = float(zi[0]), float(zi[1])
    b0, b1, b2, a1, a2 = map(float, (b[0], b[1], b[2], a[1], a[2]))
    for i in range(x.shape[0]):
        xi = float(x[i])
        yi = z0 + b0 * xi
        z0 = z1 + b1 * xi - a1 * yi
        z1 = b2 * xi - a2 * yi
        y[i] = yi
    return y


def odd_ext(x: np.ndarray, edge: int) -> np.ndarray:
    """scipy.signal._arraytools.odd_ext -- computed in x's own dtype, so an
    int16 capture wraps exactly as it does inside scipy.signal.filtfilt."""
    left_end = x[0:1]
    left_ext = x[edge:0:-1]
    right_end = x[-1:]
    right_ext = x[-2:-(edge + 2):-1]
    with np.errstate(over="ignore"):
        return np.concatenate((2 * left_end - left_ext, x, 2 * right_end - right_ext))


def filtfilt_biquad(b, a, x) -> np.ndarray:
    """scipy.signal.filtfilt(b, a, x) with its defaults (padtype='odd',
    padlen=3*max(len(a),len(b))=9, method='pad')."""
    x = np.asarray(x)
    edge = 9
    if x.shape[0] <= edge:
        raise ValueError("The length of the input vector x must be greater than "
                         "padlen, which is %d." % edge)
    ext = odd_ext(x, edge)
    zi = lfilter_zi(b, a)
    x0 = ext[0:1]
    y = lfilter_biquad(b, a, ext, zi * x0)
    y0 = y[-1:]
    y = lfilter_biquad(b, a, y[::-1], zi * y0)
    y = y[::-1]
    return y[edge:-edge].copy()


# --------------------------------------------------------------------------
# a7: demodulate  (wefax.py:166-183)
# --------------------------------------------------------------------------
def hilbert_fft(x: np.ndarray) -> np.ndarray:
    x = np.asarray(x, dtype=np.float64)
    n = x.shape[0]
    # scipy.fft.fft of a REAL array runs pocketfft's r2c and mirrors the upper
    # half; np.fft.fft would run a full complex transform (different rounding).
    half = np.fft.rfft(x)
    xf = np.empty(n, dtype=np.complex128)
    xf[:half.shape[0]] = half
    if n > 1:
        xf[half.shape[0]:] = np.conj(half[1:(n + 1) // 2][::-1])
    h = np.zeros(n, dtype=np.complex128)
    if n % 2 == 0:
        h[0] = h[n // 2] = 1
        h[1:n // 2] = 2
    else:
        h[0] = 1
        h[1:(n + 1) // 2] = 2
    return np.fft.ifft(xf * h)


def medfilt5(x: np.ndarray) -> np.ndarray:
    """scipy.signal.medfilt(x, 5): zero padding beyond both ends."""
    x = np.asarray(x)
    n = x.shape[0]
    p = np.zeros(n + 4, dtype=x.dtype)
    p[2:n + 2] = x
    win = np.stack([p[k:k + n] for k in range(5)], axis=0)
    win.sort(axis=0)
    return win[2].copy()


def demodulate(x: np.ndarray) -> np.ndarray:
    return medfilt5(np.abs(hilbert_fft(x)))


# --------------------------------------------------------------------------
# a8: digitalize  (wefax.py:185-216)
# --------------------------------------------------------------------------
def digitalize(env: np.ndarray):
    low, high = np.percentile(env, (0.5, 99.5))
    delta = high - low
    with np.errstate(invalid="ignore", divide="ignore"):
        d = np.round(255 * (env - low) / delta)
    d[d < 0] = 0
    d[d > 255] = 255
    if np.isnan(d).any():
        # the reference does [int(point) ...] (wefax.py:216) -> int(nan) raises
        raise ValueError("cannot convert float NaN to integer")
    return d.astype(np.uint8), float(low), float(high)


# --------------------------------------------------------------------------
# a9: sync search  (wefax.py:218-294)
# --------------------------------------------------------------------------
def sync_constants(sample_rate: int, frame_len: float):
    samples = lambda x: int(x * frame_len * sample_rate)   # noqa: E731  wefax.py:223
    n1 = samples(0.005)
    n0 = samples(0.001)
    mindistance = int(frame_len * sample_rate * 0.8)       # wefax.py:229
    return n1, n0, mindistance


def sync_correlation(d: np.ndarray, n1: int, n0: int) -> np.ndarray:
    """corr[i] = dot(pattern - 128, d[i:i+L] - 128), i in [0, len(d) - L)
    (wefax.py:225,232-236), exact in int64 via prefix sums: the shifted
    pattern is -127 on the two outer runs and -128 on the middle run."""
    L = 2 * n1 + n0
    n = d.shape[0]
    ncorr = n - L
    if ncorr <= 0:
        return np.zeros(0, dtype=np.int64)
    c = np.zeros(n + 1, dtype=np.int64)
    np.cumsum(d.astype(np.int64) - 128, out=c[1:])
    i = np.arange(ncorr)
    return -127 * (c[i + L] - c[i]) - (c[i + n1 + n0] - c[i + n1])


def pick_peaks(corr: np.ndarray, mindistance: int, max_peaks: int = 100):
    corr = np.ascontiguousarray(corr, dtype=np.int64)
    lib = _load_lib()
    if lib:
        pos = np.zeros(max_peaks + 1, dtype=np.int64)
        first = np.zeros(max_peaks + 1, dtype=np.int64)
        hit = ctypes.c_int(0)
        k = lib.wfo_pick_peaks(_ptr(corr, ctypes.c_int64), corr.shape[0], mindistance,
                               max_peaks, _ptr(pos, ctypes.c_int64),
                               _ptr(first, ctypes.c_int64), ctypes.byref(hit))
        return pos[:k].tolist(), first[:k].tolist(), bool(hit.value)
    peaks = [[0, 0]]
    first = [0]
    hit = False
    for i in range(corr.shape[0]):
        c = int(corr[i])
        if i - peaks[-1][0] > mindistance:
            peaks.append([i, c])
            first.append(i)
        elif c > peaks[-1][1]:
            peaks[-1] = [i, c]
        if len(peaks) == max_peaks:
            hit = True
            break
    return [p[0] for p in peaks], first, hit


def group_peaks(peaks, sample_rate: int, frame_len: float):
    """wefax.py:263-294 including its quirks: the loop bound is len(clear) but
    the index goes into ``peaks``; the last open group is never appended;
    max() of an empty list raises ValueError."""
    def dev(x):
        return (frame_len * sample_rate + 500) > x > (frame_len * sample_rate - 500)
    clear = [peaks[i] for i in range(1, len(peaks) - 1) if dev(peaks[i] - peaks[i - 1])]
    groups = []
    group = []
    for i in range(1, len(clear) - 1):
        if dev(peaks[i] - peaks[i - 1]):
            group.append(peaks[i])
        else:
            groups.append(group)
            group = []
    return max(groups, key=len)       # ValueError("max() arg is an empty sequence")


# --------------------------------------------------------------------------
# a10: image assembly  (wefax.py:296-327)
# --------------------------------------------------------------------------
def _bicubic(x: float) -> float:
    a = -0.5
    if x < 0.0:
        x = -x
    if x < 1.0:
        return ((a + 2.0) * x - (a + 3.0)) * x * x + 1
    if x < 2.0:
        return (((x - 5) * x + 8) * x - 4) * a
    return 0.0


def pillow_vertical_coeffs(h_in: int, h_out: int):
    """Pillow Resample.c precompute_coeffs + normalize_coeffs_8bpc for the
    BICUBIC filter (support 2): (kk int32 [h_out, ksize], bounds int32 [h_out, 2])."""
    scale = float(h_in) / h_out
    filterscale = max(scale, 1.0)
    support = 2.0 * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    kk = np.zeros((h_out, ksize), dtype=np.int32)
    bounds = np.zeros((h_out, 2), dtype=np.int32)
    ss = 1.0 / filterscale
    for yy in range(h_out):
        center = (yy + 0.5) * scale
        ymin = int(center - support + 0.5)
        if ymin < 0:
            ymin = 0
        ymax = int(center + support + 0.5)
        if ymax > h_in:
            ymax = h_in
        ymax -= ymin
        w = [_bicubic((y + ymin - center + 0.5) * ss) for y in range(ymax)]
        ww = 0.0
        for v in w:
            ww += v
        for y in range(ymax):
            v = w[y] / ww if ww != 0.0 else w[y]
            kk[yy, y] = int(-0.5 + v * (1 << 22)) if v < 0 else int(0.5 + v * (1 << 22))
        bounds[yy] = (ymin, ymax)
    return kk, bounds


def resize_rows_bicubic(img: np.ndarray, h_out: int) -> np.ndarray:
    """PIL.Image.resize((w, h_out)) on an 8-bit image whose width is unchanged:
    Pillow skips the horizontal pass and runs only ImagingResampleVertical_8bpc."""
    h_in, w = img.shape
    if h_in == 0 or w == 0:
        return np.zeros((h_out, w), dtype=np.uint8)
    kk, bounds = pillow_vertical_coeffs(h_in, h_out)
    img = np.ascontiguousarray(img, dtype=np.uint8)
    out = np.empty((h_out, w), dtype=np.uint8)
    lib = _load_lib()
    if lib:
        lib.wfo_resize_vertical_8bpc(_ptr(img, ctypes.c_uint8), w, h_in,
                                     _ptr(out, ctypes.c_uint8), h_out,
                                     _ptr(kk, ctypes.c_int32), kk.shape[1],
                                     _ptr(np.ascontiguousarray(bounds), ctypes.c_int32))
        return out
    src = img.astype(np.int64)
    for yy in range(h_out):
        ymin, cnt = int(bounds[yy, 0]), int(bounds[yy, 1])
        acc = np.full(w, 1 << 21, dtype=np.int64)
        for k in range(cnt):
            acc += src[ymin + k] * int(kk[yy, k])
        out[yy] = np.clip(acc >> 22, 0, 255).astype(np.uint8)
    return out


def lines_to_image(d: np.ndarray, frame_len: float, sample_rate: int) -> np.ndarray:
    """wefax.py:296-327: rows of w = int(T*fs) samples, inverted, then 4x rows."""
    w = int(frame_len * sample_rate)
    h = d.shape[0] // w
    base = (255 - d[:h * w].astype(np.int16)).astype(np.uint8).reshape(h, w)
    return resize_rows_bicubic(base, 4 * h)


# --------------------------------------------------------------------------
# a2: the whole path  (wefax.py:46-93)
# --------------------------------------------------------------------------
# ---------------------------------------------------------------------------
# Live path, one audio packet: data_packet.py:408-464.  The same chain as the file path with
# different constants: notch designed at the packet's own sample rate, medfilt 3, per-packet percentiles,
# np.rint, and a 1e-6 guard on the denominator.
# ---------------------------------------------------------------------------
def medfilt3(x: np.ndarray) -> np.ndarray:
    """scipy.signal.medfilt(x, 3): zero padding beyond both ends (data_packet.py:446)."""
    x = np.asarray(x)
    n = x.shape[0]
    p = np.zeros(n + 2, dtype=x.dtype)
    p[1:n + 1] = x
    win = np.stack([p[k:k + n] for k in range(3)], axis=0)
    win.sort(axis=0)
    return win[1].copy()


def process_packet(samples: np.ndarray, sample_rate: int, notch=(NOTCH_F0, NOTCH_Q)) -> dict:
    """DataPacket.__process_samples (data_packet.py:408-418): notch filtfilt (:420-434) -> |hilbert| + medfilt 3
    (:436-448) -> per-packet percentiles, rint with the 1e-6 guard, clip (:450-464)."""
    b, a = iirnotch(int(notch[0]), notch[1], sample_rate)
    notched = filtfilt_biquad(b, a, np.asarray(samples))
    env = medfilt3(np.abs(hilbert_fft(notched)))
    low, high = np.percentile(env, (0.5, 99.5))
    delta = high - low
    d = np.rint(255 * (env - low) / (delta + 0.000001))
    d[d < 0] = 0
    d[d > 255] = 255
    return {"notched": notched, "envelope": env, "low": float(low), "high": float(high), "samples": d.astype(np.uint8)}


# ---------------------------------------------------------------------------
# Detectors of the live path (data_packet.py:301-406): start / stop tone and sync pulse.
# scipy.signal.find_peaks is restated for the three conditions the reference uses (height, distance,
# prominence); pinned by tests/golden/packets.npz (answers of the reference's own DataPacket).
# ---------------------------------------------------------------------------
TONES = dict(start_distance=250, stop_distance=380, height=0.05, prominence=0.2, fmin=800, fmax=3200,
             amount_min=4, amount_max=6)                     # config.json "tones_settings" (data_packet.py:25-33)
SYNC_PULSE = dict(height=0.5, prominence=0.2, fmin=1400, fmax=1600)      # config.json "sync_pulse_settings" (:36-40)


def packet_spectrum(raw: np.ndarray, sample_rate: int):
    """data_packet.py:388-406 __fourier_transform: one-sided frequencies and the amplitude normalised by max + 1e-4."""
    fft = np.fft.fft(raw)
    n = len(fft)
    freq = np.arange(n) / (n / sample_rate)
    h = n // 2
    amp = np.abs(fft[:h] / h)
    return freq[:h], amp / (np.max(amp) + 0.0001)


def _local_maxima(x: np.ndarray) -> np.ndarray:
    """scipy.signal._peak_finding_utils._local_maxima_1d: strict rise, optional plateau, strict fall; plateau midpoint."""
    out = []
    n = len(x)
    i, i_max = 1, n - 1
    while i < i_max:
        if x[i - 1] < x[i]:
            ahead = i + 1
            while ahead < i_max and x[ahead] == x[i]:
                ahead += 1
            if x[ahead] < x[i]:
                out.append((i + ahead - 1) // 2)
                i = ahead
        i += 1
    return np.asarray(out, dtype=np.int64)


def _select_by_distance(peaks: np.ndarray, priority: np.ndarray, distance: float) -> np.ndarray:
    """scipy _select_by_peak_distance: highest peaks first, each removes its neighbours closer than ceil(distance)."""
    n = len(peaks)
    d = math.ceil(distance)
    keep = np.ones(n, dtype=bool)
    order = np.argsort(priority)
    for i in range(n - 1, -1, -1):
        j = order[i]
        if not keep[j]:
            continue
        k = j - 1
        while k >= 0 and peaks[j] - peaks[k] < d:
            keep[k] = False
            k -= 1
        k = j + 1
        while k < n and peaks[k] - peaks[j] < d:
            keep[k] = False
            k += 1
    return keep


def _prominences(x: np.ndarray, peaks: np.ndarray) -> np.ndarray:
    """scipy _peak_prominences with wlen = -1: peak height above the higher of the two lowest points reached before a
    higher sample (or the signal's end) on either side."""
    out = np.empty(len(peaks))
    n = len(x)
    for k, p in enumerate(peaks):
        left_min = x[p]
        i = p
        while i >= 0 and x[i] <= x[p]:
            left_min = min(left_min, x[i])
            i -= 1
        right_min = x[p]
        i = p
        while i < n and x[i] <= x[p]:
            right_min = min(right_min, x[i])
            i += 1
        out[k] = x[p] - max(left_min, right_min)
    return out


def find_peaks(x: np.ndarray, height=None, distance=None, prominence=None):
    """scipy.signal.find_peaks for the conditions the reference passes, in scipy's order: height, distance, prominence.
    Returns (indices, peak_heights)."""
    x = np.asarray(x, dtype=np.float64)
    peaks = _local_maxima(x)
    if height is not None:
        peaks = peaks[x[peaks] >= height]
    if distance is not None:
        peaks = peaks[_select_by_distance(peaks, x[peaks], distance)]
    if prominence is not None:
        peaks = peaks[_prominences(x, peaks) >= prominence]
    return peaks, x[peaks]


def contain_tone(raw: np.ndarray, sample_rate: int, distance: int, cfg=TONES) -> bool:
    """data_packet.py:366-386 __contain_tone."""
    freq, amp = packet_spectrum(raw, sample_rate)
    peaks, _ = find_peaks(amp, height=cfg["height"], distance=distance, prominence=cfg["prominence"])
    in_range = all(cfg["fmin"] <= f <= cfg["fmax"] for f in freq[peaks])
    return bool(in_range and cfg["amount_min"] <= len(peaks) <= cfg["amount_max"])


def contain_start_tone(raw, sample_rate, cfg=TONES) -> bool:        # data_packet.py:344-353
    return contain_tone(raw, sample_rate, cfg["start_distance"], cfg)


def contain_stop_tone(raw, sample_rate, cfg=TONES) -> bool:         # data_packet.py:355-364
    return contain_tone(raw, sample_rate, cfg["stop_distance"], cfg)


def packet_pattern_search(samples: np.ndarray, sample_rate: int):
    """data_packet.py:314-334: correlation with [255] + [0] * k + [255] (both shifted by 128), peaks at least
    0.4 s apart; the first (dummy) entry is dropped."""
    n = len(samples)
    sm = lambda v: int((v / (n / sample_rate)) * n)  # noqa: E731
    k = sm(0.025)
    mind = sm(0.4)
    s = np.asarray(samples, dtype=np.int64) - 128
    ls = k + 2
    cs = np.concatenate(([0], np.cumsum(s)))
    peaks = [(-mind, 0)]
    for i in range(n - ls):
        corr = 127 * s[i] - 128 * (cs[i + k + 1] - cs[i + 1]) + 127 * s[i + k + 1]
        if i - peaks[-1][0] > mind:
            peaks.append((i, corr))
        elif corr > peaks[-1][1]:
            peaks[-1] = (i, corr)
    return [p[0] for p in peaks][1:]


def packet_find_sync_pulse(raw: np.ndarray, samples: np.ndarray, sample_rate: int, cfg=SYNC_PULSE) -> dict:
    """data_packet.py:301-342 find_sync_pulse."""
    freq, amp = packet_spectrum(raw, sample_rate)
    peaks, heights = find_peaks(amp, height=cfg["height"], prominence=cfg["prominence"])
    in_range = all(cfg["fmin"] <= f <= cfg["fmax"] for f in freq[peaks])
    pulses = packet_pattern_search(samples, sample_rate)
    freq_found = bool(in_range and len(peaks) == 1)
    return {"frequency_peak_found": freq_found, "samples_peak_found": bool(len(pulses)),
            "pulse_found": bool(freq_found and len(pulses)), "peaks_fft": [freq[peaks], heights], "peaks_samples": pulses}


# ---------------------------------------------------------------------------
# The reference's loop structure, for CPU TIMING only (BASELINE.md section 3,
# SURVEY.md 8d "faithful_loops"): the same results as the vectorised stages
# above, computed the way wefax.py computes them -- one Python iteration per
# sample.  tests/test_oracle_golden.py checks that both forms agree.
# ---------------------------------------------------------------------------
def merge_channels_loop(data: np.ndarray) -> list:
    """wefax.py:360-373: per frame np.add on two np.int16 scalars (wraps), then /2."""
    out = []
    for frame in data:
        out.append(np.divide(np.add(frame[0], frame[1]), 2))
    return out


def digitalize_loop(env: np.ndarray):
    """wefax.py:185-216: vectorised rounding, then the per-sample int() list build."""
    d, low, high = digitalize(env)
    return [int(p) for p in d.astype(np.float64)], low, high


def pick_peaks_loop(d_list: list, n1: int, n0: int, mindistance: int, max_peaks: int = 100):
    """wefax.py:222-263: np.dot of two Python lists at every offset, peak list updated in place."""
    sync = [1 - 128] * n1 + [0 - 128] * n0 + [1 - 128] * n1
    shifted = [x - 128 for x in d_list]
    peaks = [(0, 0)]
    first = [0]
    hit = False
    for i in range(len(d_list) - len(sync)):
        corr = np.dot(sync, shifted[i:i + len(sync)])
        if i - peaks[-1][0] > mindistance:
            peaks.append((i, corr))
            first.append(i)
        elif corr > peaks[-1][1]:
            peaks[-1] = (i, corr)
        if len(peaks) == max_peaks:
            hit = True
            break
    return [p[0] for p in peaks], first, hit


def lines_to_image_loop(d_list: list, frame_len: float, sample_rate: int) -> np.ndarray:
    """wefax.py:296-327: Image.putpixel per sample, then Pillow's own 4x vertical resize."""
    from PIL import Image
    w = int(frame_len * sample_rate)
    h = len(d_list) // w
    image = Image.new("L", (w, h))
    px = py = 0
    for p in range(len(d_list)):
        image.putpixel((px, py), 255 - d_list[p])
        px += 1
        if px >= w:
            px = 0
            py += 1
            if py >= h:
                break
    return np.asarray(image.resize((w, 4 * h)))


def process(path: str, lines_per_minute: int = 120, want_messages: bool = True, faithful_loops: bool = False) -> dict:
    """Run every stage; returns a dict of stage outputs.  When the reference
    would raise (wefax.py:294) the exception is stored under 'exception' and
    the stages computed so far are kept.  ``faithful_loops`` swaps the vectorised
    merge / list build / sync search / image construction for the reference's
    per-sample Python loops (same results; only its running time is of interest)."""
    r: dict = {"messages": []}
    msgs = r["messages"]

    def progress(title, pct):
        msgs.append(["progress_bar", title, float(pct)])

    frame_len = 1 / (lines_per_minute / 60)             # wefax.py:33
    sr, data = read_wav(path)
    if data.ndim == 2:
        parts = data.shape[0]
        if want_messages:
            for p in range(parts):                      # wefax.py:364-370
                if p % 1000 == 0 or p == parts - 1:
                    progress("merging channels", (p + 1) / parts * 100)
        data = np.asarray(merge_channels_loop(data)) if faithful_loops else merge_channels(data)     # (a list of numpy scalars: float64, or float32 for a float32 wav)
    length = len(data) / sr
    if sr != TARGET_RATE:                               # wefax.py:60
        progress("resampling audio", 0)
        data = resample_fft(data, int(TARGET_RATE * length))
        progress("resampling audio", 100)
        sr = TARGET_RATE
        length = len(data) / sr
    r["sample_rate"], r["length"] = sr, length
    b, a = iirnotch(int(NOTCH_F0), NOTCH_Q, sr)
    audio = filtfilt_biquad(b, a, data)
    r["audio"] = audio
    progress("demodulating signal", 0)
    env = demodulate(audio)
    progress("demodulating signal", 100)
    r["demod"] = env
    progress("digitalizing signal", 0)
    if faithful_loops:
        d_list, low, high = digitalize_loop(env)
        d = np.asarray(d_list, dtype=np.uint8)
    else:
        d, low, high = digitalize(env)
    progress("digitalizing signal", 99)
    progress("digitalizing signal", 100)
    r["digitalized"], r["low"], r["high"] = d, low, high
    n1, n0, mind = sync_constants(sr, frame_len)
    if faithful_loops:
        peaks, first, hit = pick_peaks_loop(d_list, n1, n0, mind)
    else:
        corr = sync_correlation(d, n1, n0)
        peaks, first, hit = pick_peaks(corr, mind)
    r["peaks"] = peaks
    for i in first[1:]:
        progress("finding sync pulse", (i / len(d)) * 100)   # wefax.py:245
    if hit:
        progress("finding sync pulse", 100)                  # wefax.py:257
    try:
        phasing = group_peaks(peaks, sr, frame_len)
    except ValueError as e:
        r["exception"] = e
        return r
    r["phasing_signals"] = phasing
    start = phasing[-1] if phasing else 0               # wefax.py:80
    r["start_frame"] = start
    tail = d[start:]
    w = int(frame_len * sr)
    h = len(tail) // w
    # progress of the putpixel loop (wefax.py:302-323): every 50th finished row,
    # then 100 when the last row is finished
    for py in range(h):
        if py % 50 == 0:
            progress("converting signal to image", (py + 1) / h * 100)
    if h > 0 and len(tail) >= 1:
        progress("converting signal to image", 100)
    r["image"] = lines_to_image_loop(d_list[start:], frame_len, sr) if faithful_loops else lines_to_image(tail, frame_len, sr)
    msgs.append(["message", "convert_end", None])
    return r
