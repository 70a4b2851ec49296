"""Host side of the live path's detectors (/root/reference/data_packet.py:301-406): the peak conditions the
reference evaluates with ``scipy.signal.find_peaks`` on a packet's one-sided amplitude spectrum, and the search
for the 25 ms sync pulse in its digitised samples.  The spectrum itself comes from the GPU
(``wfx_packet_spectrum``); what is left here is a few thousand values per one-second packet.

Defaults are the reference's ``config/config.json`` (``tones_settings``, ``sync_pulse_settings``).
"""
from __future__ import annotations

import math

import numpy as np

TONES = dict(start_distance=250, stop_distance=380, height=0.05, prominence=0.2, fmin=800, fmax=3200,
             amount_min=4, amount_max=6)
SYNC_PULSE = dict(height=0.5, prominence=0.2, fmin=1400, fmax=1600)


def normalise(amp: np.ndarray) -> np.ndarray:
    """data_packet.py:404: amplitude / (max(amplitude) + 0.0001)."""
    return amp / (np.max(amp) + 0.0001)


def frequencies(n: int, sample_rate: int) -> np.ndarray:
    """data_packet.py:396-401: arange(n) / (n / sample_rate), first n // 2 entries."""
    return (np.arange(n) / (n / sample_rate))[:n // 2]


def local_maxima(x: np.ndarray) -> np.ndarray:
    """Indices of the local maxima as scipy.signal.find_peaks defines them: a strict rise, then a strict fall, with an
    optional plateau in between (its midpoint, rounded down, is the peak); the two end samples never are."""
    n = x.shape[0]
    if n < 3:
        return np.empty(0, dtype=np.int64)
    # runs of equal values: a run is a peak when the runs on both sides are lower; the first and last run never are
    change = np.flatnonzero(x[1:] != x[:-1]) + 1
    starts = np.concatenate(([0], change))
    ends = np.concatenate((change - 1, [n - 1]))
    vals = x[starts]
    if vals.shape[0] < 3:
        return np.empty(0, dtype=np.int64)
    is_peak = (vals[:-2] < vals[1:-1]) & (vals[2:] < vals[1:-1])
    return ((starts[1:-1][is_peak] + ends[1:-1][is_peak]) // 2).astype(np.int64)


def select_by_distance(peaks: np.ndarray, priority: np.ndarray, distance: float) -> np.ndarray:
    """Keep mask: the highest peaks first, each removing its neighbours closer than ceil(distance) samples."""
    n = peaks.shape[0]
    d = math.ceil(distance)
    keep = np.ones(n, dtype=bool)
    order = np.argsort(priority)
    for j in order[::-1]:
        if not keep[j]:
            continue
        lo = np.searchsorted(peaks, peaks[j] - d, side="right")      # peaks[j] - peaks[k] < d  <=>  peaks[k] > peaks[j] - d
        hi = np.searchsorted(peaks, peaks[j] + d, side="left")
        keep[lo:j] = False
        keep[j + 1:hi] = False
    return keep


def prominences(x: np.ndarray, peaks: np.ndarray) -> np.ndarray:
    """Height of each peak above the higher of the two minima between it and the next higher sample on either side."""
    out = np.empty(peaks.shape[0])
    n = x.shape[0]
    for k, p in enumerate(peaks):
        higher = np.flatnonzero(x[:p] > x[p])
        left = higher[-1] + 1 if higher.size else 0
        higher = np.flatnonzero(x[p + 1:] > x[p])
        right = p + 1 + higher[0] if higher.size else n
        out[k] = x[p] - max(x[left:p + 1].min(), x[p:right].min())
    return out


def find_peaks(x: np.ndarray, height=None, distance=None, prominence=None):
    """(indices, heights) under scipy.signal.find_peaks' conditions, applied in scipy's order."""
    x = np.asarray(x, dtype=np.float64)
    peaks = local_maxima(x)
    if height is not None:
        peaks = peaks[x[peaks] >= height]
    if distance is not None:
        peaks = peaks[select_by_distance(peaks, x[peaks], distance)]
    if prominence is not None:
        peaks = peaks[prominences(x, peaks) >= prominence]
    return peaks, x[peaks]


def contain_tone(freq: np.ndarray, amp: np.ndarray, distance: int, cfg=TONES) -> bool:
    """data_packet.py:366-386: 4..6 spectral peaks, all between 800 and 3200 Hz."""
    peaks, _ = find_peaks(amp, height=cfg["height"], distance=distance, prominence=cfg["prominence"])
    f = freq[peaks]
    return bool(np.all((f >= cfg["fmin"]) & (f <= cfg["fmax"])) and cfg["amount_min"] <= peaks.shape[0] <= cfg["amount_max"])


def pattern_search(samples: np.ndarray, sample_rate: int):
    """data_packet.py:314-334: positions of the maxima of the correlation with a 25 ms black gap between two white
    samples, at least 0.4 s apart.  The reference's scan (append when further than mindistance from the last peak, else
    replace it by any greater value) ends each peak on the first maximum of its moving window, which is found by
    window-maximum jumps instead of a step per sample."""
    n = samples.shape[0]
    sm = lambda v: int((v / (n / sample_rate)) * n)  # noqa: E731
    k, mind = sm(0.025), sm(0.4)
    ncorr = n - (k + 2)
    if ncorr <= 0:
        return []
    s = samples.astype(np.int64) - 128
    cs = np.concatenate(([0], np.cumsum(s)))
    corr = 127 * s[:ncorr] - 128 * (cs[k + 1:k + 1 + ncorr] - cs[1:1 + ncorr]) + 127 * s[k + 1:k + 1 + ncorr]
    peaks = []
    pos, val = -mind, 0
    i = 0
    first = True
    while i < ncorr:
        hi = min(pos + mind, ncorr - 1)
        if i <= hi:
            m = i + int(np.argmax(corr[i:hi + 1]))
            if corr[m] > val:
                pos, val = m, int(corr[m])
                i = m + 1
                continue
            i = hi + 1
            continue
        if not first:
            peaks.append(pos)
        first = False
        pos, val = i, int(corr[i])
        i += 1
    if not first:
        peaks.append(pos)
    return peaks


def find_sync_pulse(freq: np.ndarray, amp: np.ndarray, samples: np.ndarray, sample_rate: int, cfg=SYNC_PULSE) -> dict:
    """data_packet.py:301-342."""
    peaks, heights = find_peaks(amp, height=cfg["height"], prominence=cfg["prominence"])
    f = freq[peaks]
    freq_found = bool(np.all((f >= cfg["fmin"]) & (f <= cfg["fmax"])) and peaks.shape[0] == 1)
    pulses = pattern_search(np.asarray(samples), sample_rate)
    return {"frequency_peak_found": freq_found, "samples_peak_found": bool(len(pulses)),
            "pulse_found": bool(freq_found and len(pulses)), "peaks_fft": [f, heights], "peaks_samples": pulses}
