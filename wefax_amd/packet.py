"""Live path, one audio packet: drop-in for the decode half of the reference's ``DataPacket``
(/root/reference/data_packet.py:18-66 constructor, :408-464 ``__process_samples``).

``wefax_live.py:204-209`` builds one ``DataPacket`` per second of sound-card audio and appends its ``samples``
(ints 0..255) to the line buffer; this class computes the same array on the GPU through the C ABI
(``wfx_packet_process``: notch filtfilt at the packet's own rate -> |hilbert| -> medfilt 3 -> per-packet
percentiles -> rint with the 1e-6 guard), and the detectors the live decoder's state machine asks
(``contain_start_tone``, ``contain_stop_tone``, ``find_sync_pulse``: data_packet.py:301-406) from the packet's amplitude
spectrum (``wfx_packet_spectrum``) with the peak conditions evaluated on the host (``wefax_amd/detect.py``).  The
matplotlib charts of the reference class (data_packet.py:67-299) are debugging code and are not provided.
"""
from __future__ import annotations

import numpy as np

from . import _native as nat
from . import detect
from . import hostparams as hp


class DataPacket:
    def __init__(self, sample_rate, samples, lines_per_minute, directory, duration, number, ctx: nat.Context | None = None,
                 notch=None):
        # the constants the reference's constructor reads from config/config.json (data_packet.py:21-42); the shipped values
        # when there is no such file in the working directory
        if notch is None:
            notch = hp.load_notch_settings()
        tones, pulse = hp.load_detector_settings()
        self._tones = tones if tones is not None else detect.TONES
        self._pulse = pulse if pulse is not None else detect.SYNC_PULSE
        self.lines_per_minute = lines_per_minute
        self.duration = duration
        self.number = number
        self.directory = directory
        self.sample_rate = sample_rate
        self.raw_samples = samples
        self._own_ctx = ctx is None
        self._ctx = ctx if ctx is not None else nat.Context(None)
        self._spec = None
        try:
            out, self.low, self.high = _process(self._ctx, sample_rate, np.asarray(samples), notch)
            if self._own_ctx:
                self._spectrum()                          # the private context goes away below
        finally:
            if self._own_ctx:
                self._ctx.close()
                self._ctx = None
        self.samples = out.astype(int)                    # data_packet.py:464 digitalized.astype(int)

    # ---- detectors (data_packet.py:301-406) ----
    def _spectrum(self):
        """(frequencies, normalised amplitude) of data_packet.py:388-406, the FFT on the GPU."""
        if self._spec is None:
            raw = np.asarray(self.raw_samples)
            amp = self._ctx.packet_spectrum(raw)
            self._spec = (detect.frequencies(raw.shape[0], self.sample_rate), detect.normalise(amp))
        return self._spec

    def contain_start_tone(self) -> bool:
        f, a = self._spectrum()
        return detect.contain_tone(f, a, self._tones["start_distance"], self._tones)

    def contain_stop_tone(self) -> bool:
        f, a = self._spectrum()
        return detect.contain_tone(f, a, self._tones["stop_distance"], self._tones)

    def find_sync_pulse(self) -> dict:
        f, a = self._spectrum()
        return detect.find_sync_pulse(f, a, np.asarray(self.samples), self.sample_rate, self._pulse)

    def __repr__(self):
        return (f"data packet {self.number} info: {self.number * self.duration}s-{self.number * self.duration + self.duration}s "
                f"packet len: {len(self.samples)}  sample rate:{self.sample_rate}")


def _process(ctx: nat.Context, sample_rate: int, x: np.ndarray, notch=hp.DEFAULT_NOTCH):
    n = int(x.shape[0])
    if n <= 9:
        raise ValueError("The length of the input vector x must be greater than padlen, which is 9.")
    b, a = hp.iirnotch(int(notch[0]), notch[1], sample_rate)        # data_packet.py:430-432
    lo0, lo1, glo = hp.percentile_plan(n, 0.5)                      # data_packet.py:457
    hi0, hi1, ghi = hp.percentile_plan(n, 99.5)
    return ctx.packet_process(x, b, a, (lo0, lo1, hi0, hi1), glo, ghi)


def process_packets(ctx: nat.Context, sample_rate: int, packets, notch=hp.DEFAULT_NOTCH):
    """uint8 samples of every packet of an iterable / 2-D array, through one context.  Packets of one length (the
    usual case: a recorded stream cut into seconds) are decoded back to back on the device with one upload and one
    download (``wfx_packets_process``)."""
    arrs = [np.asarray(p) for p in packets]
    if len(arrs) > 1 and len({(a.shape, a.dtype) for a in arrs}) == 1 and arrs[0].ndim == 1 and arrs[0].shape[0] > 9:
        n = arrs[0].shape[0]
        b, a = hp.iirnotch(int(notch[0]), notch[1], sample_rate)
        lo0, lo1, glo = hp.percentile_plan(n, 0.5)
        hi0, hi1, ghi = hp.percentile_plan(n, 99.5)
        out, _, _ = ctx.packets_process(np.stack(arrs), b, a, (lo0, lo1, hi0, hi1), glo, ghi)
        return [out[i] for i in range(out.shape[0])]
    return [_process(ctx, sample_rate, a, notch)[0] for a in arrs]


def frames_to_image(ctx: nat.Context, data_points, sample_rate: int, time_for_one_frame: float, frames: int) -> np.ndarray:
    """The strip the live decoder renders every ``minimum_frames_per_update`` lines (wefax_live.py:124-148): the first
    ``frames`` lines of ``data_points`` (digitised samples, 0..255) as pixels 255 - value, ``int(T * rate)`` per line,
    enlarged 4x vertically by Pillow's bicubic filter -- ``wfx_lines_to_image`` with no start offset.
    Returns the uint8 array [4 * frames, width]; ``PIL.Image.fromarray(., "L")`` is the reference's ``img``."""
    w = int(time_for_one_frame * sample_rate)
    pts = np.asarray(data_points)
    if frames < 1 or pts.shape[0] < w * frames:
        raise ValueError("not enough data points for the requested number of lines")
    return ctx.lines_to_image(pts[:w * frames].astype(np.uint8), 0, w)
