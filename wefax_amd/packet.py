"""Live path, one audio packet: drop-in for the decode half of the reference's ``DataPacket``
(/root/reference/data_packet.py:18-66 constructor, :408-464 ``__process_samples``).

``wefax_live.py:204-209`` builds one ``DataPacket`` per second of sound-card audio and appends its ``samples``
(ints 0..255) to the line buffer; this class computes the same array on the GPU through the C ABI
(``wfx_packet_process``: notch filtfilt at the packet's own rate -> |hilbert| -> medfilt 3 -> per-packet
percentiles -> rint with the 1e-6 guard).  The matplotlib charts and the tone / sync-pulse detectors of the
reference class (data_packet.py:67-406) are debugging and control-plane code and are not provided.
"""
from __future__ import annotations

import numpy as np

from . import _native as nat
from . import hostparams as hp


class DataPacket:
    def __init__(self, sample_rate, samples, lines_per_minute, directory, duration, number, ctx: nat.Context | None = None,
                 notch=hp.DEFAULT_NOTCH):
        self.lines_per_minute = lines_per_minute
        self.duration = duration
        self.number = number
        self.directory = directory
        self.sample_rate = sample_rate
        self.raw_samples = samples
        self._own_ctx = ctx is None
        self._ctx = ctx if ctx is not None else nat.Context(0)
        try:
            out, self.low, self.high = _process(self._ctx, sample_rate, np.asarray(samples), notch)
        finally:
            if self._own_ctx:
                self._ctx.close()
        self.samples = out.astype(int)                    # data_packet.py:464 digitalized.astype(int)

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
    """uint8 samples of every packet of an iterable / 2-D array, through one context."""
    return [_process(ctx, sample_rate, np.asarray(p), notch)[0] for p in packets]
