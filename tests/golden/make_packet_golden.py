#!/usr/bin/env python3
"""Golden vectors of the live path's per-packet decode, made by RUNNING the reference.

    python tests/golden/make_packet_golden.py          (build container only)

Imports /root/reference/data_packet.py (cwd = the reference root: config.py opens config/config.json relative to
it), builds ``DataPacket(sample_rate, samples, lpm, directory, duration, number)`` for a handful of one-second
packets cut from the golden wav inputs (11 025 Hz, 48 kHz, 8 kHz) plus synthetic ones (all-zero, constant,
int16-extreme) and stores the packet's input samples and ``DataPacket.samples`` (what data_packet.py:408-464
produces: notch filtfilt -> |hilbert| -> medfilt 3 -> per-packet percentiles -> rint with the 1e-6 guard).
Also stored per packet: the normalised one-sided spectrum and the answers of the reference's detectors
(``contain_start_tone``, ``contain_stop_tone``, ``find_sync_pulse``: data_packet.py:301-406).
Only data is written: no reference source or bytecode.
"""
from __future__ import annotations

import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path.insert(0, REPO)

from wefax_amd import hostparams as hp  # noqa: E402


def main():
    os.environ.setdefault("MPLBACKEND", "Agg")
    sys.dont_write_bytecode = True
    cases = []
    for name, starts in (("mono_noisy_120.wav", (0, 3, 11)), ("mono_noisy_240.wav", (1, 6)), ("mono48k_noisy_120.wav", (0, 5)),
                         ("mono8k_noisy_120.wav", (2,)), ("ref_image.wav", (0,)), ("ref_start_tone.wav", (0,))):
        sr, data = hp.read_wav(os.path.join(HERE, "inputs", name))
        for k in starts:
            seg = np.ascontiguousarray(data[k * sr:(k + 1) * sr])
            if seg.shape[0] == sr:
                cases.append((f"{name[:-4]}_p{k}", sr, seg))
    rng = np.random.default_rng(5)
    cases.append(("noise_int16_extremes", 11025, rng.integers(-32768, 32767, size=11025).astype(np.int16)))
    cases.append(("short_odd_length", 11025, (6000 * np.sin(np.arange(4097) * 0.9)).astype(np.int16)))
    cases.append(("tone_plus_step", 11025, np.concatenate([np.zeros(3000), 9000 * np.sin(np.arange(8025) * 1.1)]).astype(np.int16)))
    # one-second packets of a synthetic 120 LPM transmission (5 s start tone, 30 s phasing, image, 5 s stop tone): what the
    # tone / sync-pulse detectors of the live path (data_packet.py:301-406) are for
    from wefax_amd import synth
    c2 = synth.config_c2(noise=0.02, seed=3)
    for k in (1, 3, 6, 20, 200, 636, 641, 643, 648):
        cases.append((f"c2_p{k}", 11025, np.ascontiguousarray(c2[k * 11025:(k + 1) * 11025])))
    os.chdir(REF)
    sys.path.insert(0, REF)
    import data_packet as dp  # type: ignore
    out = {}
    names = []
    for name, sr, seg in cases:
        pkt = dp.DataPacket(sr, seg, 120, "/tmp/", 1, 0)
        got = np.asarray(pkt.samples)
        assert got.min() >= 0 and got.max() <= 255
        out[name + "__in"] = seg
        out[name + "__sr"] = np.int64(sr)
        out[name + "__out"] = got.astype(np.uint8)
        # detectors (data_packet.py:301-406), run on the reference object itself
        freq, amp = pkt._DataPacket__fourier_transform()
        out[name + "__amp"] = np.asarray(amp, dtype=np.float64)
        out[name + "__freq_last"] = np.float64(freq[-1])
        out[name + "__start_tone"] = np.bool_(pkt.contain_start_tone())
        out[name + "__stop_tone"] = np.bool_(pkt.contain_stop_tone())
        sp = pkt.find_sync_pulse()
        out[name + "__sp_flags"] = np.array([sp["frequency_peak_found"], sp["samples_peak_found"], sp["pulse_found"]], dtype=np.bool_)
        out[name + "__sp_fft_freq"] = np.asarray(sp["peaks_fft"][0], dtype=np.float64)
        out[name + "__sp_fft_height"] = np.asarray(sp["peaks_fft"][1], dtype=np.float64)
        out[name + "__sp_samples"] = np.asarray(sp["peaks_samples"], dtype=np.int64)
        names.append(name)
    out["names"] = np.array(names)
    np.savez_compressed(os.path.join(HERE, "packets.npz"), **out)
    print("wrote", len(names), "packet cases:", ", ".join(names))


if __name__ == "__main__":
    main()
