"""Inputs of golden cases that are too large to keep in the repository: they are REGENERATED from these recipes
(deterministic: ``wefax_amd.synth`` + NumPy's PCG64 streams) wherever the file is missing, and checked against the SHA-256
that ``make_golden.py`` recorded in the manifest when the reference ran on them.  Data generators only -- nothing of the
reference is here.

The two cases pin the oracle in the format of BASELINE configs[3] itself (round-3 verdict): a two-channel
int16 stream at 1.536 MS/s through the reference's merge (wefax.py:360-373) and its FFT resample by 147 / 20480
(wefax.py:384), and a 192 kHz stereo capture (147 / 2560)."""
from __future__ import annotations

import hashlib
import os

import numpy as np


def _iq1536k_2s_240():
    from wefax_amd import synth
    # 0.3 s start tone + 6 lines of 0.25 s + 0.2 s of tail = 2.0 s = 3 072 000 IQ frames
    return 1536000, synth.synth_capture(1536000.0, noise=0.05, seed=51, lpm=240, iq=True, start_tone_s=0.3, phasing_lines=4, image_lines=2,
                                        stop_tone_s=0.1, black_tail_s=0.1), 240


def _stereo192k_6s_240():
    from wefax_amd import synth
    # 0.5 s start tone + 20 lines of 0.25 s + 0.5 s of tail = 6.0 s = 1 152 000 frames
    return 192000, synth.synth_capture(192000.0, noise=0.05, seed=52, lpm=240, iq=True, start_tone_s=0.5, phasing_lines=14, image_lines=6,
                                       stop_tone_s=0.25, black_tail_s=0.25), 240


def _mono48k_f32_240():
    from wefax_amd import synth
    # a float32 wav that needs resampling (3.9 MB): the one place where the reference's arithmetic is single precision
    b48 = synth.synth_capture(48000.0, noise=0.05, seed=44, lpm=240, phasing_lines=40, image_lines=30, start_tone_s=1.0, stop_tone_s=1.0,
                              black_tail_s=1.0)
    return 48000, b48.astype(np.float32) / np.float32(32768.0), 240


RECIPES = {"iq1536k_2s_240": _iq1536k_2s_240, "stereo192k_6s_240": _stereo192k_6s_240, "mono48k_f32_240": _mono48k_f32_240}


def file_sha256(path: str) -> str:
    h = hashlib.sha256()
    with open(path, "rb") as fh:
        for blk in iter(lambda: fh.read(1 << 20), b""):
            h.update(blk)
    return h.hexdigest()


def ensure_input(golden_dir: str, case: dict) -> str:
    """Path of the case's input wav; written from its recipe when it is not there (and verified against the manifest)."""
    path = os.path.join(golden_dir, case["input"])
    if os.path.exists(path) or "recipe" not in case:
        return path
    from wefax_amd import synth
    fs, data, _ = RECIPES[case["recipe"]]()
    tmp = path + ".tmp%d" % os.getpid()
    synth.write_wav(tmp, fs, data)
    got = file_sha256(tmp)
    if case.get("input_sha256") and got != case["input_sha256"]:
        os.remove(tmp)
        raise RuntimeError(f"golden input {case['name']}: regenerated wav hashes to {got}, the manifest says {case['input_sha256']} "
                           "(a different NumPy random stream?)")
    os.replace(tmp, path)
    return path
