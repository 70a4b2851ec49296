#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ by RUNNING the reference.

Run in the build container only (the reference does not exist on the GPU box):

    python tests/golden/make_golden.py [--only case1,case2]      (--only: add / refresh these cases, keep the rest of the manifest)

It imports /root/reference/wefax.py (cwd must be the reference root because
config.py:11 opens ``config/config.json`` relative to cwd), turns the
reference's ``time.sleep`` calls into no-ops, runs ``Demodulator.process()`` on
small inputs and stores, per case, the input wav and what the reference
produced at every stage boundary of SURVEY.md section 8a:

  audio (after merge/resample/notch), demodulated envelope, (low, high),
  digitalized uint8 stream, sync peaks, phasing_signals, start_frame,
  the final image, the websocket_stack message sequence, or the exception.

Only data is written (inputs and outputs); no reference source or bytecode.
Float stages are stored as a strided float64 subsample plus a SHA-256 of the
full array to keep the fixtures small.
"""
from __future__ import annotations

import hashlib
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path.insert(0, REPO)

from wefax_amd import synth  # noqa: E402
sys.path.insert(0, HERE)
import recipes  # noqa: E402

SUB = 25  # float stages: keep every SUB-th sample (the SHA-256 of the whole array is in the manifest)


def _import_reference():
    os.environ.setdefault("MPLBACKEND", "Agg")
    sys.dont_write_bytecode = True
    os.chdir(REF)
    sys.path.insert(0, REF)
    import wefax  # type: ignore
    wefax.time.sleep = lambda s: None
    return wefax


def _sha(a: np.ndarray) -> str:
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


class _PeakTap:
    """Grabs the local ``peaks`` list of wefax.py:291 when the function returns."""

    def __init__(self):
        self.peaks = None

    def __call__(self, frame, event, arg):
        if event == "return" and frame.f_code.co_name == "__find_sync_pulse":
            pk = frame.f_locals.get("peaks")
            if pk is not None:
                self.peaks = list(pk)


def run_case(wefax, name: str, wav_path: str, lpm: int) -> dict:
    out: dict = {"name": name, "lpm": lpm}
    d = wefax.Demodulator(wav_path, lines_per_minute=lpm, quiet=True, tcp_stream=True)
    out["file_info"] = {k: (v if isinstance(v, str) else float(v))
                        for k, v in d.file_info().items()}
    tap = _PeakTap()
    sys.setprofile(tap)
    try:
        d.process()
        exc = None
    except Exception as e:  # the reference raises ValueError when no group closes
        exc = e
    finally:
        sys.setprofile(None)
        sys.stdout = sys.__stdout__
    out["exception"] = None if exc is None else [type(exc).__name__, str(exc)]
    arrays = {}

    def put_float(key, a):
        a = np.asarray(a, dtype=np.float64)
        arrays[key + "_sub"] = a[::SUB].copy()
        out[key + "_sha256"] = _sha(a)
        out[key + "_len"] = int(a.shape[0])
        out["float_stride"] = SUB

    if hasattr(d, "audio_data"):
        put_float("audio", d.audio_data)            # after merge/resample/notch
        out["sample_rate"] = int(d.sample_rate)
        out["length"] = float(d.length)
    if hasattr(d, "demodulated_data"):
        put_float("demod", d.demodulated_data)
        lo, hi = np.percentile(d.demodulated_data, (0.5, 99.5))
        out["low"], out["high"] = float(lo), float(hi)
    if hasattr(d, "digitalized_data"):
        arrays["digitalized"] = np.asarray(d.digitalized_data, dtype=np.uint8)
    if tap.peaks is not None:
        arrays["peaks"] = np.asarray(tap.peaks, dtype=np.int64)
    if hasattr(d, "phasing_signals"):
        arrays["phasing_signals"] = np.asarray(d.phasing_signals, dtype=np.int64)
        out["start_frame"] = int(d.start_frame)
    if hasattr(d, "output_image"):
        img = np.asarray(d.output_image)
        arrays["image"] = img
        out["image_size"] = list(d.output_image.size)
        out["image_mode"] = d.output_image.mode
    out["websocket_stack"] = [
        [m.get("data_type"), m.get("progress_title", m.get("message_content")),
         None if "percentage" not in m else float(m["percentage"])]
        for m in d.websocket_stack]
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **arrays)
    return out


def _stage_only(wefax, name: str, wav_path: str, lpm: int = 120) -> dict:
    """Stage goldens for clips on which process() raises (1-second clips)."""
    return run_case(wefax, name, wav_path, lpm)


def main():
    only = None
    if "--only" in sys.argv:
        only = set(sys.argv[sys.argv.index("--only") + 1].split(","))
    wefax = _import_reference()
    inputs = os.path.join(HERE, "inputs")
    os.makedirs(inputs, exist_ok=True)
    manifest = {"reference": "wojlin/WEFAX wefax.py Demodulator.process()",
                "versions": {}, "cases": []}
    old_cases = []
    if only is not None:
        old_cases = [c for c in json.load(open(os.path.join(HERE, "manifest.json")))["cases"] if c["name"] not in only]
    import scipy, PIL  # noqa: E401
    manifest["versions"] = {"numpy": np.__version__, "scipy": scipy.__version__,
                            "Pillow": PIL.__version__,
                            "python": sys.version.split()[0]}

    short = dict(start_tone_s=1.0, stop_tone_s=1.0, black_tail_s=1.0)

    def emit(name, fs, data, lpm, recipe=None):
        if only is not None and name not in only:
            return
        p = os.path.join(inputs, name + ".wav")
        synth.write_wav(p, fs, data)
        print("case", name, data.shape, flush=True)
        c = run_case(wefax, name, p, lpm)
        c["input"] = "inputs/" + name + ".wav"
        if recipe:      # not kept: regenerated from tests/golden/recipes.py wherever it is missing (.gitignore lists the directory's wavs)
            c["recipe"] = recipe
            c["input_sha256"] = recipes.file_sha256(p)
        manifest["cases"].append(c)

    # The inputs are made by tests/golden/recipes.py (deterministic) and are not kept in the repository: the manifest records each
    # wav's SHA-256 as the reference saw it, and a missing input is regenerated and checked against it wherever the tests run.
    #  1 clean mono 11 025 Hz 120 LPM (even N)          2 noisy mono (non-empty phasing group, odd N)       3 240 LPM noisy, IOC 288
    #  4 heavy noise + leading silence, prime-ish N      5 48 kHz mono (resample path)                       6 48 kHz stereo (merge + resample)
    #  7 11 025 Hz stereo that overflows int16 in the merge (wefax.py:372)                                   8 8 kHz mono (upsampling)
    #  9 resample-path captures long enough that a phasing group closes: an IMAGE downstream of scipy.signal.resample (wefax.py:384)
    #    and of the stereo merge + resample (240 LPM keeps them short)
    # 10 sample formats other than int16: scipy.io.wavfile returns uint8 / int32 / float32 arrays and filtfilt's odd extension
    #    (wefax.py:72) is evaluated in THAT dtype (wrapping for the integers, rounded to float32 for float32)
    # 10b the same formats in STEREO: the merge loop (wefax.py:360-373) adds two numpy scalars of the file's dtype -- uint8 wraps
    #    modulo 256, int32 modulo 2**32, float32 stays float32 (and so does the list filtfilt later extends at its ends)
    for rname in ("mono_clean_120", "mono_noisy_120", "mono_noisy_240", "mono_noise20_lead", "mono48k_noisy_120", "stereo48k_120",
                  "stereo_overflow_120", "mono8k_noisy_120", "mono48k_image_240", "stereo48k_image_240", "mono_u8_240", "mono_f32_240",
                  "mono_i32_240", "stereo_u8_240", "stereo_i32_240", "stereo_f32_240",
                  # 10d (round 6) wav shapes the reader accepts and nothing pinned: 24-bit PCM (int32, left-justified, in scipy's hands), float64,
                  #     three channels (wefax.py:365 merges the first two and ignores the rest)
                  "mono_i24_240", "mono_f64_240", "three_ch_240"):
        if only is None or rname in only:
            fs_r, data_r, lpm_r = recipes.RECIPES[rname]()
            emit(rname, fs_r, data_r, lpm_r, recipe=rname)

    # 10b'. a float32 wav that is RESAMPLED: scipy.signal.resample keeps single precision for float32 input (complex64 transforms,
    #       float32 result, and filtfilt then extends THAT array) -- the oracle and the device compute in float64, which is more
    #       accurate and therefore not identical: the case is marked inexact, the tests assert the documented delta (<= 1 grey level
    #       on <= 0.1 % of the stream) instead of equality
    if only is None or "mono48k_f32_240" in only:
        fs_r, data_r, lpm_r = recipes.RECIPES["mono48k_f32_240"]()
        emit("mono48k_f32_240", fs_r, data_r, lpm_r, recipe="mono48k_f32_240")
        manifest["cases"][-1]["oracle_exact"] = False

    # 10c. BASELINE configs[3]'s own format through the REFERENCE: a two-channel int16 stream at 1.536 MS/s (merge of wefax.py:360-373,
    #      FFT resample by 147 / 20480 of wefax.py:384) and a 192 kHz stereo capture.  Inputs come from tests/golden/recipes.py and
    #      are not stored (12 MB / 4.6 MB); a clip this short closes no phasing group: the ValueError of wefax.py:294 is the golden
    for rname in ("iq1536k_2s_240", "stereo192k_6s_240"):
        if only is None or rname in only:
            fs_r, data_r, lpm_r = recipes.RECIPES[rname]()
            emit(rname, fs_r, data_r, lpm_r, recipe=rname)

    # 11. the reference's own 1-second clips (MIT licence, LICENSE:1-3)
    import shutil
    for clip in (() if only is not None else ("image", "stop_tone", "start_tone", "start_tone_noisy",
                 "start_tone_start")):
        src = os.path.join(REF, "test_files", "parts", clip + ".wav")
        dst = os.path.join(inputs, "ref_" + clip + ".wav")
        shutil.copyfile(src, dst)
        os.chmod(dst, 0o644)
        print("clip", clip, flush=True)
        c = _stage_only(wefax, "ref_" + clip, dst)
        c["input"] = "inputs/ref_" + clip + ".wav"
        manifest["cases"].append(c)

    if only is not None:
        manifest["cases"] = old_cases + manifest["cases"]
    with open(os.path.join(HERE, "manifest.json"), "w") as fh:
        json.dump(manifest, fh, indent=1)
    print("wrote", len(manifest["cases"]), "cases")
    recipes.compact_images(HERE)          # pixels -> SHA-256 wherever the oracle's rebuild of the image reproduces the reference's exactly


if __name__ == "__main__":
    main()
