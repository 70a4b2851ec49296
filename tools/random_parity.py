#!/usr/bin/env python3
"""Randomised whole-path parity sweep on the GPU box: random rate / length / LPM / noise / sample format / channels, the
drop-in Demodulator against the oracle -- same exception or same start_frame, identical uint8 stream, image max |delta| <= 1.

    python tools/random_parity.py [--cases 40] [--seed 0]
"""
import argparse
import json
import os
import sys
import tempfile

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from wefax_amd import Demodulator, synth      # noqa: E402
from oracle import wefax_oracle as wo          # noqa: E402


def make_case(rng):
    fs = int(rng.choice([11025, 11025, 8000, 22050, 44100, 48000, 12000, 16000]))
    lpm = int(rng.choice([60, 90, 120, 120, 240]))
    t_line = 60.0 / lpm
    phasing = int(rng.integers(12, 45))
    lines = int(rng.integers(4, 40))
    noise = float(rng.choice([0.0, 0.01, 0.05, 0.1]))
    kw = dict(lpm=lpm, start_tone_s=float(rng.uniform(0.2, 1.5)), phasing_lines=phasing, image_lines=lines,
              stop_tone_s=float(rng.uniform(0.2, 1.0)), black_tail_s=float(rng.uniform(0.1, 1.0)))
    x = synth.synth_capture(float(fs), noise=noise, seed=int(rng.integers(1 << 30)), **kw)
    x = x[:x.shape[0] - int(rng.integers(0, 1000))]               # lengths with awkward factors
    fmt = str(rng.choice(["i16", "i16", "i16", "u8", "f32", "i32", "stereo", "stereo_wrap"]))
    if fmt == "u8":
        data = ((x.astype(np.int32) >> 8) + 128).astype(np.uint8)
    elif fmt == "f32":
        data = (x / 32768.0).astype(np.float32)
    elif fmt == "i32":
        data = x.astype(np.int32) << 16
    elif fmt == "stereo":
        data = np.stack([x, (x * 0.5).astype(np.int16)], axis=1)
    elif fmt == "stereo_wrap":
        data = np.stack([x, x], axis=1)                                # L + R overflows int16: the reference wraps
    else:
        data = x
    return dict(fs=fs, lpm=lpm, noise=noise, fmt=fmt, n=int(x.shape[0]), t_line=t_line), data


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=40)
    ap.add_argument("--seed", type=int, default=0)
    a = ap.parse_args()
    rng = np.random.default_rng(a.seed)
    bad = 0
    with tempfile.TemporaryDirectory() as td:
        for k in range(a.cases):
            meta, data = make_case(rng)
            path = os.path.join(td, f"c{k}.wav")
            synth.write_wav(path, meta["fs"], data)
            ref, ref_exc = {}, None
            try:
                ref = wo.process(path, meta["lpm"], want_messages=True)
                if ref.get("exception") is not None:
                    ref_exc = type(ref["exception"]).__name__
            except Exception as e:      # noqa: BLE001
                ref_exc = type(e).__name__
            d = Demodulator(path, meta["lpm"], quiet=True)
            exc = None
            try:
                d.process()
            except Exception as e:      # noqa: BLE001
                exc = type(e).__name__
            rec = dict(meta)
            if ref_exc is not None or exc is not None:
                rec.update(exception=exc, oracle_exception=ref_exc, ok=exc == ref_exc)
            else:
                st = np.asarray(d.digitalized_data, dtype=np.uint8)
                img = d.output_array
                di = np.abs(img.astype(np.int16) - ref["image"].astype(np.int16)) if img.shape == ref["image"].shape else np.array([255])
                got = [(m.get("progress_title"), m.get("percentage")) if m["data_type"] == "progress_bar" else ("message", m["message_content"])
                       for m in d.websocket_stack]
                want = [(m[1], m[2]) if m[0] == "progress_bar" else ("message", m[1]) for m in ref["messages"]]
                rec.update(start_eq=bool(d.start_frame == ref["start_frame"]), stream_ne=int(np.count_nonzero(st != ref["digitalized"])),
                           image_max=int(di.max()) if di.size else 0, shape_eq=bool(img.shape == ref["image"].shape), messages_eq=bool(got == want))
                rec["ok"] = rec["start_eq"] and rec["stream_ne"] == 0 and rec["image_max"] <= 1 and rec["shape_eq"] and rec["messages_eq"]
            d.close()
            bad += 0 if rec["ok"] else 1
            print(json.dumps(rec), flush=True)
    print(json.dumps({"cases": a.cases, "failed": bad}))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
