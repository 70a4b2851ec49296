#!/usr/bin/env python3
"""Sharded exact decode vs the single-GPU exact path and the oracle, every rank emulated on ONE GPU
(local communicator).  Prints one JSON line per case.

    python tools/shard_check.py [--cases plain,resample,stereo,c2] [--worlds 1,2,3,8]
"""
import argparse
import json
import os
import sys
import tempfile
import time

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)

from wefax_amd import _native as nat          # noqa: E402
from wefax_amd import sharded, synth           # noqa: E402
from wefax_amd.wefax import DecodeJob          # noqa: E402


def single(x, sr, lpm=120):
    ctx = nat.Context(0)
    job = DecodeJob(ctx, x, sr, lpm)
    job.run()
    info = job.result()
    out = {"stream": job.fetch("digitalized"), "env": job.fetch("envelope"), "audio": job.fetch("audio"), "low": info.low, "high": info.high,
           "start": int(info.start_frame), "no_group": int(info.no_group), "peaks": [int(info.peak_pos[k]) for k in range(info.npeaks)]}
    if not info.no_group and info.height > 0:
        out["image"] = job.fetch("image")
    ctx.close()
    return out


def oracle(x, sr, lpm=120):
    from oracle import wefax_oracle as wo
    with tempfile.TemporaryDirectory() as td:
        p = os.path.join(td, "c.wav")
        synth.write_wav(p, sr, x)
        return wo.process(p, lpm, want_messages=False)


def check(name, x, sr, worlds, lpm=120, with_oracle=True):
    ref = single(x, sr, lpm)
    orc = oracle(x, sr, lpm) if with_oracle else None
    first = None
    for w in worlds:
        t0 = time.time()
        try:
            r = sharded.decode_emulated(x, sr, w, lpm)
        except Exception as e:
            print(json.dumps({"case": name, "world": w, "error": str(e)[:300]}), flush=True)
            continue
        rec = {"case": name, "world": w, "n": int(r["n"]), "first_radix": list(r["first_radix"]), "secs": round(time.time() - t0, 2)}
        rec["stream_vs_single"] = int(np.count_nonzero(r["digitalized"] != ref["stream"]))
        rec["blocks_eq_stream"] = bool(np.array_equal(r["digitalized"], r["digitalized_blocks"]))
        rec["env_rel_vs_single"] = float(np.max(np.abs(r["envelope"] - ref["env"])) / np.max(np.abs(ref["env"])))
        rec["audio_rel_vs_single"] = float(np.max(np.abs(r["audio"] - ref["audio"])) / max(1e-300, np.max(np.abs(ref["audio"]))))
        rec["low_high_eq_single"] = bool(r["low"] == ref["low"] and r["high"] == ref["high"])
        rec["ranks_agree"] = bool(len(set(r["lows"])) == 1 and len(set(r["highs"])) == 1)
        rec["start"] = r["sync"]["start_frame"]
        rec["start_eq_single"] = bool(r["sync"]["start_frame"] == ref["start"] and r["sync"]["peaks"] == ref["peaks"])
        if "image" in r and "image" in ref:
            rec["image_maxdiff_single"] = int(np.max(np.abs(r["image"].astype(np.int16) - ref["image"].astype(np.int16)))) if r["image"].shape == ref["image"].shape else -1
        if orc is not None:
            rec["stream_vs_oracle"] = int(np.count_nonzero(r["digitalized"] != orc["digitalized"]))
            rec["start_eq_oracle"] = bool(r["sync"]["start_frame"] == orc.get("start_frame"))
            if "image" in r and "image" in orc:
                rec["image_maxdiff_oracle"] = int(np.max(np.abs(r["image"].astype(np.int16) - orc["image"].astype(np.int16)))) if r["image"].shape == orc["image"].shape else -1
        if first is None:
            first = r
        else:
            rec["bit_identical_to_first_world"] = bool(np.array_equal(r["envelope"], first["envelope"]) and np.array_equal(r["digitalized"], first["digitalized"])
                                                       and r["low"] == first["low"] and r["high"] == first["high"]
                                                       and ("image" not in r or np.array_equal(r["image"], first["image"])))
        print(json.dumps(rec), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", default="plain,resample,stereo")
    ap.add_argument("--worlds", default="1,2,3,8")
    args = ap.parse_args()
    worlds = [int(v) for v in args.worlds.split(",")]
    cases = args.cases.split(",")
    kw130 = dict(start_tone_s=5.0, phasing_lines=20, image_lines=220, stop_tone_s=2.0, black_tail_s=3.0)     # 130 s
    if "plain" in cases:
        x = synth.synth_capture(11025.0, noise=0.05, seed=3, **kw130)
        check("plain_130s_11025", x, 11025, worlds)
    if "resample" in cases:
        kw30 = dict(start_tone_s=2.0, phasing_lines=20, image_lines=30, stop_tone_s=1.0, black_tail_s=2.0)    # 30 s
        x = synth.synth_capture(48000.0, noise=0.05, seed=4, **kw30)
        check("resample_30s_48000", x, 48000, worlds)
    if "stereo" in cases:
        kw30 = dict(start_tone_s=2.0, phasing_lines=20, image_lines=30, stop_tone_s=1.0, black_tail_s=2.0)
        x = synth.synth_capture(48000.0, noise=0.05, seed=5, iq=True, **kw30)
        check("stereo_30s_48000", x, 48000, [w for w in worlds if w <= 3])
    if "c2" in cases:
        x = synth.config_c2(noise=0.05, seed=0)
        check("c2_10min", x, 11025, worlds)
    if "lpm240" in cases:
        x = synth.synth_capture(11025.0, noise=0.05, seed=6, lpm=240, start_tone_s=5.0, phasing_lines=40, image_lines=440, stop_tone_s=2.0, black_tail_s=3.0)
        check("plain_130s_240lpm", x, 11025, worlds, lpm=240)


if __name__ == "__main__":
    main()
