#!/usr/bin/env python3
"""Randomised sweep of the sharded exact decode (every rank emulated on one GPU) against the oracle: at 11 025 Hz ANY
length (two thirds of those cases: arbitrary, even or odd; the rest 13-smooth), smooth lengths at the rates with the distributed resampler, LPM, noise, mono / stereo, world sizes 2..8.  A capture the library refuses
to shard must be refused on every rank alike, with a reason; everything else must give the oracle's uint8 stream, start_frame
and image, identically for every world size.

    python tools/random_shard_parity.py [--cases 24] [--seed 0]
"""
import argparse
import json
import os
import sys
import tempfile

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from wefax_amd import _native as nat, sharded, synth      # noqa: E402
from oracle import wefax_oracle as wo                      # noqa: E402


def smooth_length(rng, lo, hi):
    """An even length in [lo, hi) whose half is 13-smooth (what the transforms need)."""
    while True:
        n = 2
        while n < lo:
            n *= int(rng.choice([2, 2, 3, 3, 5, 5, 7, 11, 13]))
        if n < hi:
            return n


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=24)
    ap.add_argument("--seed", type=int, default=0)
    a = ap.parse_args()
    rng = np.random.default_rng(a.seed)
    bad = 0
    with tempfile.TemporaryDirectory() as td:
        for k in range(a.cases):
            fs = int(rng.choice([11025, 11025, 22050, 44100]))
            lpm = int(rng.choice([120, 240]))
            ratio = fs // 11025
            n_out = smooth_length(rng, 150000, 420000)
            if fs == 11025 and rng.integers(0, 3) > 0:              # the native rate takes ANY length (padded distributed convolution)
                n_out = int(rng.integers(150000, 420000))
            n0 = n_out * ratio                                      # whole ratio: int(11025 * n0 / fs) == n_out
            if fs != 11025 and rng.integers(0, 3) == 0:             # resampled captures of ARBITRARY length: no distributed form -> the single plan
                n0 = int(rng.integers(150000, 420000)) * ratio + int(rng.integers(0, ratio))
                n_out = int(11025 * (n0 / fs))
            kw = dict(lpm=lpm, start_tone_s=0.5, phasing_lines=int(rng.integers(20, 44)), image_lines=400, stop_tone_s=0.5, black_tail_s=0.5)
            x = synth.synth_capture(float(fs), noise=float(rng.choice([0.0, 0.02, 0.05])), seed=int(rng.integers(1 << 30)), **kw)
            if x.shape[0] < n0:
                x = np.concatenate([x, np.zeros(n0 - x.shape[0], np.int16)])
            x = x[:n0]
            stereo = bool(rng.integers(0, 2))
            data = np.stack([x, x // 2], axis=1) if stereo else x
            path = os.path.join(td, f"s{k}.wav")
            synth.write_wav(path, fs, data)
            ref = wo.process(path, lpm, want_messages=False)
            rec = dict(fs=fs, lpm=lpm, n0=int(n0), n=int(n_out), stereo=stereo, oracle_exception=type(ref["exception"]).__name__ if ref.get("exception") is not None else None)
            worlds = sorted(set(int(w) for w in rng.choice([2, 3, 4, 5, 6, 7, 8], size=2, replace=False)))
            first, ok = None, True
            for w in worlds:
                try:
                    r = sharded.decode_emulated(data, fs, w, lpm, want=("image", "stream"))
                except nat.NativeError as e:
                    rec[f"world{w}"] = "refused: " + str(e)[-90:]
                    ok = False                                      # nothing valid is refused any more (single plan)
                    continue
                res = dict(plan={0: "single", 1: "rows", 2: "columns"}[int(r["plan"])], stream_ne=int(np.count_nonzero(r["digitalized"] != ref["digitalized"])),
                           blocks_eq=bool(np.array_equal(r["digitalized"], r["digitalized_blocks"])))
                if ref.get("exception") is not None:
                    res["no_group"] = bool(r["sync"]["no_group"])
                    ok &= res["no_group"]
                else:
                    res["start_eq"] = bool(r["sync"]["start_frame"] == ref["start_frame"])
                    res["image_max"] = int(np.abs(r["image"].astype(np.int16) - ref["image"].astype(np.int16)).max()) if "image" in r and r["image"].shape == ref["image"].shape else 255
                    ok &= res["start_eq"] and res["image_max"] == 0
                ok &= res["stream_ne"] == 0 and res["blocks_eq"]
                if first is None:
                    first = r
                else:
                    res["same_as_first_world"] = bool(np.array_equal(r["digitalized"], first["digitalized"]) and r["low"] == first["low"] and r["high"] == first["high"])
                    ok &= res["same_as_first_world"]
                rec[f"world{w}"] = res
            rec["ok"] = bool(ok)
            bad += 0 if ok else 1
            print(json.dumps(rec), flush=True)
    print(json.dumps({"cases": a.cases, "failed": bad}))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
