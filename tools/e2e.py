#!/usr/bin/env python3
"""End to end, file to file: <in.wav> -> <out.png>, the HIP path next to the oracle + PIL
(what `python wefax.py in.wav 120 out.png` costs in the reference, minus its sleeps).

    python tools/e2e.py [--minutes 10]
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
from wefax_amd import Demodulator, synth     # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--minutes", type=float, default=10.0)
    ap.add_argument("--no-oracle", action="store_true")
    a = ap.parse_args()
    lines = int((a.minutes * 60 - 50) / 0.5)
    x = synth.synth_capture(11025.0, noise=0.05, seed=0, image_lines=lines)
    out = {"samples": int(x.shape[0])}
    with tempfile.TemporaryDirectory() as td:
        wav, png, png2 = os.path.join(td, "in.wav"), os.path.join(td, "out.png"), os.path.join(td, "ref.png")
        synth.write_wav(wav, 11025, x)
        d = Demodulator(wav, 120, quiet=True, tcp_stream=False)
        d.process()
        d.save_output_image(png)                 # warm-up: context, plans, allocations
        d.close()                                # the previous file's Demodulator is done: its context goes back to the idle pool
        t = {}
        t0 = time.perf_counter()
        d = Demodulator(wav, 120, quiet=True, tcp_stream=False, device=0)
        # (one Demodulator per file like the reference; its context comes from the idle pool of wefax_amd.wefax)
        d.process()
        t["process_s"] = time.perf_counter() - t0
        t1 = time.perf_counter()
        d.save_output_image(png.replace("out.png", "out2.png"))        # a new file, like every file of a service
        t["save_png_s"] = time.perf_counter() - t1
        t["total_s"] = time.perf_counter() - t0
        out["hip"] = {k: round(v, 4) for k, v in t.items()}
        out["hip"]["png_bytes"] = os.path.getsize(png.replace("out.png", "out2.png"))
        out["hip"]["msamples_per_s_file_to_file"] = round(x.shape[0] / t["total_s"] / 1e6, 2)
        if not a.no_oracle:
            from PIL import Image
            from oracle import wefax_oracle as wo
            t0 = time.perf_counter()
            ref = wo.process(wav, 120, want_messages=False)
            tp = time.perf_counter() - t0
            t1 = time.perf_counter()
            Image.fromarray(ref["image"], "L").save(png2)
            ts = time.perf_counter() - t1
            out["oracle_plus_pil"] = {"process_s": round(tp, 3), "save_png_s": round(ts, 3), "total_s": round(tp + ts, 3),
                                      "msamples_per_s_file_to_file": round(x.shape[0] / (tp + ts) / 1e6, 3)}
            out["png_pixels_equal"] = bool(np.array_equal(np.asarray(Image.open(png.replace("out.png", "out2.png"))), np.asarray(Image.open(png2))))
    print(json.dumps(out))


if __name__ == "__main__":
    main()
