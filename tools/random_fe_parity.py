#!/usr/bin/env python3
"""Randomised sweep of the oversampled-capture path (time-domain front end to the hand-over rate + exact path) against the
oracle: rate, length, LPM, noise and seed at random; prints per case how many uint8 stream bytes differ (by how much) and the
largest pixel difference -- the front end is fp32 and band-limited by filters, so this path is within +-1, not bit-identical.

    python tools/random_fe_parity.py [--cases 12] [--seed 0]
"""
import argparse
import json
import os
import sys
import tempfile

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from wefax_amd import _native as nat, polyphase as pp, sharded, synth      # noqa: E402
from oracle import wefax_oracle as wo                                       # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=12)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--fs", type=int, default=0, help="every clip at this rate (default: 1.536 MS/s, 192 kHz and 48 kHz mixed)")
    ap.add_argument("--ugly", action="store_true", help="a short-wave channel: selective fading (0-100 %%), +-50 Hz carrier drift, impulsive noise, clipping, DC offset (synth.synth_capture ugly=...)")
    ap.add_argument("--out", default=None, help="also append the per-case records and the summary to this file (JSON lines)")
    a = ap.parse_args()
    sink = open(a.out, "a") if a.out else None

    def emit(rec):
        line = json.dumps(rec)
        print(line, flush=True)
        if sink:
            sink.write(line + "\n")
            sink.flush()
    rng = np.random.default_rng(a.seed)
    ctx = nat.Context(0)
    worst = {"stream_max": 0, "image_max": 0, "image_gt1": 0, "start_ne": 0, "not_ok": 0, "stream_bytes": 0, "stream_ne": 0, "image_bytes": 0, "image_ne": 0}
    hist_stream, hist_image = {}, {}                                   # differing bytes per clip -> clips
    with tempfile.TemporaryDirectory() as td:
        for k in range(a.cases):
            fs = int(rng.choice([1536000, 1536000, 192000, 48000]))
            if a.fs:
                fs = a.fs
            iq = fs != 48000
            lpm = int(rng.choice([120, 240]))
            seconds = int(rng.integers(24, 41))
            t_line = 60.0 / lpm
            phasing = 40 if lpm == 240 else 20
            lines = int(round((seconds - 3.0) / t_line)) - phasing
            ugly = None
            if a.ugly:
                ugly = dict(fade_depth=float(rng.choice([0.3, 0.7, 1.0])), fade_hz=float(rng.uniform(0.05, 0.5)), drift_hz=float(rng.uniform(-50, 50)),
                            impulses_per_s=float(rng.choice([0.0, 2.0, 10.0])), impulse_fs=float(rng.uniform(0.3, 1.0)), clip=float(rng.choice([1.0, 1.6, 2.5])),
                            dc=float(rng.uniform(-0.1, 0.1)))
            x = synth.synth_capture(float(fs), noise=float(rng.choice([0.01, 0.05, 0.1])), seed=int(rng.integers(1 << 30)), lpm=lpm, phasing_lines=phasing,
                                    image_lines=lines, start_tone_s=1.0, stop_tone_s=1.0, black_tail_s=1.0, iq=iq, ugly=ugly)
            fe = pp.FrontEnd(fs, stop_rate=pp.FrontEnd.handover_rate(fs))
            if rng.integers(0, 2):       # not a whole number of seconds (still whole hand-over samples): int(11025 * n0 / fs) != n0 * 11025 / fs
                x = np.ascontiguousarray(x[:x.shape[0] - fe.granule() * int(rng.integers(1, 5000))])
            path = os.path.join(td, "c.wav")
            synth.write_wav(path, fs, x)
            ref = wo.process(path, lpm, want_messages=False)
            dec = sharded.FrontEndExactDecoder(ctx, fe, x, lines_per_minute=lpm)
            dec.run()
            info = dec.result()
            rec = dict(ugly=ugly, fs=fs, lpm=lpm, seconds=seconds, frames=int(x.shape[0]), whole_seconds=bool(x.shape[0] % fs == 0), f64_chain=bool(fe.f64),
                       exact_ingest=dec.fe.exact_ingest)
            if ref.get("exception") is not None or info.no_group:
                rec["no_group"] = [ref.get("exception") is not None, bool(info.no_group)]
                rec["ok"] = rec["no_group"][0] == rec["no_group"][1]
            else:
                st, img = dec.fetch("digitalized"), dec.fetch("image")
                d = np.abs(st.astype(np.int16) - ref["digitalized"].astype(np.int16))
                di = np.abs(img.astype(np.int16) - ref["image"].astype(np.int16)) if img.shape == ref["image"].shape else np.array([255])
                rec.update(start_eq=bool(info.start_frame == ref["start_frame"]), stream_ne=int(np.count_nonzero(d)), stream_max=int(d.max()),
                           image_ne=int(np.count_nonzero(di)), image_gt1=int(np.count_nonzero(di > 1)), image_max=int(di.max()))
                rec["ok"] = rec["start_eq"] and rec["stream_max"] <= 1 and rec["image_max"] <= 1
                worst["stream_max"] = max(worst["stream_max"], rec["stream_max"])
                worst["image_max"] = max(worst["image_max"], rec["image_max"])
                worst["image_gt1"] += rec["image_gt1"]
                worst["start_ne"] += 0 if rec["start_eq"] else 1
                worst["stream_bytes"] += int(d.size)
                worst["stream_ne"] += rec["stream_ne"]
                worst["image_bytes"] += int(di.size)
                worst["image_ne"] += rec["image_ne"]
                hist_stream[rec["stream_ne"]] = hist_stream.get(rec["stream_ne"], 0) + 1
                hist_image[rec["image_ne"]] = hist_image.get(rec["image_ne"], 0) + 1
            worst["not_ok"] += 0 if rec["ok"] else 1
            dec.close()
            emit(rec)
    emit({"summary": True, "cases": a.cases, "seed": a.seed, **worst,
          "clips_by_differing_stream_bytes": {str(k): v for k, v in sorted(hist_stream.items())},
          "clips_by_differing_image_bytes": {str(k): v for k, v in sorted(hist_image.items())}})
    if worst["not_ok"]:
        raise SystemExit(f"{worst['not_ok']} case(s) outside the bar (start_frame equal, stream and image within 1)")


if __name__ == "__main__":
    main()
