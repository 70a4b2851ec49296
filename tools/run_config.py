#!/usr/bin/env python3
"""Run one BASELINE.json workload through the HIP path and (optionally) the oracle.

    python tools/run_config.py c3 [--minutes 60] [--oracle] [--steps 3]

c2: 10-min 11 025 Hz mono; c3: N-min 48 kHz mono (resample path); c4: --seconds S of the
1.536 MS/s int16 IQ stream through the reference-faithful path (stereo merge with int16
wrap, exact FFT resample); prints one JSON line with timings, parity and the per-kernel
HIP-event profile.
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
from wefax_amd import _native as nat     # noqa: E402
from wefax_amd import synth              # noqa: E402
from wefax_amd.wefax import DecodeJob    # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("config", choices=["c2", "c3", "c4"])
    ap.add_argument("--minutes", type=float, default=None)
    ap.add_argument("--seconds", type=float, default=30.0)
    ap.add_argument("--oracle", action="store_true")
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--noise", type=float, default=0.05)
    a = ap.parse_args()
    t0 = time.time()
    if a.config == "c2":
        fs, x = 11025, synth.config_c2(noise=a.noise, seed=0)
    elif a.config == "c4":
        lines = int((a.seconds - 14.0) / 0.5)
        fs, x = 1536000, synth.synth_capture(1536000.0, noise=a.noise, seed=0, iq=True, start_tone_s=2.0, phasing_lines=20,
                                             image_lines=lines, stop_tone_s=1.0, black_tail_s=1.0)
    else:
        minutes = a.minutes or 60.0
        lines = int((minutes * 60.0 - 15.0) / 0.5) - 60
        fs, x = 48000, synth.synth_capture(48000.0, noise=a.noise, seed=0, image_lines=lines, black_tail_s=5.0)
    t_synth = time.time() - t0
    ctx = nat.Context(0)
    t0 = time.time()
    job = DecodeJob(ctx, x, fs, 120)
    t_upload = time.time() - t0
    job.run()
    info = job.result()          # warm-up (plans, allocations)
    t0 = time.time()
    for _ in range(a.steps):
        job.run()
    ctx.sync()
    dt = (time.time() - t0) / a.steps
    ctx.profile_reset()
    ctx.profile_enable(True)
    job.run()
    ctx.sync()
    ctx.profile_enable(False)
    prof = {k: round(v[1] * 1e3, 1) for k, v in ctx.profile().items()}     # us per step
    out = {"config": a.config, "n0": int(x.shape[0]), "n": int(job.n), "sample_rate": fs, "synth_s": round(t_synth, 1),
           "upload_s": round(t_upload, 3), "ms_per_step": round(dt * 1e3, 3),
           "msamples_per_s": round(x.shape[0] / dt / 1e6, 1), "image": [info.width, 4 * info.height],
           "start_frame": int(info.start_frame), "npeaks": info.npeaks, "no_group": info.no_group,
           "kernels_us": prof}
    if a.oracle:
        from oracle import wefax_oracle as wo
        with tempfile.TemporaryDirectory() as td:
            p = os.path.join(td, "x.wav")
            synth.write_wav(p, fs, x)
            t0 = time.time()
            ref = wo.process(p, 120, want_messages=False)
            out["oracle_s"] = round(time.time() - t0, 2)
        dig = job.fetch("digitalized")
        out["digitalized_mismatches"] = int(np.count_nonzero(dig != ref["digitalized"]))
        out["start_frame_equal"] = bool(ref.get("start_frame") == info.start_frame)
        if "image" in ref:
            img = job.fetch("image")
            out["max_abs_pixel_delta"] = int(np.max(np.abs(img.astype(np.int16) - ref["image"].astype(np.int16))))
        aud = job.fetch("audio")
        out["audio_rel_err"] = float(np.max(np.abs(aud - ref["audio"])) / np.max(np.abs(ref["audio"])))
    print(json.dumps(out))


if __name__ == "__main__":
    main()
