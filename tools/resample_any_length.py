"""What a resampled capture of ARBITRARY length costs on one GPU (wefax.py:384 takes whatever the wav holds): the 60-minute and the
10-minute 48 kHz captures whole (13-smooth halves: the mixed-radix packed resampler) and less one / two samples (the general form).
    python tools/resample_any_length.py [--minutes 60]"""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from wefax_amd import _native as nat
from wefax_amd import synth_device
from wefax_amd.wefax import DecodeJob

ap = argparse.ArgumentParser()
ap.add_argument("--minutes", type=float, default=60.0)
ap.add_argument("--steps", type=int, default=3)
args = ap.parse_args()
ctx = nat.Context(0)
lines = int(args.minutes * 120) - 90
sp = synth_device.synth_params(48000.0, noise=0.05, seed=0, iq=False, image_lines=lines, black_tail_s=5.0)
n0 = int(ctx.lib.wfx_synth_frames(sp))
ptr = synth_device.synth_slice(ctx, sp, 0, n0)
x = ctx.dev_download(ptr, (n0,), np.int16)
ctx.dev_free(ptr)
ref = None
for trim in (0, 2, 1):
    xs = np.ascontiguousarray(x[:n0 - trim])
    t0 = time.perf_counter()
    job = DecodeJob(ctx, xs, 48000, 120)
    job.run()
    info = job.result()
    first = time.perf_counter() - t0
    t0 = time.perf_counter()
    for _ in range(args.steps):
        job.run()
    job.result()
    ms = 1e3 * (time.perf_counter() - t0) / args.steps
    out = {"n0": int(xs.shape[0]), "n": int(job.n), "ms_per_decode": round(ms, 3), "first_decode_s": round(first, 2), "start_frame": int(info.start_frame),
           "low": info.low, "high": info.high}
    print(json.dumps(out), flush=True)
    del job
