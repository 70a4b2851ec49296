"""Plan 3 of the sharded decode in front of a resampler (multipole forms of scipy.signal.resample AND of the Hilbert transform, both chunk-local)
in emulated worlds on one GPU.  Every world size must give the SAME bytes (audio, envelope, stream, start frame, image); against the one-GPU decode
with the transform-based resampler the audio agrees to ~1e-13 and the stream to the parity bar (<= 1 grey level, a handful of bytes).
    python tools/shard_rs_check.py [rate] [seconds] [worlds...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from wefax_amd import _native as nat, synth, sharded
from wefax_amd.wefax import DecodeJob

rate = int(sys.argv[1]) if len(sys.argv) > 1 else 48000
secs = float(sys.argv[2]) if len(sys.argv) > 2 else 60.0
worlds = [int(v) for v in sys.argv[3:]] or [1, 2, 3, 8]
lines = max(10, int((secs - 20) * 2))
x = synth.synth_capture(float(rate), noise=0.05, seed=5, start_tone_s=3.0, phasing_lines=20, image_lines=lines, stop_tone_s=2.0, black_tail_s=3.0)
if x.shape[0] % 2:
    x = x[:-1]
print("capture", x.shape, x.dtype, "->", int(11025 * (x.shape[0] / rate)), "samples", flush=True)
c = nat.Context(0)
job = DecodeJob(c, x, rate, 120)
job.run()
info = job.result()
ref = {"dig": job.fetch("digitalized"), "env": job.fetch("envelope"), "audio": job.fetch("audio"), "start": info.start_frame, "img": job.fetch("image")}
first = None
ok = True
for w in worlds:
    t0 = time.time()
    out = sharded.decode_emulated(x, rate, w, 120, plan="fmm")
    if first is None:
        first = out
        da = np.max(np.abs(out["audio"] - ref["audio"])) / np.max(np.abs(ref["audio"]))
        dd = out["digitalized"].astype(np.int16) - ref["dig"].astype(np.int16)
        print(f"world {w} (plan {out['plan']}) against the one-GPU decode: audio {da:.2e} relative, stream {int(np.count_nonzero(dd))} bytes differ (max {int(np.max(np.abs(dd)))}), "
              f"start frame {out['sync']['start_frame']} / {ref['start']}, image max diff "
              f"{int(np.max(np.abs(out['image'].astype(np.int16) - ref['img'].astype(np.int16)))) if 'image' in out else None}", flush=True)
        ok &= da < 1e-11 and int(np.max(np.abs(dd))) <= 1 and out["sync"]["start_frame"] == ref["start"]
    same = {k: bool(np.array_equal(out[k], first[k])) for k in ("digitalized", "envelope", "audio", "digitalized_blocks")}
    same["start"] = out["sync"]["start_frame"] == first["sync"]["start_frame"]
    same["img"] = ("image" in out) == ("image" in first) and ("image" not in out or bool(np.array_equal(out["image"], first["image"])))
    wire = out["wire"]
    print(f"world {w}: plan {out['plan']} identical to world {worlds[0]}: {same}  ({time.time() - t0:.1f} s)", flush=True)
    try:
        per_rank = [sum(int(e["sent"]) for e in ws if e["name"] != "stream gather") for ws in wire]
        print("   bytes sent per rank, all collectives but the stream gather:", per_rank, "   collectives:", [(e["name"], int(e["sent"])) for e in wire[-1]], flush=True)
    except Exception:
        print("   wire:", wire[-1] if wire else None, flush=True)
    ok &= all(same.values())
print("ALL IDENTICAL" if ok else "MISMATCH", flush=True)
sys.exit(0 if ok else 1)
