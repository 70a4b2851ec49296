"""Plan 3 of the sharded decode (chunk-local fast multipole Hilbert transform) in emulated worlds on one GPU: every world size must give the
bytes of the one-GPU decode with the same Hilbert form -- audio, envelope, stream, start frame, image -- and the oracle's stream.
    python tools/shard_fmm_check.py [seconds] [worlds...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import wefax_oracle as wo
from wefax_amd import _native as nat, synth, sharded
from wefax_amd.wefax import DecodeJob

secs = float(sys.argv[1]) if len(sys.argv) > 1 else 650.0
worlds = [int(v) for v in sys.argv[2:]] or [1, 2, 3, 8]
x = synth.config_c2(noise=0.05, seed=1)
if secs < 650:
    n = int(secs * 11025) & ~1
    x = x[:n]
print("capture", x.shape, x.dtype, flush=True)
c = nat.Context(0)
job = DecodeJob(c, x, 11025, 120, hilbert_mode=nat.WFX_HILBERT_FMM)
job.run()
info = job.result()
ref = {"dig": job.fetch("digitalized"), "env": job.fetch("envelope"), "audio": job.fetch("audio"), "start": info.start_frame, "img": job.fetch("image")}
job2 = DecodeJob(c, x, 11025, 120, hilbert_mode=nat.WFX_HILBERT_FFT)
job2.run()
job2.result()
print("one GPU: multipole stream == transform stream:", bool(np.array_equal(ref["dig"], job2.fetch("digitalized"))), flush=True)
ok = True
for w in worlds:
    t0 = time.time()
    out = sharded.decode_emulated(x, 11025, w, 120, plan="fmm")
    same = {k: bool(np.array_equal(out[kk], ref[k])) for k, kk in (("dig", "digitalized"), ("env", "envelope"), ("audio", "audio"))}
    same["blocks"] = bool(np.array_equal(out["digitalized_blocks"], ref["dig"]))
    same["start"] = out["sync"]["start_frame"] == ref["start"]
    same["img"] = "image" in out and bool(np.array_equal(out["image"], ref["img"]))
    wire = out["wire"]
    per_rank = [sum(int(e["sent"]) for e in ws if e["name"] != "stream gather") for ws in wire] if wire and isinstance(wire[0], list) and wire[0] and isinstance(wire[0][0], dict) else None
    print(f"world {w}: plan {out['plan']} {same}  ({time.time() - t0:.1f} s)", flush=True)
    if per_rank is not None:
        print("   bytes sent per rank, all collectives but the stream gather:", per_rank, "   collectives:", [(e["name"], int(e["sent"])) for e in wire[-1]], flush=True)
    else:
        print("   wire:", wire[-1] if wire else None, flush=True)
    ok &= all(same.values())
print("ALL IDENTICAL" if ok else "MISMATCH", flush=True)
sys.exit(0 if ok else 1)
