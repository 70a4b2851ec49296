#!/usr/bin/env python3
"""(a) the RCCL communicator with ONE rank (what a one-GPU box can run of the real transport), (b) the device test-signal
kernels against the NumPy generator, (c) the oversampled IQ capture: front end + sharded exact path, every rank emulated
on one GPU, against the one-GPU form and the oracle.  One JSON line per check."""
import json
import os
import sys
import tempfile
import time

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)

from wefax_amd import _native as nat                   # noqa: E402
from wefax_amd import polyphase as pp                   # noqa: E402
from wefax_amd import sharded, synth, synth_device      # noqa: E402
from wefax_amd.wefax import DecodeJob                   # noqa: E402


def oracle(x, sr, lpm=120):
    from oracle import wefax_oracle as wo
    with tempfile.TemporaryDirectory() as td:
        p = os.path.join(td, "c.wav")
        synth.write_wav(p, sr, x)
        return wo.process(p, lpm, want_messages=False)


def stats(a, b):
    d = np.abs(a.astype(np.int16) - b.astype(np.int16))
    return {"max": int(d.max()), "mean": round(float(d.mean()), 5), "gt1": int(np.count_nonzero(d > 1)), "ne": int(np.count_nonzero(d))}


def rccl_one_rank():
    kw = dict(start_tone_s=5.0, phasing_lines=20, image_lines=220, stop_tone_s=2.0, black_tail_s=3.0)
    x = synth.synth_capture(11025.0, noise=0.05, seed=3, **kw)
    ctx = nat.Context(0)
    job = DecodeJob(ctx, x, 11025, 120)
    job.run()
    ref_img, ref_stream = job.fetch("image"), job.fetch("digitalized")
    uid = sharded.bootstrap_unique_id(0, 1)
    ctx2 = nat.Context(0)
    comm = nat.Comm.rccl(ctx2, uid, 1, 0)
    dec = sharded.ShardedDecoder(ctx2, comm, x.shape[0], 11025, 120, nat.WFX_IN_I16_MONO, data=x)
    dec.run()
    info = dec.result()
    img = dec.fetch("image")
    ctx2.sync()
    t0 = time.perf_counter()
    for _ in range(20):
        dec.run()
    ctx2.sync()
    ms = (time.perf_counter() - t0) / 20 * 1e3
    t0 = time.perf_counter()
    for _ in range(20):
        job.run()
    ctx.sync()
    ms1 = (time.perf_counter() - t0) / 20 * 1e3
    print(json.dumps({"check": "rccl_one_rank", "is_rccl": comm.is_rccl, "image_equal": bool(np.array_equal(img, ref_img)),
                      "stream_equal": bool(np.array_equal(dec.fetch("stream"), ref_stream)), "start": int(info.start_frame),
                      "ms_per_decode_sharded_form": round(ms, 3), "ms_per_decode_fused": round(ms1, 3)}), flush=True)
    dec.close()
    comm.close()
    ctx2.close()
    ctx.close()


def synth_vs_numpy():
    ctx = nat.Context(0)
    for fs, iq in ((11025.0, False), (192000.0, True)):
        kw = dict(start_tone_s=1.0, phasing_lines=10, image_lines=20, stop_tone_s=1.0, black_tail_s=1.0)
        ref = synth.synth_capture(fs, noise=0.0, seed=0, iq=iq, **kw)
        p = synth_device.synth_params(fs, noise=0.0, iq=iq, **kw)
        n0 = int(ctx.lib.wfx_synth_frames(p))
        ptr = synth_device.synth_slice(ctx, p, 0, n0)
        got = ctx.dev_download(ptr, ref.shape, np.int16)
        d = np.abs(got.astype(np.int32) - ref.astype(np.int32))
        # a wrapped slice must repeat the capture
        lo, hi = n0 - 1000, n0 + 1500
        ptr2 = synth_device.synth_slice(ctx, p, lo, hi)
        got2 = ctx.dev_download(ptr2, (hi - lo,) + ref.shape[1:], np.int16)
        want2 = got[np.arange(lo, hi) % n0]
        pn = synth_device.synth_params(fs, noise=0.05, seed=7, iq=iq, **kw)
        ptr3 = synth_device.synth_slice(ctx, pn, 0, n0)
        gn = ctx.dev_download(ptr3, ref.shape, np.int16).astype(np.float64) - got
        print(json.dumps({"check": "synth_vs_numpy", "fs": fs, "iq": iq, "frames": n0, "frames_numpy": int(ref.shape[0]), "max_abs_diff": int(d.max()),
                          "n_diff": int(np.count_nonzero(d)), "wrap_equal": bool(np.array_equal(got2, want2)),
                          "noise_std_counts": round(float(gn.std()), 2), "noise_std_expected": round(0.05 * 32767, 2),
                          "noise_mean": round(float(gn.mean()), 3)}), flush=True)
        for q in (ptr, ptr2, ptr3):
            ctx.dev_free(q)
    ctx.close()


def iq_front_end(worlds):
    fs = 1536000
    t_line, phasing, seconds = 0.5, 20, 40.0
    lines = int(round((seconds - 3.0) / t_line)) - phasing
    x = synth.synth_capture(float(fs), noise=0.05, seed=0, lpm=120, phasing_lines=phasing, image_lines=lines, start_tone_s=1.0, stop_tone_s=1.0,
                            black_tail_s=1.0, iq=True)
    ref = oracle(x, fs, 120)
    fe = pp.FrontEnd(fs, stop_rate=int(os.environ.get("WFX_FE_STOP", 16000)))
    ctx = nat.Context(0)
    one = sharded.FrontEndExactDecoder(ctx, fe, x, lines_per_minute=120)
    one.run()
    info1 = one.result()
    img1, st1 = one.fetch("image"), one.fetch("digitalized")
    print(json.dumps({"check": "iq_one_gpu_fused", "start_eq_oracle": bool(info1.start_frame == ref["start_frame"]), "image": stats(img1, ref["image"]),
                      "stream": stats(st1, ref["digitalized"])}), flush=True)
    one.close()
    ctx.close()
    first = None
    for w in worlds:
        mk = lambda c, m: sharded.FrontEndShardedDecoder(c, m, fe, x, lines_per_minute=120, plan="dist")      # noqa: E731
        try:
            r = sharded.decode_emulated(x, fs, w, 120, make_decoder=mk)
        except Exception as e:
            print(json.dumps({"check": "iq_sharded", "world": w, "error": str(e)[:300]}), flush=True)
            continue
        rec = {"check": "iq_sharded", "world": w, "first_radix": list(r["first_radix"]), "start_eq_oracle": bool(r["sync"]["start_frame"] == ref["start_frame"]),
               "stream_vs_oracle": stats(r["digitalized"], ref["digitalized"]), "stream_vs_one_gpu": stats(r["digitalized"], st1)}
        if "image" in r and r["image"].shape == ref["image"].shape:
            rec["image_vs_oracle"] = stats(r["image"], ref["image"])
            rec["image_vs_one_gpu"] = stats(r["image"], img1)
        if first is None:
            first = r
        else:
            rec["bit_identical_to_first_world"] = bool(np.array_equal(r["envelope"], first["envelope"]) and np.array_equal(r["digitalized"], first["digitalized"])
                                                       and np.array_equal(r["image"], first["image"]))
        print(json.dumps(rec), flush=True)


if __name__ == "__main__":
    what = sys.argv[1].split(",") if len(sys.argv) > 1 else ["rccl", "synth", "iq"]
    if "rccl" in what:
        rccl_one_rank()
    if "synth" in what:
        synth_vs_numpy()
    if "iq" in what:
        iq_front_end([1, 2, 3, 8])
