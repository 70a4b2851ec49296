#!/usr/bin/env python3
"""Where the wav -> png time of one fresh Demodulator goes (tools/e2e.py times the whole)."""
import os, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from wefax_amd import Demodulator, synth, hostparams as hp, _native as nat
from wefax_amd.wefax import DecodeJob

x = synth.synth_capture(11025.0, noise=0.05, seed=0, image_lines=1100)
with tempfile.TemporaryDirectory() as td:
    wav, png = os.path.join(td, "in.wav"), os.path.join(td, "out.png")
    synth.write_wav(wav, 11025, x)
    d = Demodulator(wav, 120, quiet=True, tcp_stream=False); d.process(); d.save_output_image(png)     # warm-up
    for rep in range(2):
        T = [("start", time.perf_counter())]
        ctx = nat.Context(0); T.append(("context", time.perf_counter()))
        sr, data = hp.read_wav(wav, alloc=ctx.staging); T.append(("read_wav", time.perf_counter()))
        job = DecodeJob(ctx, data, sr, 120, hp.load_notch_settings()); T.append(("job (params, upload)", time.perf_counter()))
        job.run(); T.append(("run (enqueue, plans)", time.perf_counter()))
        info = job.result(); T.append(("result (sync)", time.perf_counter()))
        ctx.decode_save_png(png, deflate=True); T.append(("png", time.perf_counter()))
        ctx.close(); T.append(("close", time.perf_counter()))
        print(" | ".join(f"{n} {1e3 * (t - T[i][1]):.2f}" for i, (n, t) in enumerate(T[1:])), "| total %.2f ms" % (1e3 * (T[-1][1] - T[0][1])))
