"""The envelope (a7 + median) of the transform route and of the multipole route against scipy on the SAME filtered audio, at lengths whose
transform route is the direct mixed-radix form (13-smooth half) and at general lengths (padded convolution).  Measured (round 6): the multipole route
5e-14 .. 8e-14 everywhere; the transform route 2e-15 on smooth lengths, 2e-10 .. 8e-10 on padded ones.
    python tools/route_accuracy.py"""
import sys, os, time
sys.path.insert(0, os.getcwd())
import numpy as np
from scipy.signal import hilbert, medfilt
from wefax_amd import _native as nat, synth
from wefax_amd.wefax import DecodeJob
x0 = synth.synth_capture(11025.0, noise=0.05, seed=3, image_lines=7110, black_tail_s=5.0)
for n in (39690002, 39690000, 14332502):
    x = np.concatenate([x0, x0[:2]])[:n]
    res = {}
    for name, mode in (("fft", nat.WFX_HILBERT_FFT), ("fmm", nat.WFX_HILBERT_FMM)):
        c = nat.Context(0)
        job = DecodeJob(c, x, 11025, 120, hilbert_mode=mode)
        job.run(); info = job.result()
        res[name] = (job.fetch("audio").copy(), job.fetch("envelope").copy())
        del job; c.close()
    t0 = time.time()
    a = res["fft"][0]
    print(n, "audio equal:", bool(np.array_equal(res["fft"][0], res["fmm"][0])), flush=True)
    env = medfilt(np.abs(hilbert(a)), 5)
    sc = np.max(env)
    for name in ("fft", "fmm"):
        e = np.abs(res[name][1] - env)
        print(f"   {name}: envelope against scipy on the same audio: max {e.max() / sc:.3e} at {int(e.argmax())}   ({time.time() - t0:.0f} s)", flush=True)
