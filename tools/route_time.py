"""The whole decode on the transform route and on the multipole route at several capture lengths (11 025 Hz, int16): which is faster where.
    python tools/route_time.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from wefax_amd import _native as nat, synth
from wefax_amd.wefax import DecodeJob

x10 = synth.config_c2(noise=0.05, seed=1)
for label, x in (("30 s", x10[:330750]), ("130 s", x10[:1433250]), ("5 min", x10[:3307500]), ("10 min", x10), ("20 min", np.tile(x10, 2)), ("60 min", np.tile(x10, 6)[:39690000])):
    x = np.ascontiguousarray(x[:x.shape[0] & ~1])
    res = {}
    for name, mode in (("fft", nat.WFX_HILBERT_FFT), ("fmm", nat.WFX_HILBERT_FMM)):
        c = nat.Context(0)
        job = DecodeJob(c, x, 11025, 120, hilbert_mode=mode)
        for _ in range(5):
            job.run()
        c.sync()
        best = 1e9
        for rep in range(3):
            t0 = time.perf_counter()
            for _ in range(10):
                job.run()
            c.sync()
            best = min(best, (time.perf_counter() - t0) / 10)
        res[name] = (best, job.fetch("digitalized"))
        del job
        c.close()
    same = bool(np.array_equal(res["fft"][1], res["fmm"][1]))
    print(f"{label:>7} ({x.shape[0]} samples): transform route {1e3 * res['fft'][0]:.4f} ms, multipole route {1e3 * res['fmm'][0]:.4f} ms  ({res['fft'][0] / res['fmm'][0]:.3f}x), same stream: {same}", flush=True)
