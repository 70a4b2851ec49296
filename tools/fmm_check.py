"""The fast multipole Hilbert transform on the device (csrc/wfx_fmm.hip) against the oracle's FFT form and against its NumPy model."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
import numpy as np
from oracle import wefax_oracle as wo
from wefax_amd import _native as nat, synth
ctx = nat.Context(0)
rng = np.random.default_rng(3)
for n in (32768, 40000, 100000, 250008, 1433250, 7166250):
    if n == 7166250:
        x = synth.config_c2(noise=0.05, seed=0).astype(np.float64)
    else:
        x = rng.standard_normal(n) * 1000 + 3000 * np.sin(np.arange(n) * 0.7)
    ref = wo.hilbert_fft(x).imag
    px, po = ctx.dev_malloc(n * 8 + 64), ctx.dev_malloc(n * 8 + 64)
    ctx.dev_upload(px, x)
    ok = ctx.d_hilbert_fmm(px, n, po)
    got = ctx.dev_download(po, (n,), np.float64)
    err = np.max(np.abs(got - ref)) / np.max(np.abs(ref))
    ctx.sync()
    ts = []
    for _ in range(8):
        t0 = time.perf_counter()
        ctx.d_hilbert_fmm(px, n, po, True)
        ctx.sync()
        ts.append(time.perf_counter() - t0)
    ctx.profile_reset(); ctx.profile_enable(True)
    for _ in range(5):
        ctx.d_hilbert_fmm(px, n, po, True)
    ctx.sync(); ctx.profile_enable(False)
    pr = {k: round(1e3 * v[1] / v[0], 1) for k, v in ctx.profile().items()}
    env = ctx.dev_download(po, (n,), np.float64)
    eerr = np.max(np.abs(env - np.abs(x + 1j * ref))) / np.max(np.abs(ref))
    print(f"n {n:8d} handled {ok} max relative error H {err:.3e}  envelope {eerr:.3e}  {1e3 * min(ts):.3f} ms  kernels (us): {pr}", flush=True)
    ctx.dev_free(px), ctx.dev_free(po)
