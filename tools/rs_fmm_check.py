"""GPU check + timing of the multipole resampler (wfx_d_resample_fmm) against the oracle's FFT form (= scipy.signal.resample's arithmetic).
python tools/rs_fmm_check.py [n0:num ...] [--time n0:num]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import wefax_oracle as wo
from wefax_amd import _native as nat

CASES = [(960000, 220500), (960000, 661500), (596801, 411220), (32768, 22580), (1000000, 999998), (40001, 2), (3000000, 689062), (1048576, 524288)]


def one(ctx, n0, num, time_only=False):
    rng = np.random.default_rng(n0 + num)
    x = rng.standard_normal(n0) * 1000 + 3000 * np.sin(np.arange(n0) * 0.7)
    px, py = ctx.dev_malloc(n0 * 8 + 64), ctx.dev_malloc(num * 8 + 64)
    ctx.dev_upload(px, x)
    ok = ctx.d_resample_fmm(px, n0, num, py)
    ctx.sync()
    if not ok:
        print(f"{n0} -> {num}: not handled")
        return
    if not time_only:
        got = ctx.dev_download(py, (num,), np.float64)
        ref = wo.resample_fft(x, num)
        err = np.max(np.abs(got - ref)) / np.max(np.abs(ref))
        k = int(np.argmax(np.abs(got - ref)))
        print(f"{n0} -> {num}: max relative error {err:.3e} at k = {k} (nan: {int(np.isnan(got).sum())})", flush=True)
    best = 1e9
    for _ in range(5):
        ctx.sync()
        t0 = time.perf_counter()
        ctx.d_resample_fmm(px, n0, num, py)
        ctx.sync()
        best = min(best, time.perf_counter() - t0)
    print(f"   {best * 1e3:.3f} ms  ({n0 / best / 1e9:.2f} G source samples/s)", flush=True)
    ctx.dev_free(px)
    ctx.dev_free(py)


if __name__ == "__main__":
    ctx = nat.Context(0)
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    cases = [tuple(int(v) for v in a.split(":")) for a in args] or CASES
    for n0, num in cases:
        one(ctx, n0, num)
    if "--big" in sys.argv:
        one(ctx, 57600000, 39690000, time_only=True)
        one(ctx, 172800000, 39690000, time_only=True)
