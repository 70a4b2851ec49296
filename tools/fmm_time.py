"""Time the fast multipole Hilbert transform (H + envelope) on BASELINE configs[1]'s length: wall time of the whole call and the
kernel groups by HIP events.  python tools/fmm_time.py [n]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from wefax_amd import _native as nat
n = int(sys.argv[1]) if len(sys.argv) > 1 else 7166250
mode = int(sys.argv[2]) if len(sys.argv) > 2 else 2      # 0: H, 1: envelope, 2: envelope + median
ctx = nat.Context(0)
rng = np.random.default_rng(3)
x = rng.standard_normal(n) * 1000 + 3000 * np.sin(np.arange(n) * 0.7)
px, po = ctx.dev_malloc(n * 8 + 64), ctx.dev_malloc(n * 8 + 64)
ctx.dev_upload(px, x)
for _ in range(3):
    ctx.d_hilbert_fmm(px, n, po, mode)
ctx.sync()
ts = []
for _ in range(10):
    t0 = time.perf_counter()
    for _ in range(10):
        ctx.d_hilbert_fmm(px, n, po, mode)
    ctx.sync()
    ts.append((time.perf_counter() - t0) / 10)
ctx.profile_reset(); ctx.profile_enable(True)
for _ in range(10):
    ctx.d_hilbert_fmm(px, n, po, mode)
ctx.sync(); ctx.profile_enable(False)
pr = {k: round(1e3 * v[1] / v[0], 1) for k, v in ctx.profile().items()}
print(f"n {n}: {1e3 * min(ts):.3f} ms per transform (10 back to back); kernel groups (us): up {pr.get('fmm_notch_p2m_m2m')} mid {pr.get('fmm_tiers_and_top')} tree {pr.get('fmm_tree_levels')} leaf {pr.get('fmm_near_l2p_env_median')}  all {pr}", flush=True)
