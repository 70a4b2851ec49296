#!/usr/bin/env python3
"""Randomised cross-check of the two forms of the peak scan (joined segments vs sequential): many synthetic byte
streams (pulse trains with jitter / drop-outs / level steps, noise, plateaus), several line rates.
    python tools/pick_sweep.py [--cases 60]"""
import argparse, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from wefax_amd import _native as nat, hostparams as hp  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--cases", type=int, default=60)
args = ap.parse_args()
ctx = nat.Context(0)
forms = {1: 0, -1: 0, 0: 0}
bad = 0
for case in range(args.cases):
    rng = np.random.default_rng(1000 + case)
    lpm = [120, 240, 90, 100, 180][case % 5]
    frame_len = 1 / (lpm / 60)
    n1, n0, mind = hp.sync_constants(11025, frame_len)
    w = int(frame_len * 11025)
    lines = int(rng.integers(30, 400))
    n = lines * w + int(rng.integers(0, w))
    kind = case % 6
    if kind == 0:
        d = rng.integers(0, 256, size=n).astype(np.uint8)
    elif kind == 1:
        d = np.full(n, int(rng.integers(0, 256)), np.uint8)
        d[rng.integers(0, n, 50)] = 0
    else:
        d = rng.integers(80, 256, size=n).astype(np.uint8)
        L = 2 * n1 + n0
        jitter = [0, 40, 400, 2000][kind - 2] if kind < 6 else 0
        for k in range(0, n - 600, w):
            if rng.random() < 0.08:
                continue
            j = k + int(rng.integers(0, jitter + 1))
            if j + L < n:
                d[j:j + L] = rng.integers(0, 30, size=L)
        if case % 4 == 0:
            a = int(rng.integers(0, max(1, n - 60000)))
            d[a:a + 50000] = (240 - 20 * (np.arange(50000) // 4000)).clip(0, 255).astype(np.uint8)
    os.environ.pop("WFX_PICK_SEG", None)
    got = ctx.sync_peaks(d, n1, n0, mind)
    form = ctx.debug_counters()[7]
    os.environ["WFX_PICK_SEG"] = "0"
    ref = ctx.sync_peaks(d, n1, n0, mind)
    forms[form] = forms.get(form, 0) + 1
    if got != ref:
        bad += 1
        print("MISMATCH case", case, "lpm", lpm, "kind", kind, "n", n, "form", form, "npeaks", len(got[0]), len(ref[0]))
print("cases", args.cases, "forms", forms, "mismatches", bad)
sys.exit(1 if bad else 0)
