"""HIP-event profiles of the library's kernels -> per-kernel tables and the `roofline` object of the bench line."""
from __future__ import annotations

import json
import os
import subprocess
import sys
import time

import benchlib
from benchlib import REPO, HBM_PEAK_GBS, IQ_FS


def dist_of(vals) -> dict:
    """min / median / p90 / max of a list of per-step figures (a mean of 10 hides a 30 % spread)."""
    v = sorted(float(x) for x in vals)
    if not v:
        return {}
    return {"n": len(v), "min": round(v[0], 3), "median": round(v[len(v) // 2], 3), "p90": round(v[min(len(v) - 1, int(0.9 * len(v)))], 3), "max": round(v[-1], 3)}


def kernel_table(prof: dict, steps: int) -> dict:
    return {k: {"launches_per_step": round(v[0] / steps, 2), "avg_us": round(1e3 * v[1] / v[0], 2), "us_per_step": round(1e3 * v[1] / steps, 1)}
            for k, v in prof.items()}


def profile_pass(ctx, step, steps: int, sync=None) -> dict:
    """Second pass of `steps` steps with a HIP-event pair around every launch (the pairs would otherwise sit inside the timed region)."""
    ctx.profile_reset()
    ctx.profile_enable(True)
    for _ in range(steps):
        step()
    (sync or ctx.sync)()
    ctx.profile_enable(False)
    return ctx.profile()


def roofline_of(prof: dict, steps: int, alg_bytes: int, ms_per_step: float, pmc_file: str | None, merge_fft=True) -> dict:
    """achieved = algorithmic bytes of the step (input bytes + 4 output pixels per sample: what ONE launch of a
    whole-capture kernel stands for) / the dominant kernel's average launch time."""
    fam = dict(prof)
    if merge_fft and "fft_pass_fwd" in fam and "fft_pass_inv" in fam:      # forward and inverse passes are one kernel template
        f, i = fam.pop("fft_pass_fwd"), fam.pop("fft_pass_inv")
        fam["fft_pass"] = (f[0] + i[0], f[1] + i[1])
    if not fam:      # this rank launched nothing (the single plan leaves every rank but 0 idle): no kernel to put against the roof
        return {"bound": "hbm", "kernel": None, "achieved": None, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": None, "traffic": None,
                "note": "no kernel ran on this rank"}
    dom = max(fam.items(), key=lambda kv: kv[1][1])
    avg_s = dom[1][1] / dom[1][0] / 1e3
    traffic = None
    traffic_source = None
    if pmc_file and os.path.exists(pmc_file):
        try:
            tj = json.load(open(pmc_file))
            traffic_source = (os.path.relpath(pmc_file, REPO) + " (rocprofv3 --pmc FETCH_SIZE x2 + WRITE_SIZE, separate passes; collected at "
                              + str(tj.get("_collected_at", "an earlier tree")) + ", not in this run)")
            if dom[0] == "fft_pass":
                parts = [tj[k]["hbm_bytes_per_launch"] for k in ("fft_pass_fwd", "fft_pass_inv") if k in tj]
                traffic = int(sum(parts) / len(parts)) if parts else None
            else:
                traffic = tj.get(dom[0], {}).get("hbm_bytes_per_launch")
        except Exception:
            traffic = None
    # every kernel's counter bytes of one step against the algorithmic bytes (33x for configs[1]: the exact transform path streams its work
    # arrays six times; 1.03x for the ingest) -- from the committed counter summary and THIS run's launch counts
    traffic_ratio = None
    if pmc_file and os.path.exists(pmc_file):
        try:
            tj = json.load(open(pmc_file))
            tot = sum(tj[k]["hbm_bytes_per_launch"] * (v[0] / steps) for k, v in prof.items() if k in tj and isinstance(tj[k], dict))
            traffic_ratio = round(tot / alg_bytes, 2) if tot else None
        except Exception:      # noqa: BLE001
            traffic_ratio = None
    achieved = alg_bytes / avg_s / 1e9
    return {"bound": "hbm", "kernel": dom[0], "traffic_ratio_whole_path": traffic_ratio, "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic, "traffic_source": traffic_source if traffic is not None else None,
            "algorithmic_bytes_per_launch": int(alg_bytes),
            "avg_launch_us": round(avg_s * 1e6, 2), "launches_per_step": round(dom[1][0] / steps, 2),
            "whole_path_frac": round(alg_bytes / (ms_per_step / 1e3) / 1e9 / HBM_PEAK_GBS, 5)}
