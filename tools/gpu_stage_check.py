#!/usr/bin/env python3
"""Stage-by-stage comparison of the HIP path with the oracle on the golden inputs.
Diagnostic tool (prints numbers, asserts nothing): `gpurun -- python tools/gpu_stage_check.py`."""
import json
import os
import sys
import time
import traceback

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from oracle import wefax_oracle as wo          # noqa: E402
from wefax_amd import _native as nat           # noqa: E402
from wefax_amd import hostparams as hp         # noqa: E402
from wefax_amd.wefax import Demodulator        # noqa: E402

G = os.path.join(REPO, "tests", "golden")


def rel(a, b):
    s = np.max(np.abs(b)) or 1.0
    return float(np.max(np.abs(a - b)) / s)


def main():
    only = sys.argv[1:] or None
    ctx = nat.Context(0)
    sys.path.insert(0, G)
    import recipes
    recipes.ensure_all(G)                      # the inputs are regenerated from their recipes where they are missing
    cases = json.load(open(os.path.join(G, "manifest.json")))["cases"]
    for c in cases:
        if only and c["name"] not in only:
            continue
        print("=====", c["name"], flush=True)
        path = os.path.join(G, c["input"])
        ref = wo.process(path, c["lpm"])
        sr, data = wo.read_wav(path)
        try:
            x = data
            if data.ndim == 2:
                m = ctx.merge_channels(data)
                print(" merge exact:", np.array_equal(m, wo.merge_channels(data)))
                x = m
            if sr != 11025:
                num = int(11025 * (len(x) / sr))
                t = time.time()
                y = ctx.resample(np.asarray(x, dtype=np.float64), num)
                yr = wo.resample_fft(x, num)
                print(f" resample rel err {rel(y, yr):.3e}  ({time.time()-t:.2f}s)")
                x = yr
            b, a = hp.iirnotch(2600, 1, 11025)
            au = ctx.notch_filtfilt(x, b, a)
            print(f" notch rel err {rel(au, ref['audio']):.3e}  edges {rel(au[:80], ref['audio'][:80]):.3e} {rel(au[-80:], ref['audio'][-80:]):.3e}")
            env = ctx.analytic_env(ref["audio"])
            print(f" env(fft) rel err {rel(env, ref['demod']):.3e}")
            n = len(ref["demod"])
            ranks = [hp.percentile_plan(n, 0.5)[0], hp.percentile_plan(n, 0.5)[1],
                     hp.percentile_plan(n, 99.5)[0], hp.percentile_plan(n, 99.5)[1], 0, n - 1]
            os_ = ctx.order_stats(ref["demod"], ranks)
            srt = np.sort(ref["demod"])
            print(" order stats exact:", np.array_equal(os_, srt[ranks]))
            dq, nan = ctx.quantise(ref["demod"], ref["low"], ref["high"])
            print(" quantise exact:", np.array_equal(dq, ref["digitalized"]), "nan", nan)
            n1, n0, mind = hp.sync_constants(11025, 1 / (c["lpm"] / 60))
            cr = ctx.sync_corr(ref["digitalized"], n1, n0)
            print(" corr exact:", np.array_equal(cr.astype(np.int64), wo.sync_correlation(ref["digitalized"], n1, n0)))
            pk, first, hit = ctx.sync_peaks(ref["digitalized"], n1, n0, mind)
            rpk, rfirst, rhit = wo.pick_peaks(wo.sync_correlation(ref["digitalized"], n1, n0), mind)
            print(" peaks exact:", pk == rpk, first == rfirst, hit == rhit, len(pk))
            if "image" in ref:
                w = int(1 / (c["lpm"] / 60) * 11025)
                img = ctx.lines_to_image(ref["digitalized"], ref["start_frame"], w)
                print(" image exact:", np.array_equal(img, ref["image"]), img.shape)
        except Exception:
            traceback.print_exc()
        # whole path
        try:
            d = Demodulator(path, lines_per_minute=c["lpm"], quiet=True, tcp_stream=True)
            t = time.time()
            try:
                d.process()
                exc = None
            except (ValueError, IndexError) as e:
                exc = [type(e).__name__, str(e)]
            dt = time.time() - t
            print(f" process(): {dt:.3f}s exc={exc} ref_exc={c['exception']}")
            dg = d.digitalized_data
            nd = int(np.count_nonzero(dg != ref["digitalized"]))
            print(f"  digitalized mismatches {nd}/{len(dg)} max {int(np.max(np.abs(dg.astype(int)-ref['digitalized'].astype(int))))}")
            print(f"  low/high rel {abs(d._low-ref['low'])/ref['low']:.2e} {abs(d._high-ref['high'])/ref['high']:.2e}")
            print("  peaks equal:", d.peaks == ref["peaks"])
            if exc is None:
                print("  start_frame", d.start_frame, ref["start_frame"], "phasing equal", d.phasing_signals == list(ref["phasing_signals"]))
                im = d.output_array
                print("  image", im.shape, "max|d|", int(np.max(np.abs(im.astype(int) - ref["image"].astype(int)))),
                      "mismatch px", int(np.count_nonzero(im != ref["image"])))
            msgs = [[m.get("data_type"), m.get("progress_title", m.get("message_content")),
                     None if "percentage" not in m else float(m["percentage"])] for m in d.websocket_stack]
            print("  messages equal:", msgs == ref["messages"], len(msgs), len(ref["messages"]))
            d.close()
        except Exception:
            traceback.print_exc()
    print("profile:", ctx.profile())


if __name__ == "__main__":
    main()
