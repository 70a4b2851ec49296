#!/usr/bin/env python3
"""Benchmark of the WEFAX demod->pixel hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W

A step is one pass of the whole path (ingest -> notch -> analytic envelope -> median -> percentiles -> quantise -> sync
search -> lines -> 4x bicubic image) over one synthetic capture that is already resident in HBM.

The line rank 0 prints (ONE JSON line):
  * `value` -- BASELINE.json configs[1]: a synthetic 10-minute 11 025 Hz mono capture (7 166 250 int16 samples, 120 LPM).
    At N > 1 every rank decodes its own capture of that shape: captures are independent objects, so this metric shards
    with no data-path collective ("scaling": "weak").  value = samples of all ranks / max-over-ranks time of the K steps.
  * `c4_strong` -- BASELINE.json configs[3], the curve the north star asks for: ONE 60-minute 1.536 MS/s int16 IQ stream
    (5.53 G frames, 22 GB, synthesised in HBM) decoded by all N ranks together -- each rank runs the time-domain front end on
    its 1/N of the stream, then the sharded exact path (distributed FFT resample + Hilbert over RCCL, csrc/wfx_dist.hip);
    ms per decode, frames/s, RCCL ranks observed, the one-GPU time measured in the same run and the efficiency against it.
  * `roofline` of the dominant kernel (HIP events on the library's stream, second pass of K steps) and `cpu_baseline`
    (the oracle, a NumPy/C port of wefax.py, one thread on this host) -- both at N = 1 only for the CPU leg.

With `--gpus N` and no WORLD_SIZE in the environment this process only SPAWNS the N ranks (fresh child processes, one per
GPU; the parent never touches a GPU) and relays rank 0's line.  Under `python -m torch.distributed.run` the ranks come
from the environment (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_*).  No PyTorch is imported either way: the communicator is
RCCL bound by libwefax_hip.so itself; the 128-byte unique id travels over a loopback TCP socket next to MASTER_PORT.
"""
from __future__ import annotations

import argparse
import json
import os
import subprocess
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

import benchlib                                                                                     # noqa: E402
from benchlib import HBM_PEAK_GBS, IQ_FS                                                            # noqa: E402,F401
from benchlib.ranks import Ranks, _LineGuard, rccl_probe_main, spawn_ranks                          # noqa: E402,F401
from benchlib.gpustate import GpuState                                                              # noqa: E402,F401
from benchlib.measure import dist_of, kernel_table, profile_pass, roofline_of                       # noqa: E402,F401
from benchlib.e2e import bench_e2e                                                                  # noqa: E402,F401
from benchlib.workloads import (bench_c2, bench_c2_strong, bench_c3, bench_c5, bench_fmm, bench_general_lengths, bench_iq,      # noqa: E402,F401
                                iq_recipe, wire_object)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--noise", type=float, default=0.05, help="AWGN sigma in full-scale units")
    ap.add_argument("--no-cpu", action="store_true", help="skip the CPU baseline leg")
    ap.add_argument("--no-cpu-loops", action="store_true",
                    help="skip the second CPU timing that keeps the reference's per-sample Python loops (about 7 s)")
    ap.add_argument("--short", action="store_true", help="60-line capture and a 40-second IQ stream (debugging only)")
    ap.add_argument("--workload", choices=["c2", "iq", "c3"], default="c2",
                    help="c2 (default): BASELINE configs[1] as `value` plus the c4_strong object.  iq: only BASELINE configs[3], the "
                         "IQ stream of --iq-seconds, as the line itself.  c3: BASELINE configs[2], the 60-minute 48 kHz capture "
                         "(exact FFT resample included), one capture sharded over the ranks")
    ap.add_argument("--shard", action="store_true",
                    help="c2: decode ONE 10-minute capture with all ranks (sharded exact path) instead of one capture per rank")
    ap.add_argument("--iq-seconds", type=float, default=3600.0, help="length of the IQ stream (BASELINE: 60 minutes)")
    ap.add_argument("--iq-stop-rate", type=int, default=16000, choices=[16000, 24000, 48000],
                    help="rate at which the time-domain front end hands the IQ stream to the exact FFT resampler")
    ap.add_argument("--iq-form", choices=["auto", "fused", "sharded"], default="auto",
                    help="one GPU: the fused exact decode behind the front end (auto) or the sharded form with one rank")
    ap.add_argument("--plan", choices=["auto", "dist", "single", "rows", "auto-rows", "fmm"], default="auto",
                    help="sharded decodes: auto = the library's cost model picks the distributed form or rank 0 alone (DESIGN 6.6); dist / single "
                         "force one; rows = distributed in the rows layout of rounds 2-3 (8 array transposes instead of 4: A/B runs)")
    ap.add_argument("--trim", type=int, default=0, help="c2: drop this many samples from the end of the capture (--trim 2 with --shard: the padded distributed convolution; odd: the real convolution on packed transforms)")
    ap.add_argument("--no-c5", action="store_true", help="c2: leave the c5 object (64 mixed captures on 8 contexts) out")
    ap.add_argument("--no-c4", action="store_true", help="c2: leave the c4_strong object out (quick runs)")
    ap.add_argument("--no-extras", action="store_true", help="c2: leave the general_length and c3 objects out (kernel profiles of the headline alone)")
    ap.add_argument("--no-e2e", action="store_true", help="leave the file-to-file objects (wav on tmpfs -> png on tmpfs) out")
    ap.add_argument("--no-pcie", action="store_true", help="c2: skip the PCIe-inclusive leg (it runs two decodes concurrently: keep it out of kernel profiles)")
    ap.add_argument("--batch", type=int, default=1,
                    help="captures decoded concurrently per GPU, one native context (= HIP stream) each; "
                         "BASELINE configs[4] uses 8 per GPU with mixed 120/240 LPM, IOC576/288 members")
    ap.add_argument("--rccl-probe", action="store_true", help=argparse.SUPPRESS)      # child of a rank: RCCL bootstrap + self-test, see Ranks
    return ap.parse_args()


# ---- CPU leg ---------------------------------------------------------------------------------------------------
def cpu_baseline(x, sample_rate: int, lpm: int, faithful: bool, what: str) -> dict:
    """Time the oracle (port of wefax.py) on this host: 1 process, 1 thread."""
    import tempfile
    import numpy as np
    from oracle import wefax_oracle as wo
    from wefax_amd import synth
    with tempfile.TemporaryDirectory() as td:
        p = os.path.join(td, "capture.wav")
        synth.write_wav(p, sample_rate, x)
        t0 = time.perf_counter()
        r = wo.process(p, lpm, want_messages=False)
        dt = time.perf_counter() - t0
        out = {"value": round(x.shape[0] / dt / 1e6, 4), "unit": "Msamples/s", "cores": 1, "kind": "port", "host_cpus": os.cpu_count(),
               "seconds": round(dt, 3), "sample": what, "_result": r}
        if faithful:
            # the same port with the reference's per-sample Python loops kept (list build, np.dot per offset, putpixel):
            # what wefax.py itself costs, without and with its 7 s of time.sleep
            t0 = time.perf_counter()
            rf = wo.process(p, lpm, want_messages=False, faithful_loops=True)
            dtf = time.perf_counter() - t0
            same = bool(np.array_equal(rf.get("image"), r.get("image")) and rf.get("start_frame") == r.get("start_frame"))
            out["faithful_loops"] = {"value": round(x.shape[0] / dtf / 1e6, 4), "seconds": round(dtf, 2),
                                     "value_with_reference_sleeps": round(x.shape[0] / (dtf + 7.0) / 1e6, 4), "same_result_as_vectorised": same}
    return out


def _oracle():
    """oracle/ is test infrastructure: the CPU baseline above and the in-run checks of the companions (c5's members, the file-to-file
    pixels) reach it through this one function, which only this file hands out."""
    from oracle import wefax_oracle as wo
    return wo


benchlib.ORACLE = _oracle
benchlib.CPU_BASELINE = cpu_baseline


def main():
    args = parse()
    if args.rccl_probe:
        sys.exit(rccl_probe_main())
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(spawn_ranks(args))
    rk = Ranks(args)
    guard = None
    try:
        if args.workload == "iq":
            secs = 40.0 if args.short else float(args.iq_seconds)
            c4 = bench_iq(args, rk, secs, args.steps, args.warmup, not args.no_cpu)
            line = {"metric": "Msamples/s demod->pixel", "value": c4["value"], "unit": "Msamples/s", "n_gpus": rk.world, "steps": args.steps,
                    "warmup": args.warmup, "ms_per_step": c4["ms_per_step"], "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
                    "dtype": c4["dtype"], "data": "synthetic", "config": {k: c4[k] for k in ("workload", "form", "front_end", "ranks_rccl", "start_frame", "image")},
                    "roofline": c4["roofline"], "cpu_baseline": c4.get("cpu_baseline"), "kernels": c4["kernels"],
                    "one_gpu_ms": c4.get("one_gpu_ms"), "speedup_vs_one_gpu": c4.get("speedup_vs_one_gpu"),
                    "efficiency_vs_one_gpu": c4.get("efficiency_vs_one_gpu"), "transport": c4.get("transport"), "wire": c4.get("wire"),
                    "per_step": c4.get("per_step"), "gpu_state": c4.get("gpu_state"), "model_ms": c4.get("model_ms"),
                    "measured_ms": c4.get("measured_ms"), "forced_dist": c4.get("forced_dist"), "forced_fmm": c4.get("forced_fmm"), "placement": c4.get("placement")}
        elif args.workload == "c3":
            line = bench_c3(args, rk)
        else:
            line = bench_c2(args, rk)
            if rk.world == 1 and not args.shard and args.batch == 1 and not args.short and not args.no_extras:
                # driver-timed companions of the headline: other capture lengths, and BASELINE configs[2] with its own roofline / CPU leg
                from wefax_amd import synth
                line["general_length"] = bench_general_lengths(args, rk, synth.config_c2(noise=args.noise, seed=rk.rank))
                a3 = argparse.Namespace(**vars(args))
                a3.steps, a3.warmup = max(3, min(args.steps, 10)), 2
                c3 = bench_c3(a3, rk)
                line["c3"] = {k: c3[k] for k in ("value", "unit", "ms_per_step", "steps", "config", "roofline", "cpu_baseline", "kernels", "general_length", "e2e") if k in c3}
                try:
                    line["fmm"] = bench_fmm(args, rk, synth.config_c2(noise=args.noise, seed=rk.rank))
                except Exception as e:      # noqa: BLE001
                    line["fmm"] = {"error": f"{type(e).__name__}: {e}"[:300]}
                if not args.no_c5:
                    try:
                        line["c5"] = bench_c5(args, rk)
                    except Exception as e:      # noqa: BLE001
                        line["c5"] = {"error": f"{type(e).__name__}: {e}"[:300]}
            if not args.no_c4 and not args.shard and args.batch == 1:
                rk.barrier()
                secs = 40.0 if args.short else float(args.iq_seconds)
                # (with its own CPU leg at N = 1: the oracle's reference-faithful path on a 30-s clip of the same stream format, ~15 s)
                # At N > 1 this is the one part of the line that runs data-path collectives: a rank that fails inside it leaves
                # its peers waiting in an exchange.  The headline above is already measured -- it must not be lost to that -- so a
                # guard prints the line with an `error` in place of the object when no result arrives in time, and ends the rank.
                guard = _LineGuard(line, rk, float(os.environ.get("WFX_BENCH_C4_TIMEOUT", "240"))) if rk.world > 1 else None
                try:
                    if os.environ.get("WFX_BENCH_TEST_FAIL_RANK") == str(rk.rank):      # exercises the guard (tools/, tests)
                        raise RuntimeError("injected failure on this rank")
                    line["c4_strong"] = bench_iq(args, rk, secs, min(args.steps, 10), 2, not args.no_cpu)
                    if rk.world > 1:
                        line["c2_strong"] = bench_c2_strong(args, rk)
                except Exception as e:      # noqa: BLE001 -- reported in the line, the headline stands
                    if rk.world == 1:
                        raise
                    line["c4_strong"] = {"error": f"{type(e).__name__}: {e}"[:400]}
        if rk.rccl_probe is not None:
            line["rccl_probe"] = rk.rccl_probe
        if rk.rank == 0 and isinstance(line.get("roofline"), dict):
            # the driver's record keeps `config`, `roofline` and `cpu_baseline` as objects and only the names of the rest: one compact row per
            # BASELINE config -- and the file-to-file totals -- ride inside `roofline`
            def row(o):
                r = o.get("roofline") or {}
                return {"ms_per_step": o.get("ms_per_step"), "Msamples_s": o.get("value"), "kernel": r.get("kernel"), "frac": r.get("frac"),
                        "whole_path_frac": r.get("whole_path_frac"), "traffic_ratio": r.get("traffic_ratio_whole_path")}
            cfgs = {"c2": row(line)}
            for key, name in (("c3", "c3"), ("c4_strong", "c4")):
                if isinstance(line.get(key), dict) and "error" not in line[key]:
                    cfgs[name] = row(line[key])
            if isinstance(line.get("c5"), dict) and "error" not in line["c5"]:
                cfgs["c5"] = {"Msamples_s": line["c5"]["value"], "seconds_for_64": line["c5"]["seconds_for_64"], "checked_equal": line["c5"]["checked_equal"]}
            if isinstance(line.get("fmm"), dict) and "error" not in line["fmm"]:
                f = line["fmm"]
                cfgs["c2_routes"] = {"default": line["config"].get("hilbert"), "multipole_ms": f["ms_per_step"], "transform_ms": f.get("transform_route_ms_per_step"),
                                     "notch_hilbert_envelope_median_us": f["notch_hilbert_envelope_median_us"],
                                     "f64_frac_of_the_four_kernels": f["roofline"]["frac"], "same_stream": f["stream_and_start_frame_equal_to_transform_route"]}
            e2e = {}
            if isinstance(line.get("e2e"), dict):
                e2e["c2_ms"] = line["e2e"].get("ms")
            if isinstance(line.get("c3"), dict) and isinstance(line["c3"].get("e2e"), dict):
                e2e["c3_ms"] = line["c3"]["e2e"].get("ms")
            if isinstance(line.get("c4_strong"), dict) and isinstance(line["c4_strong"].get("gpu_state"), dict):
                cfgs.setdefault("c4", {})["gpu_state_during"] = line["c4_strong"]["gpu_state"].get("during_steps")
                cfgs["c4"]["ingest_us"] = (line["c4_strong"].get("per_step") or {}).get("ingest_us")
            if rk.world > 1 and isinstance(line.get("c4_strong"), dict) and "error" not in line["c4_strong"]:
                c4s = line["c4_strong"]
                cfgs.setdefault("c4", {})["plans_ms"] = {"auto": {"plan": (c4s.get("wire") or {}).get("layout"), "ms": c4s.get("ms_per_step"), "model_ms": c4s.get("model_ms")},
                                                          **{k: {"ms": (c4s.get("forced_" + k) or {}).get("ms_per_step"), "model_ms": (c4s.get("forced_" + k) or {}).get("model_ms"),
                                                                 "error": (c4s.get("forced_" + k) or {}).get("error")} for k in ("dist", "fmm")},
                                                          "one_gpu_ms": c4s.get("one_gpu_ms")}
            line["roofline"]["configs"] = cfgs
            line["roofline"]["file_to_file_ms"] = e2e
        if guard is None or guard.claim():
            if rk.rank == 0:
                print(json.dumps(line), flush=True)
        try:
            rk.barrier()
        except Exception:      # noqa: BLE001 -- a peer that failed inside the sharded part has left; the line is out
            if guard is None:
                raise
    finally:
        if guard is not None:
            guard.printed_exit_only()
        rk.close()


if __name__ == "__main__":
    main()
