#!/usr/bin/env python3
"""Throughput of the live path's packet decode (wfx_packet_process one by one, wfx_packets_process back to back) and of
the detectors, on one-second packets of a synthetic transmission; the oracle on the same host beside it.
    python tools/packet_bench.py [--packets 64]"""
import argparse, json, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from wefax_amd import _native as nat, synth  # noqa: E402
from wefax_amd.packet import DataPacket, process_packets, _process  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--packets", type=int, default=64)
ap.add_argument("--no-cpu", action="store_true")
args = ap.parse_args()
x = synth.config_c2(noise=0.02, seed=1)[:args.packets * 11025].reshape(args.packets, 11025)
ctx = nat.Context(0)
process_packets(ctx, 11025, x[:2])
t0 = time.perf_counter(); one = [_process(ctx, 11025, p)[0] for p in x]; t1 = time.perf_counter()
many = process_packets(ctx, 11025, x); t2 = time.perf_counter()
assert all(np.array_equal(a, b) for a, b in zip(one, many))
pk = [DataPacket(11025, p, 120, "/tmp/", 1, i, ctx=ctx) for i, p in enumerate(x[:16])]
t3 = time.perf_counter(); det = [(p.contain_start_tone(), p.contain_stop_tone(), p.find_sync_pulse()["pulse_found"]) for p in pk]; t4 = time.perf_counter()
out = {"packets": args.packets, "samples_per_packet": 11025, "one_by_one_us_per_packet": (t1 - t0) / args.packets * 1e6,
       "back_to_back_us_per_packet": (t2 - t1) / args.packets * 1e6, "detectors_us_per_packet": (t4 - t3) / 16 * 1e6}
if not args.no_cpu:
    from oracle import wefax_oracle as wo
    t5 = time.perf_counter(); ref = [wo.process_packet(p, 11025)["samples"] for p in x[:16]]; t6 = time.perf_counter()
    out["oracle_us_per_packet"] = (t6 - t5) / 16 * 1e6
    out["identical_to_oracle"] = bool(all(np.array_equal(a, b) for a, b in zip(ref, many[:16])))
print(json.dumps(out))
