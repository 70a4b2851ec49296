"""One function per BASELINE config and companion measurement; each returns the object bench.py puts into its JSON line."""
from __future__ import annotations

import json
import os
import subprocess
import sys
import time

import benchlib
from benchlib import REPO, HBM_PEAK_GBS, IQ_FS
from benchlib.ranks import Ranks
from benchlib.measure import dist_of, kernel_table, profile_pass, roofline_of
from benchlib.gpustate import GpuState
from benchlib.e2e import bench_e2e


def cpu_baseline(*a, **k):
    """The CPU leg lives in bench.py (the one file that reaches oracle/): called through the hook it sets."""
    return benchlib.CPU_BASELINE(*a, **k)


# ---- the fast multipole route (round 6): a6 + a7 without a transform over the capture ----------------------------------------------
F64_PEAK_TFLOPS = 78.6          # MI355X float64, vector = matrix (tools/micro/mfma_f64_rate.hip measures 76-77 with either or any mix of the two)


FMM_FMA_PER_SAMPLE = 307        # near field 82, P2M 32, L2P 32, M2L 56, M2M + L2L 38, moments <-> nodes <-> coefficients 18, notch 49


# float64 multiply-adds per sample of the multipole route's kernel groups: notch 49 + P2M 32 + M2M 19 + moments -> nodes 9 |
# the levels above the leaf workgroups (1 / 64 of a level each) | M2L 56 + L2L 19 + nodes -> coefficients 9 + near field 82 + L2P 32
FMM_GROUP_FMA = {"fmm_notch_p2m_m2m": 109, "fmm_tiers_and_top": 2, "fmm_tree_levels": 84, "fmm_near_l2p_env_median": 114}


def fmm_roofline(hbm_view: dict, prof: dict, steps: int, n: int) -> dict:
    """The headline's roofline when the decode runs the multipole route: its dominant kernel (tree levels + near field + L2P + envelope + median
    in one launch) is bound by the float64 unit -- the matrix and the vector pipe share ONE (tools/micro/mfma_f64_rate.hip: 76-77 TFLOP/s with
    either or any mix) -- so achieved = that kernel's multiply-adds x 2 / its average launch time against 78.6 TFLOP/s.  The HBM view of the
    same kernel (algorithmic bytes of the step / its launch time) stays beside it."""
    fam = {k: v for k, v in prof.items() if k in FMM_GROUP_FMA}
    if not fam:
        return hbm_view
    dom = max(fam.items(), key=lambda kv: kv[1][1])
    fma = FMM_GROUP_FMA[dom[0]] + (FMM_GROUP_FMA["fmm_tree_levels"] if dom[0] == "fmm_near_l2p_env_median" and "fmm_tree_levels" not in prof else 0)
    avg_s = dom[1][1] / dom[1][0] / 1e3
    flops = 2.0 * fma * n
    return {"bound": "mfma", "kernel": dom[0], "achieved": round(flops / avg_s / 1e12, 2), "peak": F64_PEAK_TFLOPS, "unit": "TFLOP/s",
            "frac": round(flops / avg_s / 1e12 / F64_PEAK_TFLOPS, 4), "algorithmic_flops_per_launch": int(flops), "fma_per_sample": fma,
            "avg_launch_us": round(avg_s * 1e6, 2), "launches_per_step": round(dom[1][0] / steps, 2),
            "traffic": hbm_view.get("traffic") if hbm_view.get("kernel") == dom[0] else None, "traffic_source": hbm_view.get("traffic_source"),
            "traffic_ratio_whole_path": hbm_view.get("traffic_ratio_whole_path"), "whole_path_frac": hbm_view.get("whole_path_frac"),
            "hbm_view": {k: hbm_view.get(k) for k in ("kernel", "achieved", "peak", "unit", "frac", "algorithmic_bytes_per_launch")},
            "note": "float64 multiply-adds of the dominant launch against the float64 peak (matrix = vector pipe); hbm_view: the step's algorithmic bytes over the same launch"}


def bench_fmm(args, rk: Ranks, x) -> dict:
    """The same capture with hilbert_mode = WFX_HILBERT_FMM: notch inside the P2M kernel, near field + tree levels on the f64 matrix cores,
    envelope + median + histogram inside the leaf kernel.  Compute-bound: its roof is the float64 rate, reported beside the HBM fraction."""
    import numpy as np
    from wefax_amd.wefax import DecodeJob
    nat, ctx = rk.nat, rk.ctx
    job = DecodeJob(ctx, x, 11025, 120, hilbert_mode=nat.WFX_HILBERT_FMM)
    dt = rk.timed(job.run, args.steps, args.warmup)
    ms = 1e3 * dt / args.steps
    info = job.result()
    prof = profile_pass(ctx, job.run, args.steps)
    dig = job.fetch("digitalized")
    cref = nat.Context(rk.device)                      # (a context decodes the capture it was handed last: the comparison runs on another one)
    ref = DecodeJob(cref, x, 11025, 120, hilbert_mode=nat.WFX_HILBERT_FFT)
    for _ in range(args.warmup + 1):
        ref.run()
    cref.sync()
    t_ref = time.perf_counter()
    for _ in range(args.steps):
        ref.run()
    cref.sync()
    ms_ref = 1e3 * (time.perf_counter() - t_ref) / args.steps
    rinfo = ref.result()
    same = bool(np.array_equal(dig, ref.fetch("digitalized")) and info.start_frame == rinfo.start_frame)
    del ref
    cref.close()
    groups = {"notch_p2m_m2m": "fmm_notch_p2m_m2m", "tiers_and_top": "fmm_tiers_and_top", "tree_levels": "fmm_tree_levels", "near_l2p_env_median": "fmm_near_l2p_env_median"}
    # (since the tree levels run inside the leaf kernel, "tree_levels" is absent and "near_l2p_env_median" is tree + near field + L2P + envelope + median)
    us = {k: round(1e3 * prof[v][1] / args.steps, 1) for k, v in groups.items() if v in prof}
    t_fmm = sum(us.values()) * 1e-6
    flops = 2.0 * FMM_FMA_PER_SAMPLE * x.shape[0]
    n = x.shape[0]
    traffic = None
    pmc = os.path.join(REPO, "profiles", "pmc_traffic_fmm.json")
    if os.path.exists(pmc):
        try:
            tj = json.load(open(pmc))
            once = tj.get("fmm_notch_p2m_m2m", tj.get("fmm_tree_levels", {})).get("launches_seen", [1])[0]          # a kernel that runs once per decode
            traffic = {"bytes_per_decode": int(sum(v["hbm_bytes_per_launch"] * (v["launches_seen"][0] / max(1, once))
                                                   for k, v in tj.items() if k.startswith("fmm_") and isinstance(v, dict))),
                       "source": os.path.relpath(pmc, REPO) + " (rocprofv3 --pmc FETCH_SIZE x2 + WRITE_SIZE, separate passes; " + str(tj.get("_collected_at", "")) + ")"}
        except Exception:      # noqa: BLE001
            traffic = None
    return {"what": "BASELINE configs[1] with a6 + a7 by the fast multipole form (csrc/wfx_fmm.hip; Demodulator(hilbert_mode=4) / WEFAX_HILBERT=fmm)",
            "ms_per_step": round(ms, 4), "value": round(n / (ms / 1e3) / 1e6, 2), "unit": "Msamples/s", "stream_and_start_frame_equal_to_transform_route": same,
            "transform_route_ms_per_step": round(ms_ref, 4),
            "notch_hilbert_envelope_median_us": round(1e6 * t_fmm, 1), "kernel_groups_us": us,
            "roofline": {"bound": "f64", "achieved": round(flops / t_fmm / 1e12, 2) if t_fmm else None, "peak": F64_PEAK_TFLOPS, "unit": "TFLOP/s",
                         "frac": round(flops / t_fmm / 1e12 / F64_PEAK_TFLOPS, 4) if t_fmm else None,
                         "algorithmic_flops": int(flops), "fma_per_sample": FMM_FMA_PER_SAMPLE,
                         "hbm_frac_of_these_kernels": round((2 * n + 4 * n) / t_fmm / 1e9 / HBM_PEAK_GBS, 4) if t_fmm else None,
                         "traffic": traffic}}


# ---- BASELINE configs[4]: 64 independent captures, 8 contexts (= one capture per GPU stream, 8 per GPU) -----------------------------
def bench_c5(args, rk: Ranks) -> dict:
    """64 mixed captures (120 / 240 LPM, IOC576 / 288, seeds 0..63: synth.config_c5_member) decoded eight at a time on eight contexts of this GPU
    by eight host threads; every member's stream is hashed, members 0, 6, 12, ... (the ones tests/test_gpu_configs.py checks in full) against
    the oracle's where the CPU leg is on."""
    import hashlib
    import threading
    import numpy as np
    from wefax_amd import synth
    from wefax_amd.wefax import DecodeJob
    nat = rk.nat
    ctxs = [nat.Context(rk.device) for _ in range(8)]
    from concurrent.futures import ThreadPoolExecutor
    with ThreadPoolExecutor(max_workers=min(16, os.cpu_count() or 4)) as ex:      # (0.7 s of NumPy per member on one core)
        members = list(ex.map(lambda i: synth.config_c5_member(i, noise=args.noise), range(64)))
    total = sum(m[0].shape[0] for m in members)

    def run_round(js):
        ths = [threading.Thread(target=lambda j=j: (j.run(), j.result())) for j in js]
        for t in ths:
            t.start()
        for t in ths:
            t.join()

    digests, checked, ok = {}, [], True
    t_all = 0.0
    for rnd in range(8):
        js = [DecodeJob(ctxs[k], members[8 * rnd + k][0], 11025, members[8 * rnd + k][1]) for k in range(8)]
        run_round(js)                                   # (uploads and first-use plans outside the timed region)
        t0 = time.perf_counter()
        run_round(js)
        t_all += time.perf_counter() - t0
        for k, j in enumerate(js):
            i = 8 * rnd + k
            st = j.fetch("digitalized")
            digests[i] = hashlib.sha256(st.tobytes()).hexdigest()[:16]
            if i % 6 == 0 and not args.no_cpu and i < 24:
                wo = benchlib.ORACLE()
                import tempfile
                with tempfile.TemporaryDirectory() as td:
                    pth = os.path.join(td, "m.wav")
                    synth.write_wav(pth, 11025, members[i][0])
                    r = wo.process(pth, members[i][1], want_messages=False)
                good = bool(np.array_equal(st, r["digitalized"]) and j.result().start_frame == r["start_frame"])
                checked.append(i)
                ok &= good
    for c in ctxs:
        c.close()
    h = hashlib.sha256("".join(digests[i] for i in range(64)).encode()).hexdigest()[:16]
    return {"workload": "BASELINE configs[4]: 64 independent 11.025 kHz captures (mixed 120 / 240 LPM, IOC576 / 288, seeds 0..63), 8 contexts of ONE GPU, 8 at a time",
            "value": round(total / t_all / 1e6, 2), "unit": "Msamples/s", "seconds_for_64": round(t_all, 4), "samples": int(total), "contexts": 8,
            "stream_digest_of_all_64": h, "members_checked_against_the_oracle": checked, "checked_equal": ok if checked else None,
            "note": "captures resident in HBM when a round's clock starts; eight host threads enqueue and wait; no collective (replicas only)"}


# ---- BASELINE configs[1] as ONE capture over all ranks: plan 3 (chunk-local multipole, KBs on the wire) beside plan 1 (distributed transforms) ----
def bench_c2_strong(args, rk: Ranks) -> dict:
    """Strong scaling of the 10-minute capture, both sharded plans in the same run (N > 1 only).  Unmeasured on hardware until a multi-GPU box
    runs it: the plans' own cost-model projections ride beside the measured times."""
    import numpy as np
    from wefax_amd import sharded, synth
    nat, ctx = rk.nat, rk.ctx
    x = synth.config_c2(noise=args.noise, seed=0)
    out = {"workload": f"ONE synthetic 10-min 11.025 kHz capture (BASELINE configs[1], {x.shape[0]} samples) decoded by {rk.world} ranks together",
           "n_gpus": rk.world, "transport": rk.transport_name(), "scaling": "strong"}
    steps = max(3, min(args.steps, 10))
    for plan in ("fmm", "dist"):
        try:
            dec = sharded.ShardedDecoder(ctx, rk.comm, x.shape[0], 11025, 120, nat.WFX_IN_I16_MONO, data=x, plan=plan)
            dt = rk.timed(dec.run, steps, 2)
            ms = 1e3 * dt / steps
            info = dec.result()
            w = wire_object(rk, dec.params, dec.layout, dec.run, ctx.sync)
            sent = [e for e in w["this_rank"]]
            out[plan] = {"ms_per_step": round(ms, 4), "value": round(x.shape[0] / (ms / 1e3) / 1e6, 2), "unit": "Msamples/s",
                         "start_frame": int(info.start_frame) if rk.rank == 0 else None, "layout": w["layout"],
                         "bytes_this_rank_sent": int(w["this_rank_sent"]),
                         "bytes_this_rank_sent_without_the_stream_gather": int(sum(e["sent"] for e in sent if e["name"] != "stream gather")),
                         "collectives_us": w["this_rank_us"], "model_ms": w["model"]["dist_ms"], "model_one_gpu_ms": w["model"]["one_gpu_ms"],
                         "per_collective": [{k: e.get(k) for k in ("name", "sent", "received", "us", "wait_us", "link_GBs")} for e in sent]}
            dec.close()
        except Exception as e:      # noqa: BLE001
            out[plan] = {"error": f"{type(e).__name__}: {e}"[:300]}
    return out


# ---- BASELINE configs[3]: the oversampled IQ stream, all ranks on ONE capture ---------------------------------------
def iq_recipe(seconds: float):
    if seconds < 30:
        raise SystemExit("--iq-seconds must be at least 30")
    return dict(start_tone_s=5.0, phasing_lines=60, image_lines=int((seconds - 15.0) / 0.5) - 60, stop_tone_s=5.0, black_tail_s=5.0)


def wire_object(rk: Ranks, params, layout, run, sync) -> dict:
    """What one sharded decode puts on the wire: the plan's collectives with their bytes (host-only: every rank's exchange lists),
    what THIS rank's communicator counted AND TIMED during one decode -- per collective: `us` on the stream it ran on (HIP-event pair;
    host clock on the blocking transports), `wait_us` the compute stream stood still for it, `hidden_us` = the rest (only an exchange
    on the communicator's own stream can hide anything), `link_GBs` = its largest message / us -- and the cost model's figures
    behind the choice of plan, so that ONE run on real links calibrates the model."""
    nat = rk.nat
    plan = nat.shard_wire_plan(params, rk.world)
    rk.comm.wire_timing(True)
    run()
    sync()
    mine = rk.comm.wire_stats()
    times = rk.comm.wire_times()
    rk.comm.wire_timing(False)
    for e, t in zip(mine, times):
        e.update(t)
        e["link_GBs"] = round(e["largest_message"] / (t["us"] * 1e-6) / 1e9, 2) if (t["us"] and e["largest_message"]) else None
    tot = lambda key: round(sum(e.get(key) or 0.0 for e in mine), 1)       # noqa: E731
    return {"layout": {0: "single (rank 0 alone)", 1: "rows", 2: "columns", 3: "chunk-local multipole (plan 3)"}[int(layout.plan)], "chosen_by": "caller" if layout.plan_forced else "cost model",
            "reason": layout.plan_reason.decode(), "total_bytes": sum(e["bytes"] for e in plan),
            "array_transposes": sum(1 for e in plan if " E" in e["name"]),
            "per_collective": plan, "this_rank_sent": sum(e["sent"] for e in mine), "this_rank": mine,
            "this_rank_us": {"collectives": tot("us"), "compute_stream_waited": tot("wait_us"), "hidden": tot("hidden_us"),
                             "clock": sorted({str(e.get("clock")) for e in mine})},
            "model": {"link_GBs": float(os.environ.get("WFX_LINK_GBS", "50")), "latency_us": float(os.environ.get("WFX_LINK_LAT_US", "20")),
                      "one_gpu_ms": round(1e3 * layout.model_single_s, 3),
                      "dist_compute_ms": round(1e3 * layout.model_dist_compute_s, 3), "dist_wire_ms": round(1e3 * layout.model_dist_wire_s, 3),
                      "dist_ms": round(1e3 * (layout.model_dist_compute_s + layout.model_dist_wire_s), 3),
                      "wire_bytes": int(layout.model_wire_bytes)}}


def bench_iq(args, rk: Ranks, seconds: float, steps: int, warmup: int, with_cpu: bool) -> dict:
    """Strong scaling: the stream is fixed, every rank owns 1/world of it plus the FIR chain's halo."""
    from wefax_amd import polyphase, sharded, synth_device
    nat, ctx = rk.nat, rk.ctx
    kw = iq_recipe(seconds)
    sp = synth_device.synth_params(float(IQ_FS), noise=args.noise, seed=0, iq=True, **kw)
    n0 = int(ctx.lib.wfx_synth_frames(sp))
    fe = polyphase.FrontEnd(IQ_FS, stop_rate=args.iq_stop_rate)
    raw_loader = synth_device.SliceLoader(ctx, sp)
    # where a capture's pages lie is worth up to 15 % to the ingest's ~770 streams (docs/history/EXPERIMENTS_rounds1-5.md 9.2): the capture buffer is the
    # best of a few allocations (Context.dev_malloc_placed; every candidate's rate is in `placement`).  WFX_PLACE_TRIES=1 takes the first.
    os.environ.setdefault("WFX_PLACE_TRIES", "4")
    del synth_device.PLACEMENTS[:]
    t_syn = time.perf_counter()
    fused = rk.world == 1 and (args.iq_form == "fused" or (args.iq_form == "auto" and not rk.use_rccl))
    if fused:
        dec = sharded.FrontEndExactDecoder(ctx, fe, None, n_in_total=n0, in_kind=nat.WFX_IN_I16_STEREO, lines_per_minute=120, raw_loader=raw_loader)
        n = dec.n
        own_in, own_out = n0, n
        ia, ib = dec.chain[0][2]
    else:
        dec = sharded.FrontEndShardedDecoder(ctx, rk.comm, fe, None, n_in_total=n0, in_kind=nat.WFX_IN_I16_STEREO, lines_per_minute=120,
                                             raw_loader=raw_loader, plan=args.plan)
        n = dec.n
        own_in, own_out = dec.raw_frames, dec.layout.own_samples
    ctx.sync()
    t_syn = time.perf_counter() - t_syn
    state_before = GpuState.read()
    # The shader clock of an idle MI355X takes ~10 decodes (40-80 ms of work) to come up -- consecutive launches of the ingest kernel
    # right after an idle second read 5.4, 4.2, 4.0, 3.8, 3.7 ... 3.5 ms (tools/ingest_lab.py) -- so this object warms up by TIME:
    # untimed decodes until 0.3 s have passed (at least `warmup`), the count is printed as `warmup_steps`
    wsteps, t_w = max(warmup, 1), time.perf_counter()
    for _ in range(wsteps):
        dec.run()
    ctx.sync()
    spent = time.perf_counter() - t_w
    import numpy as np
    extra = int(min(200, np.ceil(max(0.0, float(os.environ.get("WFX_BENCH_C4_WARM_S", "0.3")) - spent) / max(spent / wsteps, 1e-4))))
    if rk.world > 1:       # every rank runs the same number of decodes (they carry collectives): rank 0's count
        extra = int(rk.comm.allgather(ctx, np.array([extra], dtype=np.int64))[0][0])
    for _ in range(extra):
        dec.run()
    ctx.sync()
    wsteps += extra
    dt = rk.timed(dec.run, steps, 0)
    ms = 1e3 * dt / steps
    # step by step (after the timed region, same buffers): wall time of every decode and the HIP-event time of its ingest launch, with
    # the clocks / power / temperature sampled meanwhile -- a slow box shows in the clocks, a slow kernel in the distribution
    step_ms, ingest_us = [], []
    with GpuState.sample() as smp:
        for _ in range(max(steps, 10)):
            ctx.profile_reset()
            ctx.profile_enable(True)
            t0 = time.perf_counter()
            dec.run()
            ctx.sync()
            step_ms.append(1e3 * (time.perf_counter() - t0))
            ctx.profile_enable(False)
            pr = ctx.profile()
            if "polyphase_ingest" in pr:
                ingest_us.append(1e3 * pr["polyphase_ingest"][1])
    prof = profile_pass(ctx, dec.run, 1)
    info = dec.result()
    alg_bytes = own_in * 4 + 4 * own_out                     # N0 * B_in + 4 N, this rank's share
    pmc = os.path.join(REPO, "profiles", "pmc_traffic_iq.json") if (rk.world == 1 and seconds == 3600.0) else None
    out = {"workload": f"ONE synthetic 1.536 MS/s int16 IQ stream of {seconds:.0f} s (BASELINE configs[3]): {n0} IQ frames -> {n} samples at "
                       f"11 025 Hz, 120 LPM, AWGN sigma {args.noise} FS, synthesised in HBM",
           "n_gpus": rk.world, "ranks_rccl": rk.world if rk.comm.is_rccl else 0, "transport": rk.transport_name(), "scaling": "strong",
           "form": ("front end + fused exact decode on one GPU" if fused else
                    ("rank 0 alone behind the sharded interface (the cost model declined the distributed plan)" if dec.layout.plan == 0 else
                     f"front end on each rank's 1/{rk.world} of the stream + chunk-local exact path (plan 3: resampler and Hilbert transform by their multipole forms on the "
                     "rank's arc; kilobytes of weights, 320 samples per seam, 2 histogram all-reduces, 1 candidate all-gather, 1 stream gather per decode)"
                     if dec.layout.plan == 3 else
                     f"front end on each rank's 1/{rk.world} of the stream + sharded exact path (distributed FFT resample and Hilbert, "
                     f"{'columns layout: 4' if dec.layout.plan == 2 else 'rows layout: 8'} array transposes, 2 histogram all-reduces, 1 candidate all-gather, "
                     "1 stream gather per decode)")),
           "front_end": fe.describe() + f" -> exact FFT resample {fe.out_rate} -> 11025 Hz",
           "ms_per_step": round(ms, 4), "value": round(n0 / (ms / 1e3) / 1e6, 2), "unit": "Msamples/s", "steps": steps, "warmup_steps": wsteps,
           "synthesis_s": round(t_syn, 2), "dtype": "i16 integer-exact ingest / f64 everywhere behind it",
           "start_frame": int(info.start_frame) if rk.rank == 0 else None, "image": [int(info.width), 4 * int(info.height)] if rk.rank == 0 else None,
           "roofline": roofline_of(prof, 1, alg_bytes, ms, pmc, merge_fft=True), "kernels": kernel_table(prof, 1),
           "per_step": {"decode_ms_profiled": dist_of(step_ms), "ingest_us": dist_of(ingest_us),
                        "note": "one decode at a time, device synchronised after each, HIP-event pairs on (adds ~0.1 ms per decode): the spread, not the level"},
           "gpu_state": {"before": state_before, "during_steps": smp.summary(), "after": GpuState.read(),
                         "source": "amdgpu sysfs (pp_dpm_*clk, hwmon), sampled every 20 ms while the per-step pass ran"}}
    if getattr(getattr(dec, "fe", None), "fused_ingest", False):
        out["front_end"] += " [stages 1+2 in one streaming kernel, csrc/wfx_ingest.hip]"
    if fused:
        p_raw, n_raw = dec.fe.p_raw, dec.fe.n_raw
        try:
            out["placement"] = {"tries": int(os.environ["WFX_PLACE_TRIES"]), "candidates_stream_GBs": [[round(v, 1) for v in r] for r in synth_device.PLACEMENTS],
                                "output_candidates_stream_GBs": [round(v, 1) for v in getattr(dec.fe, "placement_out", [])],
                                "kept_stream_GBs": round(ctx.d_stream_rate(p_raw, n_raw * 4, out_ptr=dec.fe.p_out), 1), "kept_plain_read_GBs": round(ctx.d_read_rate(p_raw, n_raw * 4, 2), 1),
                                "note": "stream = input bytes per second of the ingest kernel itself on that allocation, outputs into a scratch buffer (wfx_d_stream_rate); plain = a dense sweep"}
        except Exception as e:      # noqa: BLE001
            out["placement"] = {"error": f"{type(e).__name__}: {e}"[:200]}
    if not fused:
        out["wire"] = wire_object(rk, dec.dec.params, dec.layout, dec.run, ctx.sync)
        m = out["wire"]["model"]
        out["model_ms"] = m["one_gpu_ms"] if dec.layout.plan == 0 else m["dist_ms"]
        out["measured_ms"] = out["ms_per_step"]
        if getattr(dec, "plan_choice", None):
            # an oversampled capture left open by the caller: the choice was made with the front end's time added (sharded.choose_plan_with_front_end)
            pc = dec.plan_choice
            out["plan_choice"] = {"front_end_ms": round(1e3 * pc["front_end_s"], 3), "chosen": pc["chosen"],
                                  "candidates_model_ms": {k: {"plan": v["plan"], "ms": round(1e3 * v["model_s"], 3)} for k, v in pc["candidates"].items()}}
            out["model_ms"] = round(1e3 * pc["candidates"][pc["chosen"]]["model_s"], 3)
    dec.close()
    if not fused and rk.world > 1 and args.plan == "auto" and os.environ.get("WFX_BENCH_BOTH_PLANS", "1") != "0":
        # the other sides of the cost model's decision, in the same run: the transposing plan and the chunk-local plan forced (when `auto` chose
        # one of them, this is a second measurement of it).  model_ms beside measured_ms for both is what calibrates WFX_LINK_GBS / WFX_LINK_LAT_US
        for forced in ("dist", "fmm"):
            try:
                d2 = sharded.FrontEndShardedDecoder(ctx, rk.comm, fe, None, n_in_total=n0, in_kind=nat.WFX_IN_I16_STEREO, lines_per_minute=120,
                                                    raw_loader=raw_loader, plan=forced)
                dt2 = rk.timed(d2.run, max(2, steps // 2), 1)
                ms2 = 1e3 * dt2 / max(2, steps // 2)
                w2 = wire_object(rk, d2.dec.params, d2.layout, d2.run, ctx.sync)
                i2 = d2.result()
                out["forced_" + forced] = {"plan": forced, "ms_per_step": round(ms2, 4), "measured_ms": round(ms2, 4), "model_ms": w2["model"]["dist_ms"],
                                           "start_frame": int(i2.start_frame) if rk.rank == 0 else None, "wire": w2,
                                           "kernels": kernel_table(profile_pass(ctx, d2.run, 1), 1)}
                d2.close()
            except Exception as e:      # noqa: BLE001 -- a layout the plan does not take: said, not fatal
                out["forced_" + forced] = {"plan": forced, "error": f"{type(e).__name__}: {e}"[:300]}
    raw_loader.close()
    # the one-GPU time of the same stream, measured in this run on rank 0, and the efficiency against it
    if rk.world > 1:
        one_ms = None
        if rk.rank == 0:
            loader2 = synth_device.SliceLoader(ctx, sp)
            one = sharded.FrontEndExactDecoder(ctx, fe, None, n_in_total=n0, in_kind=nat.WFX_IN_I16_STEREO, lines_per_minute=120, raw_loader=loader2)
            for _ in range(2):
                one.run()
            ctx.sync()
            t0 = time.perf_counter()
            for _ in range(steps):
                one.run()
            ctx.sync()
            one_ms = 1e3 * (time.perf_counter() - t0) / steps
            one.close()
            loader2.close()
        rk.barrier()
        if rk.rank == 0:
            out["one_gpu_ms"] = round(one_ms, 4)
            out["speedup_vs_one_gpu"] = round(one_ms / ms, 3)
            out["efficiency_vs_one_gpu"] = round(one_ms / ms / rk.world, 4)
    else:
        out["one_gpu_ms"] = out["ms_per_step"]
        out["speedup_vs_one_gpu"], out["efficiency_vs_one_gpu"] = 1.0, 1.0
    if with_cpu and rk.rank == 0 and rk.world == 1:
        from wefax_amd import synth
        s_secs = 30.0
        xs = synth.synth_capture(float(IQ_FS), noise=args.noise, seed=0, iq=True, start_tone_s=2.0, phasing_lines=20,
                                 image_lines=int((s_secs - 14.0) / 0.5), stop_tone_s=1.0, black_tail_s=1.0)
        cb = cpu_baseline(xs, IQ_FS, 120, False, f"a self-contained {s_secs:.0f} s capture of the same stream format ({xs.shape[0]} IQ frames), "
                                                 "reference-faithful path (stereo merge + FFT resample), one run, read from a wav file")
        cb.pop("_result")
        out["cpu_baseline"] = cb
    return out


# ---- BASELINE configs[2]: 60 minutes at 48 kHz (exact FFT resample) ------------------------------------------------
def bench_c3(args, rk: Ranks) -> dict:
    import numpy as np
    from wefax_amd import sharded, synth, synth_device
    from wefax_amd.wefax import DecodeJob
    nat, ctx = rk.nat, rk.ctx
    kw = dict(image_lines=7110, black_tail_s=5.0) if not args.short else dict(start_tone_s=5.0, phasing_lines=20, image_lines=20, stop_tone_s=2.0, black_tail_s=3.0)
    sp = synth_device.synth_params(48000.0, noise=args.noise, seed=0, iq=False, **kw)
    n0 = int(ctx.lib.wfx_synth_frames(sp))
    if rk.world == 1 and not rk.use_rccl:
        ptr = synth_device.synth_slice(ctx, sp, 0, n0)
        x = ctx.dev_download(ptr, (n0,), np.int16)                  # through the host once: DecodeJob uploads its own copy
        ctx.dev_free(ptr)
        job = DecodeJob(ctx, x, 48000, 120)
        run, result, n = job.run, job.result, job.n
        own_in, own_out = n0, n
        form = "fused exact decode on one GPU (int16 capture read in place by the resampler's first pass)"
        closer = lambda: None                                       # noqa: E731
    else:
        dec = sharded.ShardedDecoder(ctx, rk.comm, n0, 48000, 120, nat.WFX_IN_I16_MONO, plan=args.plan)
        lay = dec.layout
        if lay.nseg > 1:        # columns layout: the rank's columns of every row, segment by segment into one buffer
            ptr = ctx.dev_malloc(max(2, lay.in_frames * 2))
            for sgm in range(int(lay.nseg)):
                a = int(lay.in_lo) + sgm * int(lay.in_seg_stride)
                synth_device.synth_into(ctx, sp, ptr + sgm * int(lay.in_seg_len) * 2, a, a + int(lay.in_seg_len))
        else:
            ptr = synth_device.synth_slice(ctx, sp, int(lay.in_lo), int(lay.in_hi)) if lay.in_hi > lay.in_lo else ctx.dev_malloc(64)
        dec.attach(ptr)
        run, result, n = dec.run, dec.result, dec.n
        own_in, own_out = lay.in_frames, lay.own_samples
        form = (f"sharded exact path over {rk.world} rank(s): distributed FFT resample and Hilbert, "
                f"{ {0: 'single plan (rank 0 alone)', 1: 'rows layout', 2: 'columns layout'}[int(lay.plan)] }")
        closer = lambda: (dec.close(), ctx.dev_free(ptr))          # noqa: E731
    dt = rk.timed(run, args.steps, max(args.warmup, 1))
    ms = 1e3 * dt / args.steps
    prof = profile_pass(ctx, run, args.steps)
    info = result()
    alg = own_in * 2 + 4 * own_out
    out = {"metric": "Msamples/s demod->pixel", "value": round(n0 / (ms / 1e3) / 1e6, 2), "unit": "Msamples/s", "n_gpus": rk.world, "steps": args.steps,
           "warmup": args.warmup, "ms_per_step": round(ms, 4), "higher_is_better": True, "scaling": "strong" if rk.world > 1 else "weak",
           "vs_baseline": None, "dtype": "f64", "data": "synthetic",
           "config": {"workload": f"synthetic 60-min 48 kHz WEFAX capture (BASELINE configs[2]): {n0} int16 mono samples -> {n} at 11 025 Hz by the exact "
                                  f"FFT resample (wefax.py:384), 120 LPM, AWGN sigma {args.noise} FS, synthesised in HBM" if not args.short else "SHORT 48 kHz capture",
                      "form": form, "image": [int(info.width), 4 * int(info.height)] if rk.rank == 0 else None,
                      "start_frame": int(info.start_frame) if rk.rank == 0 else None, "ranks_rccl": rk.world if rk.comm.is_rccl else 0},
           "roofline": roofline_of(prof, args.steps, alg, ms, os.path.join(REPO, "profiles", "pmc_traffic_c3.json")),
           "kernels": kernel_table(prof, args.steps), "cpu_baseline": None}
    if not (rk.world == 1 and not rk.use_rccl):
        out["wire"] = wire_object(rk, dec.params, dec.layout, run, ctx.sync)
    closer()
    if rk.world == 1 and not rk.use_rccl and not args.short and not args.no_extras:
        # A recording is as long as it is: the same capture less two samples (n0 even with a prime-ridden half, the reference's output
        # length int(11025 n0 / fs) odd) and less one (n0 odd) -- resampled by two chirp-z transforms on the mixed-radix passes
        # (round 4; rounds 1-3: Bluestein on power-of-two transforms, 41 ms), the Hilbert transform in its odd-length form behind it.
        anyl = {}
        # (trim 6: an EVEN count at 11 025 Hz -- the multipole forms of the resampler and of the Hilbert transform take it, `auto`'s choice for
        # lengths the mixed-radix resampler does not take; trims 2 and 1: odd counts, the chirp-z resampler + the odd-length Hilbert form)
        for trim in (6, 2, 1):
            j2 = DecodeJob(ctx, np.ascontiguousarray(x[:n0 - trim]), 48000, 120)
            for _ in range(2):
                j2.run()
            ctx.sync()
            t0 = time.perf_counter()
            for _ in range(3):
                j2.run()
            ctx.sync()
            t_any = 1e3 * (time.perf_counter() - t0) / 3
            anyl["n0_minus_%d" % trim] = {"n0": int(n0 - trim), "n": int(j2.n), "ms_per_step": round(t_any, 3), "ratio_to_whole_seconds": round(t_any / ms, 2),
                                          "route": "multipole (resampler + Hilbert transform)" if j2.hilbert_mode == nat.WFX_HILBERT_FMM else "transform (chirp-z resampler)"}
            del j2
        out["general_length"] = anyl
    if rk.rank == 0 and rk.world == 1 and not args.no_cpu:
        xs = synth.synth_capture(48000.0, noise=args.noise, seed=0, start_tone_s=5.0, phasing_lines=60, image_lines=1060, stop_tone_s=2.0, black_tail_s=3.0)
        cb = cpu_baseline(xs, 48000, 120, False, f"a 10-minute capture of the same format ({xs.shape[0]} samples: 1/6 of the workload), one run, read from a wav file")
        cb.pop("_result")
        out["cpu_baseline"] = cb
    if rk.rank == 0 and rk.world == 1 and not args.short and not getattr(args, "no_e2e", False):
        xs = None
        if not args.no_cpu:
            xs = synth.synth_capture(48000.0, noise=args.noise, seed=0, start_tone_s=5.0, phasing_lines=60, image_lines=1060, stop_tone_s=2.0, black_tail_s=3.0)
        out["e2e"] = bench_e2e(x, 48000, 120, f"BASELINE configs[2] file to file: a {x.nbytes + 44}-byte wav on tmpfs -> Demodulator -> png on tmpfs", not args.no_cpu,
                               reps=3, cpu_x=xs, cpu_what="a 10-minute 48 kHz wav of the same format (1/6 of the workload)")
    return out


# ---- BASELINE configs[1]: the line itself --------------------------------------------------------------------------
def bench_c2(args, rk: Ranks) -> dict:
    import numpy as np
    from wefax_amd import sharded, synth
    from wefax_amd.wefax import DecodeJob
    nat, ctx = rk.nat, rk.ctx
    if args.short:
        x = synth.synth_capture(11025.0, noise=args.noise, seed=rk.rank, start_tone_s=5.0, phasing_lines=20, image_lines=220, stop_tone_s=2.0, black_tail_s=3.0)
    else:
        x = synth.config_c2(noise=args.noise, seed=0 if args.shard else rk.rank)
    if args.trim:
        x = np.ascontiguousarray(x[:x.shape[0] - args.trim])
    extra = []
    if args.shard:      # ONE capture, all ranks (exercises the sharded exact path on the 10-minute size)
        job = sharded.ShardedDecoder(ctx, rk.comm, x.shape[0], 11025, 120, nat.WFX_IN_I16_MONO, data=x, plan=args.plan)
        n0 = n = job.n
        total = n0
    else:
        job = DecodeJob(ctx, x, 11025, 120)
        n0, n = job.n0, job.n
        for b in range(1, args.batch):                  # BASELINE configs[4] members: own context and stream each
            xb, lpm_b = synth.config_c5_member(rk.rank * args.batch + b, noise=args.noise)
            cb = nat.Context(rk.device)
            extra.append((cb, DecodeJob(cb, xb, 11025, lpm_b)))
        total = (n0 + sum(jb.n0 for _, jb in extra)) * rk.world

    def step():
        job.run()
        for _, jb in extra:
            jb.run()

    def sync_all():
        ctx.sync()
        for cb, _ in extra:
            cb.sync()

    dt = rk.timed(step, args.steps, args.warmup, sync_all)
    ms = 1e3 * dt / args.steps
    info = job.result()
    out = {"metric": "Msamples/s demod->pixel", "value": round(total * args.steps / dt / 1e6, 2), "unit": "Msamples/s", "n_gpus": rk.world,
           "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms, 4), "higher_is_better": True,
           "scaling": "strong" if args.shard else "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
           "config": {"workload": ("synthetic 10-min 11.025 kHz WEFAX capture (BASELINE configs[1]): "
                                   f"{n0} int16 mono samples, 120 LPM, AWGN sigma {args.noise} FS" if not args.short else "SHORT debugging capture"),
                      "captures_per_gpu": args.batch,
                      "hilbert": ("sharded: the plan's forms" if args.shard else
                                  ("multipole route (exact to 1e-13; csrc/wfx_fmm.hip)" if getattr(job, "hilbert_mode", None) == nat.WFX_HILBERT_FMM else "transform route (fft, exact)")),
                      "image": [int(info.width), 4 * int(info.height)] if rk.rank == 0 else None,
                      "start_frame": int(info.start_frame) if rk.rank == 0 else None,
                      "parallelism": (f"ONE capture sharded over {rk.world} rank(s): distributed Hilbert transform, 1 stream gather" if args.shard else
                                      "1 capture per GPU, no data-path collective"),
                      "ranks_rccl": rk.world if rk.comm.is_rccl else 0, "transport": rk.transport_name()}}
    prof = profile_pass(ctx, job.run, args.steps) if (args.shard or rk.rank == 0) else None      # (a sharded decode is a collective: every rank takes part)
    if args.shard:
        out["wire"] = wire_object(rk, job.params, job.layout, job.run, ctx.sync)
    if rk.rank == 0:
        alg_bytes = (n0 * 2 + 4 * n) // (rk.world if args.shard else 1)          # N0 * B_in + 4 N (this rank's share when sharded)
        on_fmm = (not args.shard) and getattr(job, "hilbert_mode", None) == nat.WFX_HILBERT_FMM
        out["roofline"] = roofline_of(prof, args.steps, alg_bytes, ms, os.path.join(REPO, "profiles", "pmc_traffic_fmm.json" if on_fmm else "pmc_traffic.json"))
        if on_fmm:
            out["roofline"] = fmm_roofline(out["roofline"], prof, args.steps, n)
        out["kernels"] = kernel_table(prof, args.steps)
    if rk.rank == 0 and rk.world == 1 and not args.shard and args.batch == 1 and not args.short and not args.no_extras:
        out["cache_state"] = bench_cold(ctx, job, max(5, min(args.steps, 20)))
    # host buffers in -> host image out (PCIe both ways): the capture in pinned host memory, uploaded by DMA, decoded, the image
    # copied back into pinned memory -- every step enqueued, one wait per capture.  Never `value`.
    if rk.rank == 0 and not args.shard and not args.no_pcie:
        xin = nat.pinned_empty(x.shape, x.dtype)
        xin[...] = x
        img_host = nat.pinned_empty((4 * (n // job.width) * job.width,), np.uint8)
        reps = 10
        for r in range(reps + 2):
            if r == 2:
                t1 = time.perf_counter()
            job.reload(xin)
            job.run()
            job.fetch_image_async(img_host)
            job.result()
        serial = n0 * reps / (time.perf_counter() - t1) / 1e6
        info2 = job.result()
        out["pcie_inclusive_image_equal"] = bool(np.array_equal(img_host[:4 * info2.height * info2.width].reshape(4 * info2.height, info2.width), job.fetch("image")))
        # the same with two captures in flight (two contexts = two streams): one capture's copies overlap the other's kernels
        ctx2 = nat.Context(rk.device)
        jobs = [job, DecodeJob(ctx2, x, 11025, 120)]
        outs = [img_host, nat.pinned_empty(img_host.shape, np.uint8)]
        for r in range(2 * reps + 4):
            if r == 4:
                for j in jobs:
                    j.result()
                t1 = time.perf_counter()
            j = jobs[r & 1]
            if r >= 2:
                j.result()                       # its previous capture has left the device
            j.reload(xin)
            j.run()
            j.fetch_image_async(outs[r & 1])
        for j in jobs:
            j.result()
        piped = n0 * 2 * reps / (time.perf_counter() - t1) / 1e6
        out["pcie_inclusive_msamples_s"] = round(max(serial, piped), 1)
        out["pcie_inclusive"] = {"one_capture_at_a_time": round(serial, 1), "two_in_flight": round(piped, 1),
                                 "how": "capture in pinned host memory -> DMA upload -> decode -> DMA of the image into pinned host memory, per capture"}
        ctx2.close()
    out["cpu_baseline"] = None
    if rk.rank == 0 and (rk.world == 1 or args.shard) and not args.no_cpu:        # the CPU leg is reported at N = 1 only; a sharded decode is still CHECKED against it
        cpu = cpu_baseline(x, 11025, 120, (not args.no_cpu_loops) and rk.world == 1, f"the whole capture ({x.shape[0]} samples), one run, read from a wav file")
        ref = cpu.pop("_result")
        img = job.fetch("image")
        stream = job.fetch("stream" if args.shard else "digitalized")
        if rk.world == 1:
            out["cpu_baseline"] = cpu
        out["parity_vs_oracle"] = {"start_frame_equal": bool(ref.get("start_frame") == info.start_frame),
                                   "max_abs_pixel_delta": (int(np.max(np.abs(img.astype(np.int16) - ref["image"].astype(np.int16))))
                                                           if "image" in ref and img.shape == ref["image"].shape else None),
                                   "digitalized_mismatches": int(np.count_nonzero(stream != ref["digitalized"]))}
    if args.shard:
        job.close()
    for cb, _ in extra:
        cb.close()
    if rk.rank == 0 and rk.world == 1 and not args.shard and args.batch == 1 and not args.short and not args.no_extras and not getattr(args, "no_e2e", False) and not args.trim:
        out["e2e"] = bench_e2e(x, 11025, 120, f"BASELINE configs[1] file to file: a {x.nbytes + 44}-byte wav on tmpfs -> Demodulator -> png on tmpfs", not args.no_cpu)
    return out


def bench_cold(ctx, job, steps: int) -> dict:
    """The timed steps of the headline re-decode ONE resident capture, so its 14 MB of samples and part of the 57 MB arrays are
    still in the 256 MiB Infinity Cache when the next step starts.  Here every decode starts behind a 512 MB device-to-device copy
    (1 GB of traffic: nothing of the previous decode is left in the L2s or the Infinity Cache) and is timed on its own with HIP
    events on the library's stream; `warm` is the same per-decode timing without the copy.  Never `value`."""
    nb = 512 << 20
    scratch = ctx.dev_malloc(2 * nb)
    try:
        res = {}
        for name, flush in (("warm", False), ("cold", True)):
            ts = []
            for r in range(steps + 2):
                if flush:
                    ctx.dev_copy(scratch + nb, scratch, nb)
                else:
                    ctx.sync()
                ctx.timer_start()
                job.run()
                ms = ctx.timer_stop()
                if r >= 2:
                    ts.append(ms)
            ts.sort()
            res[name + "_ms"] = round(sum(ts) / len(ts), 4)
            res[name + "_median_ms"] = round(ts[len(ts) // 2], 4)
    finally:
        ctx.dev_free(scratch)
    res["how"] = ("one decode per measurement between HIP events, the stream idle in front of it; cold: behind a 512 MB device-to-device copy "
                  "that leaves nothing of the previous decode in the L2s / Infinity Cache")
    return res


def bench_general_lengths(args, rk: Ranks, x) -> dict:
    """The reference decodes whatever length the wav has (wefax.py:174 calls scipy on it).  The headline length has a 13-smooth
    half (every BASELINE size does) and takes the unpadded transforms; one sample more makes it odd (real samples against scipy's real
    kernel: two packed transforms of M/2 >= N points and a glue pass, round 4), two samples more even with a half that has a large prime factor (packed convolution
    zero-padded to the cheapest 13-smooth M >= N - 1).  Same capture plus 1 / 2 trailing samples, same kernels otherwise."""
    import numpy as np
    from wefax_amd.wefax import DecodeJob
    nat, ctx = rk.nat, rk.ctx
    out = {}
    for extra in (1, 2):
        xe = np.concatenate([x, x[-extra:]])
        job = DecodeJob(ctx, xe, 11025, 120)
        for _ in range(2):
            job.run()
        ctx.sync()
        steps = max(3, min(args.steps, 10))
        t0 = time.perf_counter()
        for _ in range(steps):
            job.run()
        ctx.sync()
        ms = 1e3 * (time.perf_counter() - t0) / steps
        info = job.result()
        n = int(xe.shape[0])
        rec = {"n": n, "ms_per_step": round(ms, 4), "msamples_s": round(n / ms / 1e3, 1), "start_frame": int(info.start_frame)}
        if extra == 2:
            m = nat.padded_length(n - 1)
            rec["form"] = (f"packed convolution of n/2 = {n // 2} points zero-padded to the 13-smooth M = {m} ({nat.plan_describe(m)})" if m else
                           "packed convolution padded to a power of two")
        else:
            mh = nat.padded_length(n)
            rec["form"] = (f"odd length: real samples x real kernel as two PACKED transforms of M/2 = {mh} points ({nat.plan_describe(mh)}) + one glue pass" if mh
                           else f"odd length: unpacked convolution padded to 2^{int(np.ceil(np.log2(2 * n - 1)))}")
        rec["kernels"] = kernel_table(profile_pass(ctx, job.run, steps), steps)
        if not args.no_cpu:
            ref = cpu_baseline(xe, 11025, 120, False, "")["_result"]
            rec["digitalized_mismatches"] = int(np.count_nonzero(job.fetch("digitalized") != ref["digitalized"]))
            rec["start_frame_equal"] = bool(ref.get("start_frame") == info.start_frame)
        out[f"n_plus_{extra}"] = rec
    return out
