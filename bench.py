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

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured copy)
IQ_FS = 1536000


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


# ---- N ranks from one command ---------------------------------------------------------------------------------
def spawn_ranks(args) -> int:
    """`--gpus N` without a launcher: start N fresh processes (this script again, one rank each) BEFORE anything touches a
    GPU, relay rank 0's stdout.  The parent imports nothing but the standard library.  Each child sees every device and
    takes the one its LOCAL_RANK names (RCCL needs its peers' devices visible for P2P / IPC); WFX_BENCH_OVERSUBSCRIBE=1 puts
    several ranks on the devices there are.  A child that dies takes its siblings with it: the parent polls all of them, terminates the rest on
    the first non-zero exit (a rank inside a collective would otherwise wait for ever) and gives up after WFX_BENCH_TIMEOUT s."""
    import secrets
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    with socket.socket() as sk:                       # the unique-id bootstrap gets a port of its own, reserved here
        sk.bind(("127.0.0.1", 0))
        boot = sk.getsockname()[1]
    nonce = secrets.token_hex(8)
    over = os.environ.get("WFX_BENCH_OVERSUBSCRIBE") == "1"
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ)
        env.update(RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   WFX_BOOT_PORT=str(boot), WFX_JOB_NONCE=nonce,
                   HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        # Every rank keeps ALL devices visible and picks its own by LOCAL_RANK: RCCL's intra-node path opens IPC handles on its
        # peers' devices and checks peer access, which a rank that sees only its own GPU cannot do (it would fall back to host
        # staging or fail).  WFX_BENCH_ISOLATE=1 restores one visible device per rank (HIP_VISIBLE_DEVICES), for experiments.
        if not over and os.environ.get("WFX_BENCH_ISOLATE") == "1":
            vis = os.environ.get("HIP_VISIBLE_DEVICES")
            devs = vis.split(",") if vis else [str(k) for k in range(args.gpus)]
            if r < len(devs):
                env.update(HIP_VISIBLE_DEVICES=devs[r], LOCAL_RANK="0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__), *sys.argv[1:]], env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL))
    import threading
    out_box = []
    reader = threading.Thread(target=lambda: out_box.append(procs[0].stdout.read()), daemon=True)
    reader.start()
    deadline = time.time() + float(os.environ.get("WFX_BENCH_TIMEOUT", "1800"))
    rc = 0
    while True:
        codes = [p.poll() for p in procs]
        bad = [c for c in codes if c not in (None, 0)]
        if bad or time.time() > deadline:
            rc = bad[0] if bad else 124
            for p in procs:
                if p.poll() is None:
                    p.terminate()
            for p in procs:
                try:
                    p.wait(timeout=10)
                except subprocess.TimeoutExpired:
                    p.kill()
            break
        if all(c == 0 for c in codes):
            break
        time.sleep(0.05)
    reader.join(timeout=10)
    sys.stdout.write((out_box[0] if out_box else b"").decode())
    sys.stdout.flush()
    return rc


class _StdoutToStderr:
    """RCCL prints a version banner to stdout when a communicator is created; the contract
    is ONE JSON line on stdout, so fd 1 points at stderr while the collectives warm up."""

    def __enter__(self):
        import ctypes
        self.libc = ctypes.CDLL(None)
        sys.stdout.flush()
        self.libc.fflush(None)
        self.saved = os.dup(1)
        os.dup2(2, 1)
        return self

    def __exit__(self, *exc):
        sys.stdout.flush()
        self.libc.fflush(None)          # RCCL printf()s into C stdio, which is fully buffered on a pipe
        os.dup2(self.saved, 1)
        os.close(self.saved)


def _rank_env():
    addr = os.environ.get("MASTER_ADDR", "127.0.0.1")
    mport = int(os.environ.get("MASTER_PORT", "29511"))
    # spawn_ranks reserves a port of its own; under a launcher: next to its rendezvous port, kept inside the valid range
    port = int(os.environ.get("WFX_BOOT_PORT", str(mport + 1009 if mport + 1009 + 16 < 65536 else mport - 1009)))
    return addr, mport, port


def _rank_device(nat) -> int:
    dev = int(os.environ.get("LOCAL_RANK", "0"))
    if os.environ.get("WFX_BENCH_OVERSUBSCRIBE") == "1":       # several ranks per GPU: only to exercise the launch path on a small box
        dev = dev % max(1, nat.device_count())
    return dev


def rccl_probe_main() -> int:
    """`bench.py --rccl-probe`, started by every rank as a CHILD before it creates its own communicator: RCCL bootstrap, the first
    collective, and the communicator's self-test (grouped send / recv exchanges in stream order and on the communicator's own stream,
    all-reduce, all-gather, every answer checked).  One line on stdout: `ok`, or the error text.  A hang -- a bootstrap that never
    completes, P2P that does not come up -- stays in this child, which its parent kills after a time limit; the parent then runs
    the job on the host-staged transport and reports why."""
    from wefax_amd import _native as nat
    from wefax_amd import sharded
    world, rank = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0"))
    addr, mport, port = _rank_env()
    nonce = os.environ.get("WFX_JOB_NONCE", f"{addr}:{mport}") + ":probe"
    t0 = time.time()
    try:
        ctx = nat.Context(_rank_device(nat))
        with _StdoutToStderr():
            uid = sharded.bootstrap_unique_id(rank, world, addr=addr, port=port, nonce=nonce, timeout=float(os.environ.get("WFX_BENCH_RCCL_PROBE_S", "75")))
            comm = nat.Comm.rccl(ctx, uid, world, rank)
            comm.barrier(ctx)
            sys.stderr.write(f"rccl probe rank {rank}: communicator up after {time.time() - t0:.1f} s\n")
            comm.selftest(ctx, 3, 20261004)
            comm.barrier(ctx)
        comm.close()
        ctx.close()
    except Exception as e:      # noqa: BLE001 -- the text IS the result
        print(("error: " + f"{type(e).__name__}: {e}").replace("\n", " ")[:300], flush=True)
        return 1
    print(f"ok ({time.time() - t0:.1f} s)", flush=True)
    return 0


class Ranks:
    """This process's place in the job: context on its GPU, RCCL communicator when there is more than one rank (or when
    WFX_BENCH_FORCE_DIST=1 asks for the real transport with a single rank on a one-GPU box)."""

    def _probe_rccl(self) -> str:
        """Run the RCCL self-test in a child of this rank (see rccl_probe_main); 'ok ...' or what went wrong, never a hang."""
        limit = float(os.environ.get("WFX_BENCH_RCCL_PROBE_S", "75"))
        try:
            r = subprocess.run([sys.executable, os.path.abspath(__file__), "--rccl-probe"], capture_output=True, text=True, timeout=limit + 15)
            last = (r.stdout.strip().splitlines() or ["(no output)"])[-1]
            if r.returncode == 0 and last.startswith("ok"):
                return last
            tail = " | ".join(ln for ln in r.stderr.strip().splitlines()[-3:])
            return f"rc {r.returncode}: {last}" + (f" [stderr: {tail[-300:]}]" if tail and not last.startswith("error") else "")
        except subprocess.TimeoutExpired:
            return (f"timeout: RCCL bootstrap + self-test of {self.world} ranks did not complete within {limit + 15:.0f} s on rank {self.rank} "
                    "(communicator never came up, or a send/recv never completed)")

    def __init__(self, args):
        from wefax_amd import _native as nat
        from wefax_amd import sharded
        self.nat = nat
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.rank = int(os.environ.get("RANK", "0"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        if args.gpus != self.world and self.world > 1:
            raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={self.world}")
        dev = _rank_device(nat)
        self.device = dev
        self.rccl_probe = None               # what the RCCL self-test said (world > 1 on RCCL only)
        self.use_rccl = self.world > 1 or os.environ.get("WFX_BENCH_FORCE_DIST") == "1"
        # transport: RCCL (one rank per GPU), or the library's shared-memory communicator -- real processes, host-staged messages --
        # where RCCL cannot run the job: several ranks on ONE GPU (WFX_BENCH_OVERSUBSCRIBE=1 on a box with fewer devices than ranks).
        # WFX_BENCH_COMM=rccl|shm forces one.
        over = os.environ.get("WFX_BENCH_OVERSUBSCRIBE") == "1"
        self.transport = os.environ.get("WFX_BENCH_COMM") or ("shm" if over and self.world > max(1, nat.device_count()) else "rccl")
        addr, mport, port = _rank_env()
        # Before this rank touches RCCL itself, a CHILD of it runs the RCCL bootstrap and the communicator's self-test under a time
        # limit (rccl_probe_main).  All ranks then compare notes over the host-staged communicator (plain shared memory and sockets):
        # unanimous `ok` -> RCCL; anything else -> the job runs on the host-staged transport (the weak-scaling headline has no
        # data-path collective and is measured all the same) and the line carries every rank's verdict in `rccl_probe`.
        probing = self.world > 1 and self.transport == "rccl" and os.environ.get("WFX_BENCH_RCCL_PROBE", "1") != "0"
        verdict = self._probe_rccl() if probing else None
        self.ctx = nat.Context(dev)
        GpuState.bind(self.ctx)
        if self.use_rccl and (self.transport == "shm" or probing):
            job = os.environ.get("WFX_JOB_NONCE")
            if not job:
                # under a launcher there is no nonce from spawn_ranks: rank 0 draws one per LAUNCH and hands it out over the
                # bootstrap socket (a job name reused across launches would let a rank attach to the control block a crashed
                # earlier run left in /dev/shm)
                blob = sharded.bootstrap_unique_id(self.rank, self.world, addr=addr, port=port, nonce=f"{addr}:{mport}:job",
                                                   make_id=lambda: os.urandom(nat.WFX_COMM_ID_BYTES))
                job = "p" + str(mport) + "-" + bytes(blob[:8]).hex()
            shm = nat.Comm.shm(self.ctx, job, self.world, self.rank, timeout=float(os.environ.get("WFX_BENCH_TIMEOUT", "600")))
            shm.barrier(self.ctx)
            if probing:
                import numpy as np
                mine = np.zeros(320, dtype=np.uint8)
                raw = verdict.encode()[:320]
                mine[:len(raw)] = np.frombuffer(raw, dtype=np.uint8)
                every = [bytes(row).rstrip(b"\0").decode(errors="replace") for row in shm.allgather(self.ctx, mine)]
                self.rccl_probe = {"ok": all(v.startswith("ok") for v in every), "per_rank": every,
                                   "what": "child process per rank: RCCL bootstrap, barrier, 3 rounds of checked exchanges (in stream order and on the "
                                           "communicator's own stream) / all-reduce / all-gather, under a time limit"}
                if self.rccl_probe["ok"]:
                    shm.close()
                else:
                    self.transport = "shm"
                    sys.stderr.write(f"bench.py rank {self.rank}: RCCL self-test failed ({every}); running on the host-staged transport\n")
            if self.transport == "shm":
                self.comm = shm
        if self.use_rccl and self.transport == "shm":
            pass
        elif self.use_rccl:
            nonce = os.environ.get("WFX_JOB_NONCE", f"{addr}:{mport}")        # peers of another job on this host are turned away
            with _StdoutToStderr():
                uid = sharded.bootstrap_unique_id(self.rank, self.world, addr=addr, port=port, nonce=nonce)
                self.comm = nat.Comm.rccl(self.ctx, uid, self.world, self.rank)
                self.comm.barrier(self.ctx)                                   # the communicator's first collective sets it up
        else:
            self.comm = nat.Comm.local(1)[0]

    def barrier(self):
        self.comm.barrier(self.ctx)

    def transport_name(self) -> str:
        if self.comm.is_rccl:
            return "RCCL"
        if getattr(self.comm, "is_shm", False):
            return "shared memory (host-staged, one process per rank; NOT a performance figure: PCIe both ways)"
        return "none (one rank)"

    def max_over_ranks(self, seconds: float) -> float:
        import numpy as np
        return float(np.max(self.comm.allgather(self.ctx, np.array([seconds], dtype=np.float64))))

    def timed(self, step, steps: int, warmup: int, sync=None) -> float:
        """W untimed steps, then exactly K steps between barrier + device sync on both sides; max over ranks (seconds)."""
        sync = sync or self.ctx.sync
        with _StdoutToStderr():
            for _ in range(warmup):
                step()
            sync()
            self.barrier()
        t0 = time.perf_counter()
        for _ in range(steps):
            step()
        sync()
        self.barrier()
        return self.max_over_ranks(time.perf_counter() - t0)

    def close(self):
        self.comm.close()
        self.ctx.close()


# ---- what the box was doing: clocks, power, temperature from sysfs ------------------------------------------------
class GpuState:
    """Reads the amdgpu sysfs nodes of the first GPU (shader / memory / fabric clock, socket power, temperatures) -- at a
    point in time (``read``) or sampled by a thread while a timed region runs (``with GpuState.sample() as s``): a reader of the
    line can then tell a slow box from a slow kernel.  Everything is optional: a node that is missing or unreadable is left out."""
    _dev = None
    _pci = None          # PCI address of the GPU the bench's context runs on (GpuState.bind): the card to read
    _matched = False

    @classmethod
    def bind(cls, ctx):
        """Read the card whose PCI address is the context's device's (round-5 verdict: on an 8-GPU box the first card is somebody else's GPU)."""
        try:
            cls._pci = ctx.pci_bus_id().lower()
        except Exception:       # noqa: BLE001
            cls._pci = None
        cls._dev = None

    @classmethod
    def dev(cls):
        if cls._dev is None:
            import glob
            cls._dev = ""
            first = ""
            for d in sorted(glob.glob("/sys/class/drm/card*/device")):
                try:
                    if open(os.path.join(d, "vendor")).read().strip() != "0x1002" or not os.path.exists(os.path.join(d, "pp_dpm_sclk")):
                        continue
                except OSError:
                    continue
                first = first or d
                if cls._pci and os.path.basename(os.path.realpath(d)).lower() == cls._pci:
                    cls._dev, cls._matched = d, True
                    break
            if not cls._dev:
                cls._dev, cls._matched = first, False
        return cls._dev

    @staticmethod
    def _cur_mhz(path):
        try:
            for ln in open(path).read().splitlines():
                if ln.rstrip().endswith("*"):
                    return int("".join(ch for ch in ln.split(":")[1] if ch.isdigit()))
        except (OSError, ValueError, IndexError):
            pass
        return None

    @classmethod
    def read(cls) -> dict:
        import glob
        d = cls.dev()
        out = {}
        if not d:
            return out
        out["pci"] = os.path.basename(os.path.realpath(d))
        out["is_the_contexts_gpu"] = bool(cls._matched)
        for key, node in (("sclk_mhz", "pp_dpm_sclk"), ("mclk_mhz", "pp_dpm_mclk"), ("fclk_mhz", "pp_dpm_fclk")):
            v = cls._cur_mhz(os.path.join(d, node))
            if v is not None:
                out[key] = v
        for hw in glob.glob(os.path.join(d, "hwmon", "hwmon*")):
            for key, node, scale in (("power_w", "power1_average", 1e-6), ("power_w", "power1_input", 1e-6), ("temp_c", "temp1_input", 1e-3),
                                     ("temp_mem_c", "temp3_input", 1e-3), ("power_cap_w", "power1_cap", 1e-6)):
                if key in out:
                    continue
                try:
                    out[key] = round(int(open(os.path.join(hw, node)).read().strip()) * scale, 1)
                except (OSError, ValueError):
                    pass
        try:
            out["busy_pct"] = int(open(os.path.join(d, "gpu_busy_percent")).read().strip())
        except (OSError, ValueError):
            pass
        return out

    class _Sampler:
        def __init__(self, period):
            import threading
            self.period, self.rows, self._stop = period, [], threading.Event()
            self._thr = threading.Thread(target=self._run, daemon=True)

        def _run(self):
            while not self._stop.is_set():
                r = GpuState.read()
                if r:
                    self.rows.append(r)
                self._stop.wait(self.period)

        def __enter__(self):
            self._thr.start()
            return self

        def __exit__(self, *exc):
            self._stop.set()
            self._thr.join(timeout=2.0)

        def summary(self) -> dict:
            out = {"samples": len(self.rows)}
            for key in ("sclk_mhz", "mclk_mhz", "fclk_mhz", "power_w", "temp_c", "temp_mem_c"):
                vals = sorted(r[key] for r in self.rows if key in r)
                if vals:
                    out[key] = {"min": vals[0], "median": vals[len(vals) // 2], "max": vals[-1]}
            return out

    @classmethod
    def sample(cls, period: float = 0.02):
        return cls._Sampler(period)


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



# ---- the fast multipole route (round 6): a6 + a7 without a transform over the capture ----------------------------------------------
F64_PEAK_TFLOPS = 78.6          # MI355X float64, vector = matrix (tools/micro/mfma_f64_rate.hip measures 76-77 with either or any mix of the two)
FMM_FMA_PER_SAMPLE = 307        # near field 82, P2M 32, L2P 32, M2L 56, M2M + L2L 38, moments <-> nodes <-> coefficients 18, notch 49


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
    ref = DecodeJob(cref, x, 11025, 120)
    ref.run()
    rinfo = ref.result()
    same = bool(np.array_equal(dig, ref.fetch("digitalized")) and info.start_frame == rinfo.start_frame)
    del ref
    cref.close()
    groups = {"notch_p2m_m2m": "fmm_notch_p2m_m2m", "tiers_and_top": "fmm_tiers_and_top", "tree_levels": "fmm_tree_levels", "near_l2p_env_median": "fmm_near_l2p_env_median"}
    us = {k: round(1e3 * prof[v][1] / args.steps, 1) for k, v in groups.items() if v in prof}
    t_fmm = sum(us.values()) * 1e-6
    flops = 2.0 * FMM_FMA_PER_SAMPLE * x.shape[0]
    n = x.shape[0]
    traffic = None
    pmc = os.path.join(REPO, "profiles", "pmc_traffic_fmm.json")
    if os.path.exists(pmc):
        try:
            tj = json.load(open(pmc))
            traffic = {"bytes_per_decode": int(sum(v["hbm_bytes_per_launch"] * (v["launches_seen"][0] / max(1, tj["fmm_tree_levels"]["launches_seen"][0]))
                                                   for k, v in tj.items() if k.startswith("fmm_") and isinstance(v, dict))),
                       "source": os.path.relpath(pmc, REPO) + " (rocprofv3 --pmc FETCH_SIZE x2 + WRITE_SIZE, separate passes; " + str(tj.get("_collected_at", "")) + ")"}
        except Exception:      # noqa: BLE001
            traffic = None
    return {"what": "BASELINE configs[1] with a6 + a7 by the fast multipole form (csrc/wfx_fmm.hip; Demodulator(hilbert_mode=4) / WEFAX_HILBERT=fmm)",
            "ms_per_step": round(ms, 4), "value": round(n / (ms / 1e3) / 1e6, 2), "unit": "Msamples/s", "stream_and_start_frame_equal_to_transform_route": same,
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
                from oracle import wefax_oracle as wo
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


# ---- file to file: wav on tmpfs -> Demodulator -> png on tmpfs (SURVEY.md 8d asks for both timings) -------------------
def bench_e2e(x, sample_rate: int, lpm: int, what: str, with_cpu: bool, reps: int = 5, cpu_x=None, cpu_what: str = "") -> dict:
    """What a user of the drop-in sees: ``Demodulator(path).process(); save_output_image(png)`` with the wav and the png on tmpfs
    (/dev/shm), one warm-up, then best of ``reps`` with a FRESH Demodulator per file (its context comes from the idle pool, like a
    service's would).  ``stages_ms`` splits one more pass through the same calls the Demodulator makes: read_and_upload (page cache ->
    the context's pinned staging buffer -> device, pipelined), decode (all kernels, to the synchronised result), png (device
    deflate + DMA of the file image to pinned host memory + check sums), write (the file image to tmpfs, a few threads).  With ``with_cpu``: the oracle + PIL on
    the same file beside it (one run) -- or on ``cpu_x``, a bounded sample of the same format, when the workload itself would keep
    the host busy for minutes (``cpu_what`` says what it is)."""
    import shutil
    import tempfile
    import numpy as np
    from wefax_amd import Demodulator, synth
    from wefax_amd import hostparams as hp
    from wefax_amd.wefax import DecodeJob, _acquire_context, _release_context
    base = "/dev/shm" if os.path.isdir("/dev/shm") and os.access("/dev/shm", os.W_OK) else None
    td = tempfile.mkdtemp(prefix="wfx_e2e_", dir=base)
    out = {"workload": what, "samples": int(x.shape[0]), "files_on": "tmpfs (/dev/shm)" if base else "the default temporary directory"}
    try:
        wav, png = os.path.join(td, "in.wav"), os.path.join(td, "out.png")
        synth.write_wav(wav, sample_rate, x)
        out["wav_bytes"] = os.path.getsize(wav)

        def once(k):
            t0 = time.perf_counter()
            d = Demodulator(wav, lpm, quiet=True, tcp_stream=False)
            d.process()
            t1 = time.perf_counter()
            d.save_output_image(os.path.join(td, f"out{k}.png"))
            t2 = time.perf_counter()
            d.close()
            return t2 - t0, t1 - t0, t2 - t1

        once(0)                                              # warm-up: context, plans, buffers, filter tables, page cache
        runs = [once(1 + k) for k in range(reps)]
        best = min(runs)
        out["ms"] = round(1e3 * best[0], 3)
        out["process_ms"], out["save_png_ms"] = round(1e3 * best[1], 3), round(1e3 * best[2], 3)
        out["all_ms"] = [round(1e3 * r[0], 3) for r in runs]
        out["png_bytes"] = os.path.getsize(os.path.join(td, "out1.png"))
        out["value"] = round(x.shape[0] / best[0] / 1e6, 2)
        out["unit"] = "Msamples/s file to file"
        # the stage split, through the calls Demodulator.process / save_output_image make (wefax_amd/wefax.py): a 16-bit PCM file is
        # read and uploaded as ONE pipeline (DecodeJob.from_wav: slices go to the device while later ones are still being read);
        # the two legs on their own, one after the other, are timed beside it
        import ctypes
        ctx = _acquire_context(0)
        notch = hp.load_notch_settings()
        layout = hp.wav_pcm16_layout(wav)
        T = [time.perf_counter()]
        job = DecodeJob.from_wav(ctx, wav, layout, lpm, notch) if layout is not None else DecodeJob(ctx, hp.read_wav(wav, alloc=ctx.staging)[1], sample_rate, lpm, notch)
        ctx.sync()
        T.append(time.perf_counter())
        job.run()
        job.result()
        T.append(time.perf_counter())
        pp_, nn_ = ctypes.c_void_p(0), ctypes.c_size_t(0)
        ctx._check(ctx.lib.wfx_decode_png_ex(ctx.h, 1, ctypes.byref(pp_), ctypes.byref(nn_)))      # kernels + DMA + check sums: the file image in pinned memory
        T.append(time.perf_counter())
        ctx.decode_save_png(png, deflate=True)                                                     # the same again + the write (a few threads)
        T.append(time.perf_counter())
        png_ms = 1e3 * (T[3] - T[2])
        out["stages_ms"] = {"read_and_upload": round(1e3 * (T[1] - T[0]), 3), "decode": round(1e3 * (T[2] - T[1]), 3), "png": round(png_ms, 3),
                            "write": round(max(0.0, 1e3 * (T[4] - T[3]) - png_ms), 3)}
        t0 = time.perf_counter()
        sr, data = hp.read_wav(wav, alloc=ctx.staging)
        t1 = time.perf_counter()
        j2 = DecodeJob(ctx, data, sr, lpm, notch)
        ctx.sync()
        t2 = time.perf_counter()
        out["stages_ms"]["read_wav_alone"], out["stages_ms"]["upload_alone"] = round(1e3 * (t1 - t0), 3), round(1e3 * (t2 - t1), 3)
        out["stages_ms"]["pipelined"] = layout is not None
        del j2
        job.run()
        job.result()
        ctx.profile_reset()
        ctx.profile_enable(True)
        ctx._check(ctx.lib.wfx_decode_png_ex(ctx.h, 1, ctypes.byref(pp_), ctypes.byref(nn_)))
        ctx.sync()
        ctx.profile_enable(False)
        out["png_kernels_ms"] = round(sum(v[1] for v in ctx.profile().values()), 3)      # (histogram, encode, scan, gather, CRC: profiled under the image id)
        _release_context(ctx, 0)
        if with_cpu:
            from PIL import Image
            from oracle import wefax_oracle as wo
            cwav, cn = wav, x.shape[0]
            if cpu_x is not None:
                cwav, cn = os.path.join(td, "cpu.wav"), cpu_x.shape[0]
                synth.write_wav(cwav, sample_rate, cpu_x)
            t0 = time.perf_counter()
            ref = wo.process(cwav, lpm, want_messages=False)
            t1 = time.perf_counter()
            if "image" in ref:
                Image.fromarray(ref["image"], "L").save(os.path.join(td, "ref.png"))
            t2 = time.perf_counter()
            out["cpu_baseline"] = {"kind": "port", "cores": 1, "process_ms": round(1e3 * (t1 - t0), 1), "save_png_ms": round(1e3 * (t2 - t1), 1),
                                   "ms": round(1e3 * (t2 - t0), 1), "value": round(cn / (t2 - t0) / 1e6, 3), "unit": "Msamples/s file to file",
                                   "sample": (cpu_what if cpu_x is not None else "the same wav file") + "; oracle (NumPy/C port of wefax.py) + PIL's PNG writer, one run"}
            if "image" in ref and cpu_x is None:
                got = np.asarray(Image.open(os.path.join(td, "out1.png")))
                out["png_pixels_equal_to_oracle"] = bool(got.shape == ref["image"].shape and np.array_equal(got, ref["image"]))
    finally:
        shutil.rmtree(td, ignore_errors=True)
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
        for trim in (2, 1):
            j2 = DecodeJob(ctx, np.ascontiguousarray(x[:n0 - trim]), 48000, 120)
            for _ in range(2):
                j2.run()
            ctx.sync()
            t0 = time.perf_counter()
            for _ in range(3):
                j2.run()
            ctx.sync()
            t_any = 1e3 * (time.perf_counter() - t0) / 3
            anyl["n0_minus_%d" % trim] = {"n0": int(n0 - trim), "n": int(j2.n), "ms_per_step": round(t_any, 3), "ratio_to_whole_seconds": round(t_any / ms, 2)}
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
                      "captures_per_gpu": args.batch, "hilbert": "fft (exact)",
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
        out["roofline"] = roofline_of(prof, args.steps, alg_bytes, ms, os.path.join(REPO, "profiles", "pmc_traffic.json"))
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


class _LineGuard:
    """Armed around the sharded part of a multi-rank default run.  If the part does not come back within `seconds` (a peer died or
    a collective hangs), rank 0 prints the line it has -- the headline is complete by then -- with the reason in `c4_strong`, and
    every rank leaves through os._exit (a rank stuck inside a collective cannot be unwound).  After the line is out the guard only
    bounds the final barrier and the communicator's teardown."""

    def __init__(self, line, rk, seconds):
        import threading
        self.line, self.rk, self.seconds = line, rk, seconds
        self.lock = threading.Lock()
        self.done = False
        # (rank 0's guard prints the line; the other ranks' guards fire a little later, so that their non-zero exit -- on which a
        # parent that spawned the ranks terminates the rest -- cannot cut rank 0 off before the line is out)
        self.timer = threading.Timer(seconds + (0.0 if rk.rank == 0 else 10.0), self._fire)
        self.timer.daemon = True
        self.timer.start()

    def claim(self) -> bool:
        """True for the caller that gets to print the line (main thread or timer, never both)."""
        with self.lock:
            first = not self.done
            self.done = True
            return first

    def _fire(self):
        if self.claim() and self.rk.rank == 0:
            out = dict(self.line)
            out["c4_strong"] = {"error": f"no result from the sharded decode within {self.seconds:.0f} s (a rank failed or a collective did not complete)"}
            sys.stdout.write(json.dumps(out) + "\n")
            sys.stdout.flush()
        sys.stderr.write(f"bench.py rank {self.rk.rank}: guard fired after {self.seconds:.0f} s\n")
        sys.stderr.flush()
        os._exit(0 if self.rk.rank == 0 else 3)

    def printed_exit_only(self):
        """Past the print: keep bounding barrier + teardown for a little while, then stand down with the process."""
        self.timer.cancel()
        import threading
        t = threading.Timer(60.0, lambda: os._exit(0))
        t.daemon = True
        t.start()


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
                cfgs["c2_fmm_route"] = {"ms_per_step": f["ms_per_step"], "notch_hilbert_envelope_median_us": f["notch_hilbert_envelope_median_us"],
                                        "f64_frac": f["roofline"]["frac"], "same_stream": f["stream_and_start_frame_equal_to_transform_route"]}
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
