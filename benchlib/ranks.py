"""One process per GPU: spawning the ranks, reading them from the environment, the RCCL probe, the headline guard."""
from __future__ import annotations

import json
import os
import subprocess
import sys
import time

import benchlib
from benchlib import REPO, HBM_PEAK_GBS, IQ_FS
from benchlib.gpustate import GpuState


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
        procs.append(subprocess.Popen([sys.executable, os.path.join(REPO, "bench.py"), *sys.argv[1:]], env=env,
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
            r = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--rccl-probe"], capture_output=True, text=True, timeout=limit + 15)
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
