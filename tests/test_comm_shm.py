"""The shared-memory communicator (include/wefax_hip.h wfx_comm_create_shm): one PROCESS per rank, messages staged through
/dev/shm.  CPU tests drive the protocol with host buffers (no GPU: ctx = NULL); GPU tests run the same collectives on device
buffers and the whole sharded decode with several real processes on ONE GPU -- separate address spaces, contexts and streams,
ranks that reach a phase at different times: what the in-process emulation (lock-step by construction) cannot show."""
import multiprocessing as mp
import os
import secrets
import time

import numpy as np
import pytest

from wefax_amd import _native as nat


def _job():
    return "t" + secrets.token_hex(6)


def _selftest_worker(job, world, rank, rounds, seed, q, device):
    try:
        ctx = nat.Context(device) if device is not None else None
        comm = nat.Comm.shm(ctx, job, world, rank, timeout=60.0)
        comm.selftest(ctx, rounds, seed)
        comm.barrier(ctx)
        comm.close()
        if ctx is not None:
            ctx.close()
        q.put((rank, "ok"))
    except Exception as e:          # noqa: BLE001
        q.put((rank, f"{type(e).__name__}: {e}"))


def _run(world, target, args_of, timeout=120):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=target, args=args_of(r) + (q,)) for r in range(world)]
    # the ranks start out of order and apart in time
    for r in reversed(range(world)):
        procs[r].start()
        time.sleep(0.05)
    out = {}
    for _ in range(world):
        r, msg = q.get(timeout=timeout)
        out[r] = msg
    for p in procs:
        p.join(30)
    return out


@pytest.mark.parametrize("world", [2, 3, 5])
def test_randomised_collectives_between_processes_host_memory(world):
    job = _job()
    out = _run(world, _selftest_worker_q, lambda r: (job, world, r, 12, 1234 + world, None))
    assert out == {r: "ok" for r in range(world)}
    assert not [f for f in os.listdir("/dev/shm") if job in f]              # the ranks cleaned up after themselves


def _selftest_worker_q(job, world, rank, rounds, seed, device, q):
    _selftest_worker(job, world, rank, rounds, seed, q, device)


def _dying_worker(job, world, rank, q):
    try:
        comm = nat.Comm.shm(None, job, world, rank, timeout=3.0)
        if rank == 1:
            os._exit(17)                                                     # dies after the rendezvous, before the first collective
        comm.selftest(None, 4, 7)
        q.put((rank, "ok"))
    except Exception as e:          # noqa: BLE001
        q.put((rank, f"{type(e).__name__}: {e}"))


def test_a_rank_that_dies_is_an_error_on_the_others_not_a_hang():
    job = _job()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_dying_worker, args=(job, 3, r, q)) for r in range(3)]
    t0 = time.time()
    for p in procs:
        p.start()
    got = dict(q.get(timeout=60) for _ in range(2))
    for p in procs:
        p.join(30)
    assert time.time() - t0 < 40
    assert set(got) == {0, 2} and all("NativeError" in m and ("timed out" in m or "peer" in m) for m in got.values()), got
    for f in os.listdir("/dev/shm"):
        if job in f:
            os.unlink(os.path.join("/dev/shm", f))


def _disagree_worker(job, rank, q):
    """Rank 0 runs the selftest with one seed, rank 1 with another: their exchange plans differ."""
    try:
        comm = nat.Comm.shm(None, job, 2, rank, timeout=5.0)
        comm.selftest(None, 3, 100 + rank)
        q.put((rank, "ok"))
    except Exception as e:          # noqa: BLE001
        q.put((rank, f"{type(e).__name__}: {e}"))


def test_ranks_that_disagree_about_a_message_fail_loudly():
    job = _job()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_disagree_worker, args=(job, r, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=60) for _ in range(2))
    for p in procs:
        p.join(30)
    assert all("NativeError" in m for m in got.values()), got
    assert any("bytes" in m or "messages" in m or "wrong" in m or "expected" in m for m in got.values()), got
    for f in os.listdir("/dev/shm"):
        if job in f:
            os.unlink(os.path.join("/dev/shm", f))


def test_bad_arguments():
    with pytest.raises(nat.NativeError, match="job name"):
        nat.Comm.shm(None, "no/slash", 1, 0)
    with pytest.raises(nat.NativeError, match="bad rank"):
        nat.Comm.shm(None, _job(), 2, 2)
    with pytest.raises(nat.NativeError, match="never created"):
        nat.Comm.shm(None, _job(), 2, 1, timeout=0.3)
    c = nat.Comm.shm(None, _job(), 1, 0)
    c.selftest(None, 3, 5)
    assert (c.world, c.rank, c.is_rccl) == (1, 0, False)
    c.close()


# ---------------------------------------------------------------------------------------------------------------
# GPU: several processes on one device
# ---------------------------------------------------------------------------------------------------------------
@pytest.mark.gpu
@pytest.mark.parametrize("world", [2, 4])
def test_randomised_collectives_between_processes_device_memory(world):
    job = _job()
    out = _run(world, _selftest_worker_q, lambda r: (job, world, r, 8, 99 + world, 0), timeout=300)
    assert out == {r: "ok" for r in range(world)}


def _decode_worker(job, world, rank, seed, rate, lpm, out_dir, q, trim=0, plan="dist"):
    """One rank of a sharded decode in its own process: own context, own slice of the capture, the shm transport."""
    try:
        from wefax_amd import sharded, synth
        x = _capture(rate, seed, lpm, trim)
        ctx = nat.Context(0)
        comm = nat.Comm.shm(ctx, job, world, rank, timeout=120.0)
        time.sleep(0.02 * ((rank * 7) % 5))                                 # the ranks drift apart
        dec = sharded.ShardedDecoder(ctx, comm, x.shape[0], rate, lpm, sharded.capture_kind(x), data=x, plan=plan)
        for rep in range(3):                                                 # buffers are reused decode after decode
            dec.run()
            if rank % 2 == rep % 2:
                time.sleep(0.01)
        info = dec.result()
        np.save(os.path.join(out_dir, f"env{rank}.npy"), dec.fetch("envelope"))
        if rank == 0:
            np.save(os.path.join(out_dir, "stream.npy"), dec.fetch("stream"))
            np.save(os.path.join(out_dir, "image.npy"), dec.fetch("image"))
            np.save(os.path.join(out_dir, "sync.npy"), np.array([info.start_frame, info.height, info.npeaks]))
        comm.barrier(ctx)
        dec.close()
        comm.close()
        ctx.close()
        q.put((rank, "ok"))
    except Exception as e:          # noqa: BLE001
        import traceback
        q.put((rank, f"{type(e).__name__}: {e}\n{traceback.format_exc()}"))


def _capture(rate, seed, lpm, trim=0):
    from wefax_amd import synth
    n_lines = {120: 70, 240: 148}[lpm]             # whole seconds: lengths with 13-smooth halves (what the distributed transforms take)
    x = synth.synth_capture(float(rate), noise=0.05, seed=seed, lpm=lpm, start_tone_s=1.0, phasing_lines=40 if lpm == 240 else 20,
                            image_lines=n_lines, stop_tone_s=1.0, black_tail_s=1.0)
    return np.ascontiguousarray(x[:x.shape[0] - trim]) if trim else x


@pytest.mark.gpu
@pytest.mark.parametrize("world,rate,lpm,trim", [(2, 11025, 240, 0), (3, 48000, 240, 0), (4, 11025, 120, 0), (8, 48000, 120, 0),
                                                 (3, 11025, 120, 4478), (8, 11025, 240, 1234), (4, 11025, 120, 3333), (8, 11025, 240, 777)])
def test_sharded_decode_in_real_processes_equals_the_one_gpu_decode(tmp_path, world, rate, lpm, trim):
    """ShardedDecoder over the shm transport, `world` processes on one GPU, three decodes back to back with the ranks drifting
    apart: uint8 stream, image, start_frame and the float64 envelope blocks equal the fused one-GPU decode / the in-process
    emulation bit for bit.  ``trim``: an arbitrary even length (half-length not 13-smooth): the padded distributed convolution,
    whose first decode carries three extra phases (on a transform object of their own: the kernel has taps in the padding rows
    too, the capture does not); an odd trim gives an ODD length (a real convolution on packed transforms: the last pair of samples is half empty)."""
    from wefax_amd import sharded
    from wefax_amd.wefax import DecodeJob
    x = _capture(rate, 5, lpm, trim)
    assert sharded.layout_supported(x.shape[0], rate, world, lpm, sharded.capture_kind(x))
    job = _job()
    ctxm = mp.get_context("spawn")
    q = ctxm.Queue()
    procs = [ctxm.Process(target=_decode_worker, args=(job, world, r, 5, rate, lpm, str(tmp_path), q, trim)) for r in range(world)]
    for r in reversed(range(world)):
        procs[r].start()
        time.sleep(0.05)
    out = dict(q.get(timeout=600) for _ in range(world))
    for p_ in procs:
        p_.join(30)
    assert out == {r: "ok" for r in range(world)}, out
    ctx = nat.Context(0)
    ref = DecodeJob(ctx, x, rate, lpm)
    ref.run()
    info = ref.result()
    assert np.array_equal(np.load(tmp_path / "stream.npy"), ref.fetch("digitalized"))
    assert np.array_equal(np.load(tmp_path / "image.npy"), ref.fetch("image"))
    assert list(np.load(tmp_path / "sync.npy")) == [info.start_frame, info.height, info.npeaks]
    emu = sharded.decode_emulated(x, rate, world, lpm, want=("envelope",))
    from wefax_amd.wefax import build_params
    p, _ = build_params(sharded.capture_kind(x), x.shape[0], rate, 1 / (lpm / 60), shard_plan=sharded.plan_code("dist"))
    lays = [nat.shard_layout(p, world, r) for r in range(world)]
    assert np.array_equal(sharded.assemble(lays, [np.load(tmp_path / f"env{r}.npy") for r in range(world)], emu["n"]), emu["envelope"])
    assert lays[0].plan == 2                                # the columns layout -- since later in round 4 also for arbitrary lengths (padded forms)
    if trim:
        sizes = [nat.shard_layout(p, world, r).own_hi - nat.shard_layout(p, world, r).own_lo for r in range(world)]
        assert min(sizes) > 0 and max(sizes) - min(sizes) <= x.shape[0] // 16   # only the rows that hold samples are dealt: equal shares
    ctx.close()


@pytest.mark.gpu
@pytest.mark.parametrize("world,lpm,trim", [(2, 240, 0), (3, 120, 4478), (8, 240, 1234)])
def test_multipole_plan_in_real_processes_equals_the_one_gpu_decode(tmp_path, world, lpm, trim):
    """Plan 3 (chunk-local fast multipole Hilbert transform, csrc/wfx_shard.hip run_phase_fmm) over the shm transport: `world` processes on one
    GPU, three decodes back to back, the ranks drifting apart -- stream, image, start frame equal the one-GPU decode, the envelope blocks
    the one-GPU decode in the same Hilbert form bit for bit."""
    from wefax_amd import sharded
    from wefax_amd.wefax import DecodeJob, build_params
    x = _capture(11025, 5, lpm, trim)
    job = _job()
    ctxm = mp.get_context("spawn")
    q = ctxm.Queue()
    procs = [ctxm.Process(target=_decode_worker, args=(job, world, r, 5, 11025, lpm, str(tmp_path), q, trim, "fmm")) for r in range(world)]
    for r in reversed(range(world)):
        procs[r].start()
        time.sleep(0.05)
    out = dict(q.get(timeout=600) for _ in range(world))
    for p_ in procs:
        p_.join(30)
    assert out == {r: "ok" for r in range(world)}, out
    ctx = nat.Context(0)
    ref = DecodeJob(ctx, x, 11025, lpm, hilbert_mode=nat.WFX_HILBERT_FMM)
    ref.run()
    info = ref.result()
    assert np.array_equal(np.load(tmp_path / "stream.npy"), ref.fetch("digitalized"))
    assert np.array_equal(np.load(tmp_path / "image.npy"), ref.fetch("image"))
    assert list(np.load(tmp_path / "sync.npy")) == [info.start_frame, info.height, info.npeaks]
    p, _ = build_params(sharded.capture_kind(x), x.shape[0], 11025, 1 / (lpm / 60), shard_plan=sharded.plan_code("fmm"))
    lays = [nat.shard_layout(p, world, r) for r in range(world)]
    assert lays[0].plan == 3
    assert np.array_equal(sharded.assemble(lays, [np.load(tmp_path / f"env{r}.npy") for r in range(world)], x.shape[0]), ref.fetch("envelope"))
    ctx.close()


@pytest.mark.gpu
@pytest.mark.parametrize("world,lpm", [(2, 240), (5, 120)])
def test_multipole_plan_in_front_of_a_resampler_in_real_processes(tmp_path, world, lpm):
    """Plan 3 for a 48 kHz capture (the resampler's multipole form sharded as well; csrc/wfx_shard.hip run_phase_rs) over the shm transport: the
    processes' stream, image and envelope blocks are the in-process emulation's bit for bit, the stream within the parity bar of the one-GPU
    decode (whose resampler is the transform over the capture: the same sums in another order)."""
    from wefax_amd import sharded
    from wefax_amd.wefax import DecodeJob
    x = _capture(48000, 5, lpm)
    job = _job()
    ctxm = mp.get_context("spawn")
    q = ctxm.Queue()
    procs = [ctxm.Process(target=_decode_worker, args=(job, world, r, 5, 48000, lpm, str(tmp_path), q, 0, "fmm")) for r in range(world)]
    for r in reversed(range(world)):
        procs[r].start()
        time.sleep(0.05)
    out = dict(q.get(timeout=600) for _ in range(world))
    for p_ in procs:
        p_.join(30)
    assert out == {r: "ok" for r in range(world)}, out
    emu = sharded.decode_emulated(x, 48000, world, lpm, plan="fmm")
    assert emu["plan"] == 3
    assert np.array_equal(np.load(tmp_path / "stream.npy"), emu["digitalized"]) and np.array_equal(np.load(tmp_path / "image.npy"), emu["image"])
    lays = [nat.shard_layout(build_params_fmm(x, 48000, lpm), world, r) for r in range(world)]
    assert np.array_equal(sharded.assemble(lays, [np.load(tmp_path / f"env{r}.npy") for r in range(world)], emu["n"]), emu["envelope"])
    ctx = nat.Context(0)
    ref = DecodeJob(ctx, x, 48000, lpm)
    ref.run()
    info = ref.result()
    d = np.abs(emu["digitalized"].astype(np.int16) - ref.fetch("digitalized").astype(np.int16))
    assert d.max() <= 1 and np.count_nonzero(d) <= 2 and list(np.load(tmp_path / "sync.npy"))[0] == info.start_frame
    ctx.close()


def build_params_fmm(x, rate, lpm):
    from wefax_amd import sharded
    from wefax_amd.wefax import build_params
    return build_params(sharded.capture_kind(x), x.shape[0], rate, 1 / (lpm / 60), shard_plan=sharded.plan_code("fmm"))[0]


def _silence_worker(job, rank, q):
    """A constant capture overflows the select's candidate lists: the library repeats the decode by itself on every rank."""
    try:
        from wefax_amd import sharded
        x = np.full(1433250, 1200, dtype=np.int16)
        ctx = nat.Context(0)
        comm = nat.Comm.shm(ctx, job, 2, rank, timeout=120.0)
        dec = sharded.ShardedDecoder(ctx, comm, x.shape[0], 11025, 120, nat.WFX_IN_I16_MONO, data=x, plan="dist")
        dec.run()
        info = dec.result()
        ok = info.low == info.high and (rank != 0 or info.nan_count > 1000000)
        dec.run()                       # and the shard stays usable at the larger capacity
        info2 = dec.result()
        ok = ok and info2.low == info.low
        comm.barrier(ctx)
        dec.close()
        comm.close()
        ctx.close()
        q.put((rank, "ok" if ok else f"unexpected result {info.low} {info.high} {info.nan_count}"))
    except Exception as e:          # noqa: BLE001
        q.put((rank, f"{type(e).__name__}: {e}"))


@pytest.mark.gpu
def test_candidate_overflow_is_repeated_inside_the_library_on_real_ranks():
    job = _job()
    out = _run(2, _silence_worker, lambda r: (job, r), timeout=300)
    assert out == {0: "ok", 1: "ok"}, out


def _timed_worker(job, world, rank, q):
    try:
        comm = nat.Comm.shm(None, job, world, rank, timeout=30.0)
        comm.wire_timing(True)
        comm.selftest(None, 3, 99)
        stats, times = comm.wire_stats(), comm.wire_times()
        comm.wire_timing(False)
        comm.selftest(None, 1, 5)
        after = (len(comm.wire_stats()), len(comm.wire_times()), [t["clock"] for t in comm.wire_times()])
        comm.barrier(None)
        comm.close()
        q.put((rank, (stats, times, after)))
    except Exception as e:          # noqa: BLE001
        q.put((rank, f"{type(e).__name__}: {e}"))


def test_collectives_are_timed_one_record_each_in_call_order():
    """wfx_comm_wire_timing / wfx_comm_wire_times (round 5): one time record per collective, parallel to the byte records -- the shape
    `bench.py` prints per collective (`us`, `wait_us`, `hidden_us`, `stream`, `clock`).  On this transport a collective completes
    before the call returns: the host clock times it and nothing of it is hidden."""
    job = _job()
    out = _run(2, _timed_worker, lambda r: (job, 2, r))
    for r in range(2):
        assert not isinstance(out[r], str), out[r]
        stats, times, after = out[r]
        assert len(stats) == len(times) == 9 and [s["name"] for s in stats] == ["exchange", "all-reduce", "all-gather"] * 3
        for t in times:
            assert set(t) == {"us", "wait_us", "hidden_us", "stream", "clock"}
            assert t["clock"] == "host" and t["stream"] == "context" and t["us"] >= 0 and t["wait_us"] == t["us"] and t["hidden_us"] == 0
        assert after == (3, 3, [None, None, None])          # timing off: records are counted, not timed


def _disagreeing_env_worker(job, rank, q):
    """Rank 1 was started with another WFX_SHARD_CHUNKS: the plans differ in the number of k1 subsets, i.e. in the exchanges."""
    try:
        from wefax_amd import sharded
        os.environ["WFX_SHARD_CHUNKS"] = "4" if rank == 0 else "2"
        x = _capture(11025, 5, 240)
        ctx = nat.Context(0)
        comm = nat.Comm.shm(ctx, job, 2, rank, timeout=60.0)
        try:
            sharded.ShardedDecoder(ctx, comm, x.shape[0], 11025, 240, nat.WFX_IN_I16_MONO, data=x, plan="dist")
            q.put((rank, "created"))
        except nat.NativeError as e:
            q.put((rank, str(e)))
        comm.close()
        ctx.close()
    except Exception as e:          # noqa: BLE001
        q.put((rank, f"{type(e).__name__}: {e}"))


@pytest.mark.gpu
def test_ranks_whose_environments_disagree_about_the_plan_are_told_so_at_creation():
    """Round-4 advisor finding: part of the sharded plan comes from each process's own environment; ranks that decided differently used to
    issue mismatched exchanges (RCCL: a hang).  wfx_shard_create compares a digest across the ranks and fails with WFX_ERR_COMM."""
    job = _job()
    out = _run(2, _disagreeing_env_worker, lambda r: (job, r), timeout=300)
    for r in (0, 1):
        assert "k1 subsets" in out[r] and "environments" in out[r] and "error -5" in out[r], out


def _abandoning_worker(job, world, rank, q):
    try:
        keep = nat.Comm.shm(None, job, world, rank, timeout=10.0)             # (held: dropping it would close it)
        q.put((rank, "created" if keep is not None else "?"))
        time.sleep(0.2)
        os._exit(0)                                                          # no clean-up: the control block and the outboxes stay in /dev/shm
    except Exception as e:          # noqa: BLE001
        q.put((rank, f"{type(e).__name__}: {e}"))


def test_a_peer_that_joined_the_leftover_of_an_earlier_job_attaches_again():
    """A job that died without cleaning up leaves an initialised control block under its name.  In the next launch of that name a
    peer that starts BEFORE rank 0 passes every check on the leftover and waits at its first barrier there; rank 0 then replaces
    the block.  The peer must notice (barrier failed, its block no longer linked) and join the new one -- not fail, not leave rank 0
    waiting for the timeout."""
    job = _job()
    out = _run(2, _abandoning_worker, lambda r: (job, 2, r))
    assert out == {0: "created", 1: "created"}
    assert [f for f in os.listdir("/dev/shm") if job in f]                   # the leftover is there
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = {r: ctx.Process(target=_selftest_worker, args=(job, 2, r, 6, 99, q, None)) for r in (0, 1)}
    t0 = time.time()
    procs[1].start()
    time.sleep(1.0)                                                          # the peer is on the leftover's barrier by now
    procs[0].start()
    got = dict(q.get(timeout=60) for _ in range(2))
    for p in procs.values():
        p.join(30)
    assert got == {0: "ok", 1: "ok"}
    assert time.time() - t0 < 30.0                                           # (the communicator's timeout is 60 s: nobody sat it out)
    assert not [f for f in os.listdir("/dev/shm") if job in f]
