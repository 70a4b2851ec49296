"""Lab bench of the oversampled front end's ingest (csrc/wfx_ingest.hip, round 5): the six scripts that produced DESIGN 3.6 / the
"Round 5" rows of EXPERIMENTS.md, as sub-commands of one tool.  Parts of the kernel can be switched off only in a LAB build
(`bash tools/build_variant.sh lab wfx_ingest -DWFX_LAB`, then WFX_LIB=wefax_amd/variants/libwefax_hip.lab.so): the shipped library
ignores WFX_INGEST_DBG / _DBG_LDS / _CLK.

    python tools/ingest_lab.py switches [GiB]     parts of the kernel switched off (wrong results, the times asked for), run lengths, the tile kernels beside it
    python tools/ingest_lab.py stream             the 22 GB stream itself: does the rate depend on the data or on what runs between launches?
    python tools/ingest_lab.py where              the same stream in three allocations of one process
    python tools/ingest_lab.py allocations        how often is a 22 GB allocation a slow one?
    python tools/ingest_lab.py taps               the /32 decimator with 369 / 256 / 128 / 32 taps: is the tap loop the limit?
    python tools/ingest_lab.py realloc            one buffer allocated, used and freed several times
"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

def cmd_switches(argv):
    """Where the streaming ingest kernel (csrc/wfx_ingest.hip) spends its time: the fused / 32 -> / 3 chain on 16 GiB of int16 IQ frames
with parts of the kernel switched off (WFX_INGEST_DBG=flags -- 1 = no LDS stash, 2 = no stage 2, 4 = no stage 1, 8 = no barrier B, 16 = stage 2's stores dropped; results are
wrong, times are what is asked), other run lengths, and the tile kernels of rounds 1-4 beside it.
    gpurun -- 'python tools/ingest_lab.py [GiB]'"""
    sys.argv = [sys.argv[0]] + list(argv)

    import numpy as np
    from wefax_amd import _native as nat
    from wefax_amd import polyphase as pp

    gib = float(sys.argv[1]) if len(sys.argv) > 1 else 16.0
    ctx = nat.Context(0)
    fe = pp.FrontEnd(1536000)
    s1, s2 = fe.stages
    frames = int(gib * (1 << 30)) // 4
    n1 = frames // 32 - 16
    n2 = (n1 - s2.ntaps) // 3 + 1
    n1 = (n2 - 1) * 3 + s2.ntaps
    p_in = ctx.dev_malloc(frames * 4)
    p_mid = ctx.dev_malloc(n1 * 8)
    p_out = ctx.dev_malloc(n2 * 8)
    x = (np.arange(1 << 20, dtype=np.int32) % 2001 - 1000).astype(np.int16)
    blk = np.stack([x, x[::-1]], axis=1).copy()
    for off in range(0, frames, 1 << 20):
        ctx.dev_upload(p_in + off * 4, blk[:min(1 << 20, frames - off)])


    def timed(fn, reps=3):
        fn()
        ctx.sync()
        best = 1e9
        for _ in range(reps):
            t0 = time.perf_counter()
            fn()
            ctx.sync()
            best = min(best, time.perf_counter() - t0)
        return best


    def fused():
        assert ctx.d_ingest_chain(p_in, nat.WFX_IN_I16_STEREO, frames, 32, s1.coef64, s1.fix_shift, 3, s2.coef64, p_out, n2)


    def stage1_only():
        assert ctx.d_ingest_chain(p_in, nat.WFX_IN_I16_STEREO, frames, 32, s1.coef64, s1.fix_shift, 0, None, p_mid, n1)


    def tile():
        ctx.d_decimate_fir64(p_in, nat.WFX_IN_I16_STEREO, frames, 0, 32, s1.coef64, p_mid, n1, s1.fix_shift)


    def report(name, dt):
        print(f"{name:44s} {dt * 1e3:8.3f} ms  {frames * 4 / dt / 1e12:5.2f} TB/s", flush=True)


    print(f"plain read of the same buffer: {ctx.d_read_rate(p_in, frames * 4, 3) / 1e3:.2f} TB/s", flush=True)
    # does the clock ramp?  consecutive calls right after an idle second, each timed on its own
    time.sleep(1.0)
    seq = []
    for _ in range(40):
        t0 = time.perf_counter()
        fused()
        ctx.sync()
        seq.append(1e3 * (time.perf_counter() - t0))
    print("consecutive fused calls after 1 s idle (ms):", " ".join(f"{v:.2f}" for v in seq), flush=True)
    report("fused", timed(fused))
    report("stage 1 only (y1 to memory)", timed(stage1_only))
    os.environ["WFX_INGEST_TILE"] = "1"
    report("tile kernel (rounds 1-4), stage 1", timed(tile))
    del os.environ["WFX_INGEST_TILE"]
    for flags in (0, 16, 2, 4, 6, 7, 15):
        rows = 8
        os.environ["WFX_INGEST_DBG"] = str(flags)
        what = " ".join(w for b, w in ((1, "no-stash"), (2, "no-stage2"), (4, "no-stage1"), (8, "no-barrier-B"), (16, "no-stores")) if flags & b)
        report(f"fused  rows={rows} flags={flags} ({what})", timed(fused))
    del os.environ["WFX_INGEST_DBG"]
    for extra in (0, 8192, 36000, 100000):
        os.environ["WFX_INGEST_DBG_LDS"] = str(extra)
        report(f"fused  + {extra} bytes of LDS ({160 * 1024 // (46656 + extra)} workgroups per CU)", timed(fused))
    del os.environ["WFX_INGEST_DBG_LDS"]
    for ni in (4, 8, 16, 24):
        os.environ["WFX_INGEST_NI"] = str(ni)
        report(f"fused  run length {ni} iterations", timed(fused))
    del os.environ["WFX_INGEST_NI"]


def cmd_stream(argv):
    """The fused ingest on the 60-minute stream itself (22 GB): does the rate depend on the DATA (synthesised WEFAX + noise against a
repeating ramp) or on what runs between two launches (the rest of the decode)?"""
    sys.argv = [sys.argv[0]] + list(argv)

    import numpy as np
    from wefax_amd import _native as nat
    from wefax_amd import polyphase as pp, sharded, synth_device
    import bench

    ctx = nat.Context(0)
    fe = pp.FrontEnd(1536000)
    s1, s2 = fe.stages
    kw = bench.iq_recipe(3600.0)
    sp = synth_device.synth_params(1536000.0, noise=0.05, seed=0, iq=True, **kw)
    n0 = int(ctx.lib.wfx_synth_frames(sp))
    loader = synth_device.SliceLoader(ctx, sp)
    dec = sharded.FrontEndExactDecoder(ctx, fe, None, n_in_total=n0, in_kind=nat.WFX_IN_I16_STEREO, lines_per_minute=120, raw_loader=loader)
    p_in, frames = dec.fe.p_raw, dec.fe.n_raw
    n2 = dec.fe.n_out
    p_out = dec.fe.p_out


    def fused():
        assert ctx.d_ingest_chain(p_in, nat.WFX_IN_I16_STEREO, frames, 32, s1.coef64, s1.fix_shift, 3, s2.coef64, p_out, n2)


    def timed(fn, reps=5, warm=12):
        for _ in range(warm):
            fn()
        ctx.sync()
        ts = []
        for _ in range(reps):
            t0 = time.perf_counter()
            fn()
            ctx.sync()
            ts.append(1e3 * (time.perf_counter() - t0))
        return min(ts), sorted(ts)[len(ts) // 2]


    print(f"frames {frames} ({frames * 4 / 1e9:.2f} GB), p_in {p_in:#x}", flush=True)
    print("plain read: %.2f TB/s" % (ctx.d_read_rate(p_in, frames * 4, 3) / 1e3), flush=True)
    print("synthesised stream, ingest back to back: min %.3f median %.3f ms" % timed(fused), flush=True)
    # ingest inside the whole decode: HIP-event time of the ingest launch
    for _ in range(12):
        dec.run()
    ctx.sync()
    ctx.profile_reset()
    ctx.profile_enable(True)
    for _ in range(5):
        dec.run()
    ctx.sync()
    ctx.profile_enable(False)
    pr = ctx.profile()
    print("ingest inside the decode (events): %.3f ms per launch; whole decode kernels %.3f ms" % (pr["polyphase_ingest"][1] / pr["polyphase_ingest"][0], sum(v[1] for v in pr.values()) / 5), flush=True)
    for ni in (8, 16, 32):
        os.environ["WFX_INGEST_NI"] = str(ni)
        print("  run length %2d iterations: min %.3f median %.3f ms" % ((ni,) + timed(fused, 5, 4)), flush=True)
    del os.environ["WFX_INGEST_NI"]
    FLAGS = ((1, "no-stash"), (2, "no-stage2"), (4, "no-stage1"), (8, "no-barrier-B"), (16, "no-stores"))


    def parts(tag):
        """the kernel on what the buffer holds now: whole, and with parts switched off (results wrong, times asked for); one more launch of
        each with the in-kernel clock probe on (its line goes to stderr)"""
        print("%s: plain read %.2f TB/s" % (tag, ctx.d_read_rate(p_in, frames * 4, 3) / 1e3), flush=True)
        for flags in (0, 16, 2, 6, 15, 0):
            os.environ["WFX_INGEST_DBG"] = str(flags)
            what = " ".join(w for b, w in FLAGS if flags & b) or "everything on"
            print("%s: %-40s min %.3f median %.3f ms" % ((tag, what) + timed(fused, 5, 8)), flush=True)
            os.environ["WFX_INGEST_CLK"] = "1"
            fused()
            del os.environ["WFX_INGEST_CLK"]
        del os.environ["WFX_INGEST_DBG"]


    parts("synthesised stream")
    x = (np.arange(1 << 20, dtype=np.int32) % 2001 - 1000).astype(np.int16)
    blk = np.stack([x, x[::-1]], axis=1).copy()
    for off in range(0, frames, 1 << 20):
        ctx.dev_upload(p_in + off * 4, blk[:min(1 << 20, frames - off)])
    parts("repeating ramp in the same buffer")
    z = np.zeros((1 << 22, 2), dtype=np.int16)
    for off in range(0, frames, 1 << 22):
        ctx.dev_upload(p_in + off * 4, z[:min(1 << 22, frames - off)])
    parts("zeros in the same buffer")
    rng = np.random.default_rng(1)
    r = rng.integers(-32768, 32767, size=(1 << 22, 2), dtype=np.int16)
    for off in range(0, frames, 1 << 22):
        ctx.dev_upload(p_in + off * 4, r[:min(1 << 22, frames - off)])
    parts("uniform random int16 in the same buffer")


def cmd_where(argv):
    """Does the ingest's rate depend on WHERE the capture lies?  The same 22 GB stream in three buffers of one process: one allocated before
anything else, the decoder's own, one allocated last.  (Plain reads do not care; the ingest's 768 separate streams might: page tables.)"""
    sys.argv = [sys.argv[0]] + list(argv)

    from wefax_amd import _native as nat
    from wefax_amd import polyphase as pp, sharded, synth_device
    import bench

    ctx = nat.Context(0)
    fe = pp.FrontEnd(1536000)
    s1, s2 = fe.stages
    kw = bench.iq_recipe(3600.0)
    sp = synth_device.synth_params(1536000.0, noise=0.05, seed=0, iq=True, **kw)
    n0 = int(ctx.lib.wfx_synth_frames(sp))
    first = ctx.dev_malloc(n0 * 4 + (1 << 21))
    loader = synth_device.SliceLoader(ctx, sp)
    dec = sharded.FrontEndExactDecoder(ctx, fe, None, n_in_total=n0, in_kind=nat.WFX_IN_I16_STEREO, lines_per_minute=120, raw_loader=loader)
    p_in, frames = dec.fe.p_raw, dec.fe.n_raw
    n2, p_out = dec.fe.n_out, dec.fe.p_out
    for _ in range(3):
        dec.run()
    ctx.sync()
    last = ctx.dev_malloc(frames * 4 + (1 << 21))
    ctx.dev_copy(first, p_in, frames * 4)
    ctx.dev_copy(last, p_in, frames * 4)
    ctx.sync()


    def timed(p, reps=5, warm=8):
        def fn():
            assert ctx.d_ingest_chain(p, nat.WFX_IN_I16_STEREO, frames, 32, s1.coef64, s1.fix_shift, 3, s2.coef64, p_out, n2)
        for _ in range(warm):
            fn()
        ctx.sync()
        ts = []
        for _ in range(reps):
            t0 = time.perf_counter()
            fn()
            ctx.sync()
            ts.append(1e3 * (time.perf_counter() - t0))
        return min(ts), sorted(ts)[len(ts) // 2]


    for rep in range(2):
        for name, p in (("allocated first", first), ("the decoder's", p_in), ("allocated last", last), ("allocated first + 1 MiB", first + (1 << 20))):
            rr = ctx.d_read_rate(p, frames * 4, 3) / 1e3
            os.environ.pop("WFX_INGEST_DBG", None)
            a = timed(p)
            os.environ["WFX_INGEST_DBG"] = "15"
            b = timed(p)
            del os.environ["WFX_INGEST_DBG"]
            print(f"{name:24s} {p:#x}: plain read {rr:.2f} TB/s; ingest min {a[0]:.3f} median {a[1]:.3f} ms; loads only min {b[0]:.3f} median {b[1]:.3f} ms", flush=True)


def cmd_allocations(argv):
    """How often is a 22 GB allocation a slow one for the ingest's 768 streams?  Allocations held side by side, before and after a few whole
decodes have churned the context's buffers; the kernel with parts switched off (data does not matter to it) and a plain read of each."""
    sys.argv = [sys.argv[0]] + list(argv)

    from wefax_amd import _native as nat
    from wefax_amd import polyphase as pp, sharded, synth_device
    import bench

    ctx = nat.Context(0)
    fe = pp.FrontEnd(1536000)
    s1, s2 = fe.stages
    kw = bench.iq_recipe(3600.0)
    sp = synth_device.synth_params(1536000.0, noise=0.05, seed=0, iq=True, **kw)
    frames = int(ctx.lib.wfx_synth_frames(sp))
    n2 = fe.n_out(frames) if hasattr(fe, "n_out") else None
    chain = pp.FrontEnd(1536000, stop_rate=pp.FrontEnd.handover_rate(1536000))
    n2 = chain.n_out(frames)
    p_out = ctx.dev_malloc(n2 * 8 + 64)


    def skeleton(p, reps=4, warm=6, flags="15"):
        os.environ["WFX_INGEST_DBG"] = flags
        def fn():
            assert ctx.d_ingest_chain(p, nat.WFX_IN_I16_STEREO, frames, 32, s1.coef64, s1.fix_shift, 3, s2.coef64, p_out, n2)
        for _ in range(warm):
            fn()
        ctx.sync()
        ts = []
        for _ in range(reps):
            t0 = time.perf_counter()
            fn()
            ctx.sync()
            ts.append(1e3 * (time.perf_counter() - t0))
        del os.environ["WFX_INGEST_DBG"]
        return min(ts)


    def round_of(tag, k=8):
        ps = []
        for i in range(k):
            t0 = time.perf_counter()
            p = ctx.dev_malloc(frames * 4 + (1 << 21))
            dt = time.perf_counter() - t0
            ps.append(p)
            print(f"{tag} allocation {i} at {p:#x} ({dt * 1e3:.1f} ms to allocate): loads only {skeleton(p):.3f} ms, loads + stage 2 {skeleton(p, flags='13'):.3f}, stage 2 without its stores {skeleton(p, flags='16'):.3f},  all {skeleton(p, flags='0'):.3f}, plain read {ctx.d_read_rate(p, frames * 4, 2) / 1e3:.2f} TB/s", flush=True)
        for p in ps:
            ctx.dev_free(p)


    round_of("fresh process:", 6)
    loader = synth_device.SliceLoader(ctx, sp)
    dec = sharded.FrontEndExactDecoder(ctx, chain, None, n_in_total=frames, in_kind=nat.WFX_IN_I16_STEREO, lines_per_minute=120, raw_loader=loader)
    for _ in range(3):
        dec.run()
    ctx.sync()
    print(f"the decoder's own buffer at {dec.fe.p_raw:#x}: loads only {skeleton(dec.fe.p_raw):.3f} ms", flush=True)
    round_of("after three decodes:", 4)
    dec.close()


def cmd_taps(argv):
    """What bounds the front end's first stage: the integer-exact /32 decimator on 16 GiB of int16 IQ frames with 369, 256, 128 and 32 taps
(12, 8, 4 and 1 taps per polyphase row) -- if the time does not move with the tap count, the tap loop is not the limit."""
    sys.argv = [sys.argv[0]] + list(argv)
    import numpy as np
    from wefax_amd import _native as nat

    ctx = nat.Context(0)
    frames = (16 << 30) // 4
    n_out = frames // 32 - 64
    p_in = ctx.dev_malloc(frames * 4)
    p_out = ctx.dev_malloc(n_out * 8)
    x = (np.arange(1 << 20, dtype=np.int32) % 2001 - 1000).astype(np.int16)
    blk = np.stack([x, x[::-1]], axis=1).copy()
    for off in range(0, frames, 1 << 20):       # fill with something non-constant
        ctx.dev_upload(p_in + off * 4, blk[:min(1 << 20, frames - off)])
    for taps in (369, 256, 128, 32):
        from wefax_amd import polyphase as pp
        c = np.hanning(taps + 2)[1:-1].astype(np.float64)
        c /= c.sum()
        sh = pp.fix_shift_for(c, 32)
        c = pp.quantize_taps(c, sh)
        for rep in range(2):
            ctx.d_decimate_fir64(p_in, nat.WFX_IN_I16_STEREO, frames, 0, 32, c, p_out, n_out, sh)
        ctx.sync()
        t0 = time.perf_counter()
        for rep in range(3):
            ctx.d_decimate_fir64(p_in, nat.WFX_IN_I16_STEREO, frames, 0, 32, c, p_out, n_out, sh)
        ctx.sync()
        dt = (time.perf_counter() - t0) / 3
        print(f"{taps:4d} taps: {dt * 1e3:.3f} ms  {frames * 4 / dt / 1e12:.2f} TB/s", flush=True)


def cmd_realloc(argv):
    """Does the rate of the ingest kernels depend on WHERE a buffer landed?  Round 5 saw the same binary run the same kernel at 3.3 or 3.8 ms
(16 GiB of IQ frames) in two processes started one after the other on one box.  Here, inside ONE process: the input buffer is
allocated, used and freed several times, then several inputs live at once; every kernel is timed on every buffer."""
    sys.argv = [sys.argv[0]] + list(argv)

    import numpy as np
    from wefax_amd import _native as nat
    from wefax_amd import polyphase as pp

    ctx = nat.Context(0)
    fe = pp.FrontEnd(1536000)
    s1, s2 = fe.stages
    gib = float(sys.argv[1]) if len(sys.argv) > 1 else 16.0
    frames = int(gib * (1 << 30)) // 4
    n1 = frames // 32 - 16
    n2 = (n1 - s2.ntaps) // 3 + 1
    n1 = (n2 - 1) * 3 + s2.ntaps
    x = (np.arange(1 << 20, dtype=np.int32) % 2001 - 1000).astype(np.int16)
    blk = np.stack([x, x[::-1]], axis=1).copy()


    def fill(p):
        for off in range(0, frames, 1 << 20):
            ctx.dev_upload(p + off * 4, blk[:min(1 << 20, frames - off)])


    def timed(fn, reps=4):
        for _ in range(6):          # (clock ramp)
            fn()
        ctx.sync()
        best = 1e9
        for _ in range(reps):
            t0 = time.perf_counter()
            fn()
            ctx.sync()
            best = min(best, time.perf_counter() - t0)
        return best * 1e3


    def measure(tag, p_in, p_mid, p_out):
        f = lambda: ctx.d_ingest_chain(p_in, nat.WFX_IN_I16_STEREO, frames, 32, s1.coef64, s1.fix_shift, 3, s2.coef64, p_out, n2)       # noqa: E731
        g = lambda: ctx.d_ingest_chain(p_in, nat.WFX_IN_I16_STEREO, frames, 32, s1.coef64, s1.fix_shift, 0, None, p_mid, n1)            # noqa: E731
        t_f, t_g = timed(f), timed(g)
        os.environ["WFX_INGEST_TILE"] = "1"
        t_t = timed(lambda: ctx.d_decimate_fir64(p_in, nat.WFX_IN_I16_STEREO, frames, 0, 32, s1.coef64, p_mid, n1, s1.fix_shift))
        del os.environ["WFX_INGEST_TILE"]
        rr = ctx.d_read_rate(p_in, frames * 4, 3) / 1e3
        print(f"{tag:28s} in {p_in:#x} mid {p_mid:#x} out {p_out:#x}: fused {t_f:6.3f}  stage-1-only {t_g:6.3f}  tile {t_t:6.3f} ms   plain read {rr:.2f} TB/s", flush=True)


    for trial in range(3):
        p_in, p_mid, p_out = ctx.dev_malloc(frames * 4), ctx.dev_malloc(n1 * 8), ctx.dev_malloc(n2 * 8)
        fill(p_in)
        measure(f"alloc/free trial {trial}", p_in, p_mid, p_out)
        measure(f"  same buffers again", p_in, p_mid, p_out)
        ctx.dev_free(p_out), ctx.dev_free(p_mid), ctx.dev_free(p_in)
    # outputs allocated BEFORE the input, and a spacer in between
    p_mid, p_out = ctx.dev_malloc(n1 * 8), ctx.dev_malloc(n2 * 8)
    spacer = ctx.dev_malloc(3 << 30)
    p_in = ctx.dev_malloc(frames * 4)
    fill(p_in)
    measure("outputs first, 3 GiB spacer", p_in, p_mid, p_out)
    ins = [p_in]
    for k in range(3):
        q = ctx.dev_malloc(frames * 4)
        fill(q)
        ins.append(q)
    for k, q in enumerate(ins):
        measure(f"four inputs alive, #{k}", q, p_mid, p_out)


COMMANDS = {"switches": cmd_switches, "stream": cmd_stream, "where": cmd_where, "allocations": cmd_allocations, "taps": cmd_taps, "realloc": cmd_realloc}

if __name__ == "__main__":
    if len(sys.argv) < 2 or sys.argv[1] not in COMMANDS:
        raise SystemExit(__doc__)
    COMMANDS[sys.argv[1]](sys.argv[2:])
