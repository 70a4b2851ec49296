"""Does the ingest's rate depend on WHERE the capture lies?  The same 22 GB stream in three buffers of one process: one allocated before
anything else, the decoder's own, one allocated last.  (Plain reads do not care; the ingest's 768 separate streams might: page tables.)"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
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
