"""The fused ingest on the 60-minute stream itself (22 GB): does the rate depend on the DATA (synthesised WEFAX + noise against a
repeating ramp) or on what runs between two launches (the rest of the decode)?"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
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
