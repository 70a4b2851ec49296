"""How often is a 22 GB allocation a slow one for the ingest's 768 streams?  Allocations held side by side, before and after a few whole
decodes have churned the context's buffers; the kernel with parts switched off (data does not matter to it) and a plain read of each."""
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

