"""Where the streaming ingest kernel (csrc/wfx_ingest.hip) spends its time: the fused / 32 -> / 3 chain on 16 GiB of int16 IQ frames
with parts of the kernel switched off (WFX_INGEST_DBG=flags -- 1 = no LDS stash, 2 = no stage 2, 4 = no stage 1, 8 = no barrier B, 16 = stage 2's stores dropped; results are
wrong, times are what is asked), other run lengths, and the tile kernels of rounds 1-4 beside it.
    gpurun -- 'python tools/ingest_lab.py [GiB]'"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
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
