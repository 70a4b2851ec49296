"""Does the rate of the ingest kernels depend on WHERE a buffer landed?  Round 5 saw the same binary run the same kernel at 3.3 or 3.8 ms
(16 GiB of IQ frames) in two processes started one after the other on one box.  Here, inside ONE process: the input buffer is
allocated, used and freed several times, then several inputs live at once; every kernel is timed on every buffer."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
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
