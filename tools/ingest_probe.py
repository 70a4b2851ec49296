"""What bounds the front end's first stage: the integer-exact /32 decimator on 16 GiB of int16 IQ frames with 369, 256, 128 and 32 taps
(12, 8, 4 and 1 taps per polyphase row) -- if the time does not move with the tap count, the tap loop is not the limit."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
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
