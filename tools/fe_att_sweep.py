#!/usr/bin/env python3
"""Stop-band attenuation of the time-domain front end vs agreement with the reference (40-s noisy 1.536 MS/s IQ clip, one-GPU
fused form) and vs the cost of the ingest kernel on the 60-minute stream.  One JSON line per attenuation."""
import json
import os
import sys
import tempfile
import time

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from wefax_amd import _native as nat, polyphase as pp, sharded, synth, synth_device      # noqa: E402

fs = 1536000
x = synth.synth_capture(float(fs), noise=0.05, seed=0, lpm=120, phasing_lines=20, image_lines=54, start_tone_s=1.0, stop_tone_s=1.0, black_tail_s=1.0, iq=True)
from oracle import wefax_oracle as wo      # noqa: E402
with tempfile.TemporaryDirectory() as td:
    p = os.path.join(td, "c.wav")
    synth.write_wav(p, fs, x)
    ref = wo.process(p, 120, want_messages=False)
ctx = nat.Context(0)
kw = dict(start_tone_s=5.0, phasing_lines=60, image_lines=int((3600 - 15.0) / 0.5) - 60, stop_tone_s=5.0, black_tail_s=5.0)
sp = synth_device.synth_params(float(fs), noise=0.05, seed=0, iq=True, **kw)
n0 = int(ctx.lib.wfx_synth_frames(sp))
big = synth_device.synth_slice(ctx, sp, -20000, n0 + 20000)
for att in [float(a) for a in (sys.argv[1:] or ["90", "105", "120", "135"])]:
    fe = pp.FrontEnd(fs, att_db=att, stop_rate=int(os.environ.get("WFX_FE_STOP", 16000)))
    dec = sharded.FrontEndExactDecoder(ctx, fe, x, lines_per_minute=120)
    dec.run()
    info = dec.result()
    img, st = dec.fetch("image"), dec.fetch("digitalized")
    dec.close()
    d = np.abs(st.astype(np.int16) - ref["digitalized"].astype(np.int16))
    di = np.abs(img.astype(np.int16) - ref["image"].astype(np.int16))
    rec = {"att_db": att, "taps": [getattr(s, "ntaps", getattr(s, "taps", 0)) for s in fe.stages], "stream_ne": int(np.count_nonzero(d)), "stream_max": int(d.max()),
           "image_ne": int(np.count_nonzero(di)), "image_gt1": int(np.count_nonzero(di > 1)), "image_max": int(di.max()), "start_eq": bool(info.start_frame == ref["start_frame"])}
    # cost on the 60-minute stream
    def loader(lo, hi):
        return big + (lo + 20000) * 4, hi - lo
    full = sharded.FrontEndExactDecoder(ctx, fe, None, n_in_total=n0, in_kind=nat.WFX_IN_I16_STEREO, lines_per_minute=120, raw_loader=loader)
    for _ in range(2):
        full.run()
    ctx.sync()
    t0 = time.perf_counter()
    for _ in range(5):
        full.run()
    ctx.sync()
    rec["ms_60min"] = round((time.perf_counter() - t0) / 5 * 1e3, 3)
    ctx.profile_reset(); ctx.profile_enable(True); full.run(); ctx.sync(); ctx.profile_enable(False)
    pr = ctx.profile()
    rec["ingest_us"] = round(pr["polyphase_ingest"][1] * 1e3, 1)
    rec["stages_us"] = round(pr["polyphase_stages"][1] * 1e3, 1)
    full.close()
    print(json.dumps(rec), flush=True)
