"""File to file: wav on tmpfs -> Demodulator -> png on tmpfs (SURVEY.md 8d asks for both timings)."""
from __future__ import annotations

import json
import os
import subprocess
import sys
import time

import benchlib
from benchlib import REPO, HBM_PEAK_GBS, IQ_FS


# ---- file to file: wav on tmpfs -> Demodulator -> png on tmpfs (SURVEY.md 8d asks for both timings) -------------------
def bench_e2e(x, sample_rate: int, lpm: int, what: str, with_cpu: bool, reps: int = 5, cpu_x=None, cpu_what: str = "") -> dict:
    """What a user of the drop-in sees: ``Demodulator(path).process(); save_output_image(png)`` with the wav and the png on tmpfs
    (/dev/shm), one warm-up, then best of ``reps`` with a FRESH Demodulator per file (its context comes from the idle pool, like a
    service's would).  ``stages_ms`` splits one more pass through the same calls the Demodulator makes: read_and_upload (page cache ->
    the context's pinned staging buffer -> device, pipelined), decode (all kernels, to the synchronised result), png (device
    deflate + DMA of the file image to pinned host memory + check sums), write (the file image to tmpfs, a few threads).  With ``with_cpu``: the oracle + PIL on
    the same file beside it (one run) -- or on ``cpu_x``, a bounded sample of the same format, when the workload itself would keep
    the host busy for minutes (``cpu_what`` says what it is)."""
    import shutil
    import tempfile
    import numpy as np
    from wefax_amd import Demodulator, synth
    from wefax_amd import hostparams as hp
    from wefax_amd.wefax import DecodeJob, _acquire_context, _release_context
    base = "/dev/shm" if os.path.isdir("/dev/shm") and os.access("/dev/shm", os.W_OK) else None
    td = tempfile.mkdtemp(prefix="wfx_e2e_", dir=base)
    out = {"workload": what, "samples": int(x.shape[0]), "files_on": "tmpfs (/dev/shm)" if base else "the default temporary directory"}
    try:
        wav, png = os.path.join(td, "in.wav"), os.path.join(td, "out.png")
        synth.write_wav(wav, sample_rate, x)
        out["wav_bytes"] = os.path.getsize(wav)

        def once(k):
            t0 = time.perf_counter()
            d = Demodulator(wav, lpm, quiet=True, tcp_stream=False)
            d.process()
            t1 = time.perf_counter()
            d.save_output_image(os.path.join(td, f"out{k}.png"))
            t2 = time.perf_counter()
            d.close()
            return t2 - t0, t1 - t0, t2 - t1

        once(0)                                              # warm-up: context, plans, buffers, filter tables, page cache
        runs = [once(1 + k) for k in range(reps)]
        best = min(runs)
        out["ms"] = round(1e3 * best[0], 3)
        out["process_ms"], out["save_png_ms"] = round(1e3 * best[1], 3), round(1e3 * best[2], 3)
        out["all_ms"] = [round(1e3 * r[0], 3) for r in runs]
        out["png_bytes"] = os.path.getsize(os.path.join(td, "out1.png"))
        out["value"] = round(x.shape[0] / best[0] / 1e6, 2)
        out["unit"] = "Msamples/s file to file"
        # the stage split, through the calls Demodulator.process / save_output_image make (wefax_amd/wefax.py): a 16-bit PCM file is
        # read and uploaded as ONE pipeline (DecodeJob.from_wav: slices go to the device while later ones are still being read);
        # the two legs on their own, one after the other, are timed beside it
        import ctypes
        ctx = _acquire_context(0)
        notch = hp.load_notch_settings()
        layout = hp.wav_pcm16_layout(wav)
        T = [time.perf_counter()]
        job = DecodeJob.from_wav(ctx, wav, layout, lpm, notch) if layout is not None else DecodeJob(ctx, hp.read_wav(wav, alloc=ctx.staging)[1], sample_rate, lpm, notch)
        ctx.sync()
        T.append(time.perf_counter())
        job.run()
        job.result()
        T.append(time.perf_counter())
        pp_, nn_ = ctypes.c_void_p(0), ctypes.c_size_t(0)
        ctx._check(ctx.lib.wfx_decode_png_ex(ctx.h, 1, ctypes.byref(pp_), ctypes.byref(nn_)))      # kernels + DMA + check sums: the file image in pinned memory
        T.append(time.perf_counter())
        ctx.decode_save_png(png, deflate=True)                                                     # the same again + the write (a few threads)
        T.append(time.perf_counter())
        png_ms = 1e3 * (T[3] - T[2])
        out["stages_ms"] = {"read_and_upload": round(1e3 * (T[1] - T[0]), 3), "decode": round(1e3 * (T[2] - T[1]), 3), "png": round(png_ms, 3),
                            "write": round(max(0.0, 1e3 * (T[4] - T[3]) - png_ms), 3)}
        t0 = time.perf_counter()
        sr, data = hp.read_wav(wav, alloc=ctx.staging)
        t1 = time.perf_counter()
        j2 = DecodeJob(ctx, data, sr, lpm, notch)
        ctx.sync()
        t2 = time.perf_counter()
        out["stages_ms"]["read_wav_alone"], out["stages_ms"]["upload_alone"] = round(1e3 * (t1 - t0), 3), round(1e3 * (t2 - t1), 3)
        out["stages_ms"]["pipelined"] = layout is not None
        del j2
        job.run()
        job.result()
        ctx.profile_reset()
        ctx.profile_enable(True)
        ctx._check(ctx.lib.wfx_decode_png_ex(ctx.h, 1, ctypes.byref(pp_), ctypes.byref(nn_)))
        ctx.sync()
        ctx.profile_enable(False)
        out["png_kernels_ms"] = round(sum(v[1] for v in ctx.profile().values()), 3)      # (histogram, encode, scan, gather, CRC: profiled under the image id)
        _release_context(ctx, 0)
        if with_cpu:
            from PIL import Image
            wo = benchlib.ORACLE()
            cwav, cn = wav, x.shape[0]
            if cpu_x is not None:
                cwav, cn = os.path.join(td, "cpu.wav"), cpu_x.shape[0]
                synth.write_wav(cwav, sample_rate, cpu_x)
            t0 = time.perf_counter()
            ref = wo.process(cwav, lpm, want_messages=False)
            t1 = time.perf_counter()
            if "image" in ref:
                Image.fromarray(ref["image"], "L").save(os.path.join(td, "ref.png"))
            t2 = time.perf_counter()
            out["cpu_baseline"] = {"kind": "port", "cores": 1, "process_ms": round(1e3 * (t1 - t0), 1), "save_png_ms": round(1e3 * (t2 - t1), 1),
                                   "ms": round(1e3 * (t2 - t0), 1), "value": round(cn / (t2 - t0) / 1e6, 3), "unit": "Msamples/s file to file",
                                   "sample": (cpu_what if cpu_x is not None else "the same wav file") + "; oracle (NumPy/C port of wefax.py) + PIL's PNG writer, one run"}
            if "image" in ref and cpu_x is None:
                got = np.asarray(Image.open(os.path.join(td, "out1.png")))
                out["png_pixels_equal_to_oracle"] = bool(got.shape == ref["image"].shape and np.array_equal(got, ref["image"]))
    finally:
        shutil.rmtree(td, ignore_errors=True)
    return out
