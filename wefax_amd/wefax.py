"""Drop-in replacement for wojlin/WEFAX ``wefax.py``: same ``Demodulator`` API
(/root/reference/wefax.py:18-408), same progress messages, same exceptions, same
output image layout -- the arithmetic runs in hand-written HIP kernels on an
MI355X through the C ABI of include/wefax_hip.h.

    python wefax.py <in.wav> <lines_per_minute> <out.png>        (wefax.py:411-424)

What is deliberately different from the reference (none of it changes a pixel):
  * no ``time.sleep`` calls (the reference sleeps 7 s, wefax.py:59,73,75,77,193,202);
  * ``quiet=True`` silences this module's own prints instead of replacing
    ``sys.stdout`` process-wide (wefax.py:48-51,92-93);
  * ``digitalized_data`` is a numpy uint8 array, not a Python list of int;
  * stage arrays (``audio_data`` ...) are copied from the GPU lazily on first access.
"""
from __future__ import annotations

import os
import threading
import sys

import numpy as np

from . import _native as nat
from . import hostparams as hp

# how a6 + a7 run: "fft" = notch kernel + one packed circular convolution on the mixed-radix transform passes; "fmm" = notch + near field +
# fast multipole far field in four kernels (csrc/wfx_fmm.hip; even N >= 32768, other lengths take the transform form).  "auto" (the default
# since round 6): the multipole route where it is the faster one on one GPU -- captures at 11 025 Hz of an even length >= 400 000 samples
# (tools/route_time.py: 0.98x at 30 s, 1.05x at 40 s, 1.0-1.3x at 50-110 s, 1.09x at 130 s, 1.20x at 5 min, 1.06x at 10 min, 1.13x at 20 min, 1.05x at 60 min; and no
# dependence on the length's factors: the padded transforms of a general length cost 1.55x) -- the transform route otherwise (short
# captures; resampled ones, whose multipole resampler is the slower one on one GPU).  Both give the same uint8 stream.
HILBERT_AUTO = -1
FMM_FROM_SAMPLES = 400000
DEFAULT_HILBERT_MODE = {"fft": nat.WFX_HILBERT_FFT, "fmm": nat.WFX_HILBERT_FMM}.get(os.environ.get("WEFAX_HILBERT", "auto"), HILBERT_AUTO)


RS_FMM_FROM_FRAMES = 1000000      # (a resampled capture of GENERAL length: the multipole resampler against the chirp-z form, see below)


def resolve_hilbert_mode(mode: int, n: int, resampled: bool, n0: int | None = None) -> int:
    """The route a one-GPU decode of n samples at 11 025 Hz (from n0 frames at the capture's rate) takes under ``mode``.  HILBERT_AUTO: see
    above; a RESAMPLED capture takes the multipole route -- resampler and Hilbert transform -- where the transform-based resampler would need
    its chirp-z form (lengths whose halves are not 13-smooth: most real recordings) and the multipole forms exist (down-sampling to an even
    count): 1.35 against 2.2 ms for ten minutes at 48 kHz, the same stream."""
    if mode != HILBERT_AUTO:
        return int(mode)
    if resampled:
        if n0 is not None and n % 2 == 0 and n0 > n and n0 >= RS_FMM_FROM_FRAMES and n >= 32768 and n0 < (1 << 31) and not nat.resample_direct(n0, n):
            return nat.WFX_HILBERT_FMM
        return nat.WFX_HILBERT_FFT
    return nat.WFX_HILBERT_FMM if (n % 2 == 0 and n >= FMM_FROM_SAMPLES) else nat.WFX_HILBERT_FFT



def build_params(kind: int, n0: int, sample_rate, frame_len: float, notch=hp.DEFAULT_NOTCH,
                 hilbert_mode: int = DEFAULT_HILBERT_MODE, n_out: int | None = None, shard_plan: int = 0):
    """The scalar arithmetic of the reference for a capture of ``n0`` frames at ``sample_rate`` (lengths, notch
    coefficients, percentile ranks and weights, sync constants) as the C ABI's ``wfx_decode_params``, plus the derived
    lengths.  Same expressions as wefax.py, evaluated in Python floats / NumPy scalars like there.

    ``n_out``: the number of 11 025 Hz samples, when the ``n0`` samples are an intermediate of a longer chain (the time-domain
    front end's hand-over: the reference's ``int(11025 * length)`` refers to the ORIGINAL capture, polyphase.FrontEnd.n_target)."""
    input_length = n0 / sample_rate                                    # wefax.py:357
    resampled = sample_rate != hp.TARGET_RATE                          # wefax.py:60
    n = int(hp.TARGET_RATE * input_length) if resampled else n0       # wefax.py:384
    if n_out is not None:
        if not resampled and n_out != n0:
            raise ValueError("n_out given for a capture that is not resampled")
        n = int(n_out)
    if n <= 9:
        raise ValueError("The length of the input vector x must be greater than padlen, which is 9.")
    b, a = hp.iirnotch(int(notch[0]), notch[1], hp.TARGET_RATE)        # wefax.py:63-70
    p = nat.DecodeParams()
    p.in_kind, p.n0, p.n, p.resample = kind, n0, n, int(resampled)
    p.notch_b[:] = [float(v) for v in b]
    p.notch_a[:] = [float(v) for v in a]
    # (HILBERT_AUTO is resolved by the one-GPU decode, DecodeJob._configure; a sharded decode's plan decides the forms itself and wants the
    # exact transform mode here)
    p.hilbert_mode = nat.WFX_HILBERT_FFT if hilbert_mode == HILBERT_AUTO else hilbert_mode
    p.shard_plan = int(shard_plan)       # sharded decodes only: 0 cost model, 1 distributed, 2 single; + 16 rows layout (include/wefax_hip.h)
    lo0, lo1, glo = hp.percentile_plan(n, 0.5)                         # wefax.py:194-196
    hi0, hi1, ghi = hp.percentile_plan(n, 99.5)
    p.rank_lo[:] = [lo0, lo1]
    p.rank_hi[:] = [hi0, hi1]
    p.gamma_lo, p.gamma_hi = glo, ghi
    n1, n0g, mind = hp.sync_constants(hp.TARGET_RATE, frame_len)
    p.n1, p.n0_gap, p.mindistance = n1, n0g, mind
    p.frame_samples = frame_len * hp.TARGET_RATE                       # wefax.py:265-266
    p.width = int(frame_len * hp.TARGET_RATE)                          # wefax.py:298
    return p, {"input_length": input_length, "resampled": resampled, "n": n, "length": n / hp.TARGET_RATE}   # wefax.py:393


# Idle contexts, per device.  Creating a context (stream, pinned mirrors, first allocations: 2-8 ms) and destroying one (10 ms of
# hipFree) cost more than decoding a ten-minute capture (0.34 ms), and a service makes one Demodulator per file: a Demodulator
# takes an idle context when it starts processing and hands it back when it is closed or collected.  Buffers, transform plans and
# filter tables stay with the context.  WFX_CTX_POOL=<n> idle contexts are kept per device (default 2, 0 = none).
_POOL_LOCK = threading.Lock()
_POOL: dict = {}


def _device_index(device) -> int:
    """None = the process's default device, as nat.Context resolves it."""
    return int(os.environ.get("WEFAX_DEVICE", os.environ.get("LOCAL_RANK", "0"))) if device is None else int(device)


def _acquire_context(device: int):
    with _POOL_LOCK:
        idle = _POOL.get(device)
        if idle:
            return idle.pop()
    return nat.Context(device)


def _release_context(ctx, device: int):
    keep = int(os.environ.get("WFX_CTX_POOL", "2"))
    with _POOL_LOCK:
        idle = _POOL.setdefault(device, [])
        if len(idle) < keep:
            idle.append(ctx)
            return
    ctx.close()


def release_contexts():
    """Destroy the idle contexts (and with them the device memory they hold)."""
    with _POOL_LOCK:
        idle = [c for lst in _POOL.values() for c in lst]
        _POOL.clear()
    for c in idle:
        c.close()


class _Shape:
    """What process() asks of the samples before they are decoded (their count and whether there are two channels), for a file
    whose samples it never holds (DecodeJob.from_wav)."""

    def __init__(self, frames: int, channels: int):
        self.ndim = 2 if channels > 1 else 1
        self._n = frames

    def __len__(self):
        return self._n


_FE_CACHE: dict = {}


def _front_end_for(sample_rate: int):
    """polyphase.FrontEnd per input rate (the least-squares designs of the 1.536 MS/s pair take 2 s: once per process)."""
    from . import polyphase
    if sample_rate not in _FE_CACHE:
        if sample_rate != int(sample_rate):
            raise ValueError("not an integer rate")
        _FE_CACHE[sample_rate] = polyphase.FrontEnd(int(sample_rate))
    return _FE_CACHE[sample_rate]


class DecodeJob:
    """One capture resident on the GPU: upload once, run the path any number of
    times (bench.py times ``run()``), then read results."""

    def __init__(self, ctx: nat.Context, data: np.ndarray, sample_rate: int,
                 lines_per_minute: int = 120, notch=hp.DEFAULT_NOTCH,
                 hilbert_mode: int = DEFAULT_HILBERT_MODE):
        self.ctx = ctx
        self.frame_len = 1 / (lines_per_minute / 60)                       # wefax.py:33
        data = np.asarray(data)
        self.merged_on_host = False
        self._in_shape, self._in_dtype = tuple(data.shape), data.dtype
        data, kind, ext = self._host_form(data)
        self._configure(kind, int(data.shape[0]), sample_rate, notch, hilbert_mode)
        if ext is not None and not self.resampled:
            self.params.has_ext = 1
            self.params.ext_left[:] = [float(v) for v in ext[0]]
            self.params.ext_right[:] = [float(v) for v in ext[1]]
        ctx.decode_upload(data, self.params)
        self.info = None

    def _host_form(self, data: np.ndarray):
        """What reaches the device for a capture as read from the wav: (array, in_kind, odd extension or None)."""
        ext = None
        if data.ndim == 2:
            if data.dtype == np.int16:
                kind = nat.WFX_IN_I16_STEREO
                data = np.ascontiguousarray(data[:, :2])
            elif data.dtype in nat.STEREO_KIND_OF:
                # uint8 / int32 / float32 wavs: merged ON THE DEVICE with numpy's scalar semantics of wefax.py:372 (the add wraps
                # in the file's dtype; float32 stays float32).  A float32 wav's merged list is float32 when filtfilt extends it
                # at its two ends (wefax.py:72): those 9 + 9 numbers are formed here from the first / last ten frames, in float32
                # (golden stereo_f32_240: the first audio sample is off by 1.7e-10 otherwise)
                kind = nat.STEREO_KIND_OF[data.dtype]
                data = np.ascontiguousarray(data[:, :2])
                if data.dtype == np.float32 and data.shape[0] > 9:
                    with np.errstate(over="ignore"):
                        head = np.divide(np.add(data[:10, 0], data[:10, 1]), 2)
                        tail = np.divide(np.add(data[-10:, 0], data[-10:, 1]), 2)
                    ext = (hp.odd_extension(head)[0], hp.odd_extension(tail)[1])
            else:   # anything else scipy might hand over (float64 stereo): numpy on the host
                with np.errstate(over="ignore"):
                    data = np.divide(np.add(data[:, 0], data[:, 1]), 2).astype(np.float64)
                kind = nat.WFX_IN_F64_MONO
                self.merged_on_host = True
        elif data.dtype == np.int16:
            kind = nat.WFX_IN_I16_MONO
        else:
            # uint8 / int32 / float32 captures: the arithmetic runs on a float64 copy, but filtfilt's odd extension is what scipy
            # computes in the file's own dtype (wrapping / float32 rounding): evaluated here, handed over as 9 + 9 numbers
            ext = hp.odd_extension(data) if (data.dtype != np.float64 and data.shape[0] > 9) else None
            data = data.astype(np.float64)
            kind = nat.WFX_IN_F64_MONO
        return data, kind, ext

    def _configure(self, kind, n0, sample_rate, notch, hilbert_mode, n_out=None):
        """The scalar arithmetic of the reference (lengths, notch, percentile ranks, sync constants) -> self.params."""
        p, meta = build_params(kind, n0, sample_rate, self.frame_len, notch, hilbert_mode, n_out)
        p.hilbert_mode = resolve_hilbert_mode(hilbert_mode, meta["n"], meta["resampled"], n0)
        self.hilbert_mode = int(p.hilbert_mode)
        self.input_length, self.resampled = meta["input_length"], meta["resampled"]
        self.n0, self.n = n0, meta["n"]
        self.sample_rate = hp.TARGET_RATE
        self.length = meta["length"]
        self.params = p
        self.width = p.width

    @classmethod
    def from_wav(cls, ctx: nat.Context, path: str, layout, lines_per_minute: int = 120, notch=hp.DEFAULT_NOTCH,
                 hilbert_mode: int = DEFAULT_HILBERT_MODE):
        """A 16-bit PCM wav (``layout`` = hostparams.wav_pcm16_layout(path)) read and uploaded as ONE pipeline: the copy out of the
        page cache and the DMA overlap slice by slice (wefax.py:349 + the upload; include/wefax_hip.h: wfx_decode_upload_fd).
        Same state as ``DecodeJob(ctx, read_wav(path)[1], ...)``."""
        rate, ch, body, frames = layout
        job = object.__new__(cls)
        job.ctx = ctx
        job.frame_len = 1 / (lines_per_minute / 60)
        job.merged_on_host = False
        job._in_shape, job._in_dtype = ((frames, 2) if ch == 2 else (frames,)), np.dtype(np.int16)
        kind = nat.WFX_IN_I16_STEREO if ch == 2 else nat.WFX_IN_I16_MONO
        job._configure(kind, frames, rate, notch, hilbert_mode)
        with open(path, "rb") as fh:
            ctx.decode_upload_fd(fh.fileno(), body, frames * ch * 2, job.params)
        job.info = None
        return job

    @classmethod
    def from_device(cls, ctx: nat.Context, dev_ptr: int, n: int, lines_per_minute: int = 120, notch=hp.DEFAULT_NOTCH,
                    hilbert_mode: int = DEFAULT_HILBERT_MODE, sample_rate: int = hp.TARGET_RATE,
                    n_out: int | None = None):
        """Decode ``n`` float64 samples at ``sample_rate`` (11 025 Hz, or a rate the exact resampler brings there) that
        already sit in device memory (e.g. the output of the time-domain front end, wefax_amd/polyphase.py): the same
        fused path, nothing uploaded.  The memory stays owned by the caller and must outlive the job."""
        job = object.__new__(cls)
        job.ctx = ctx
        job.frame_len = 1 / (lines_per_minute / 60)
        job.merged_on_host = False
        job._in_shape = job._in_dtype = None
        job._configure(nat.WFX_IN_F64_MONO, int(n), sample_rate, notch, hilbert_mode, n_out)
        ctx.decode_attach(int(dev_ptr), job.params)
        job.info = None
        return job

    def reload(self, data: np.ndarray):
        """Another capture of the same shape and dtype into the same job (no plan or buffer is rebuilt).  From an array in
        pinned memory (``_native.pinned_empty``) the copy is a DMA enqueued on the stream."""
        data = np.asarray(data)
        if getattr(self, "_in_shape", None) is None:
            raise ValueError("reload: this job decodes caller-owned device memory (from_device)")
        if tuple(data.shape) != self._in_shape or data.dtype != self._in_dtype:
            raise ValueError(f"reload: the job was built for a {self._in_dtype} capture of shape {self._in_shape}, "
                             f"got {data.dtype} {tuple(data.shape)}")
        data, _, ext = self._host_form(data)
        self.ctx.decode_reload(data, ext if (ext is not None and not self.resampled) else None)
        self.info = None

    def fetch_image_async(self, out: np.ndarray):
        """Enqueue the copy of the image into ``out`` (uint8, at least 4 * width * (n // width) bytes, ideally pinned); the rows
        that exist are known with ``result()``: ``out.reshape(-1)[:4 * info.height * info.width]``."""
        self.ctx.decode_fetch_async(nat.WFX_BUF_IMAGE, out)

    def run(self):
        """Enqueue the whole path (asynchronous)."""
        self.ctx.decode_run()
        self.info = None

    def result(self) -> nat.DecodeInfo:
        if self.info is None:
            self.info = self.ctx.decode_result()
        return self.info

    def fetch(self, which: str) -> np.ndarray:
        info = self.result()
        if which == "audio":
            return self.ctx.decode_fetch(nat.WFX_BUF_AUDIO, (self.n,), np.float64)
        if which == "envelope":
            return self.ctx.decode_fetch(nat.WFX_BUF_ENVELOPE, (self.n,), np.float64)
        if which == "digitalized":
            return self.ctx.decode_fetch(nat.WFX_BUF_DIGITAL, (self.n,), np.uint8)
        if which == "image":
            return self.ctx.decode_fetch(nat.WFX_BUF_IMAGE, (4 * info.height, info.width), np.uint8)
        raise KeyError(which)


_PNG_FALLBACK_SAID = False


class Demodulator:
    def __init__(self, filepath: str,
                 lines_per_minute: int = 120,
                 quiet: bool = False,
                 tcp_stream: bool = True,
                 device: int | None = None,
                 hilbert_mode: int = DEFAULT_HILBERT_MODE,
                 front_end: str | None = None):
        """``front_end`` (not in the reference; default ``"exact"``, or the environment's WEFAX_FRONT_END): how a capture that is not
        at 11 025 Hz gets there.  ``"exact"``: the reference's own operators -- merge, then ONE FFT resample over the whole capture
        (wefax.py:351-394), bit-identical to it.  ``"time-domain"``: int16 captures at integer rates >= 28 kHz go through the
        halo-local decimator chain of wefax_amd/polyphase.py (1.536 MS/s: / 32 integer-exact and / 3 float64 in one streaming
        kernel) down to a hand-over rate, where the exact FFT resampler takes the last step -- the form BASELINE configs[3] is
        timed on and the one that shards; uint8 stream and image within 1 grey level of the reference's (measured: identical on
        98 % of random clips, tests/test_polyphase.py).  Captures that form cannot keep the reference's sampling grid for (other
        dtypes, rates, lengths that are not a whole number of hand-over samples) take the exact route either way."""
        if not os.path.exists(filepath):                                   # wefax.py:24-25
            raise Exception(f"INVALID FILE: file at path: {filepath} does not exist")
        if filepath.split('.')[-1] != 'wav':                               # wefax.py:27-28
            raise Exception("INVALID FILETYPE: only .wav files are supported at this moment")
        self.filepath = filepath
        self.filename = self.filepath.split('/')[-1]
        self.lines_per_minute = lines_per_minute
        self.time_for_one_frame = 1 / (self.lines_per_minute / 60)  # in s
        self.quiet = quiet
        self.stream = tcp_stream
        self.websocket_stack = []
        self._device = _device_index(device)
        self._hilbert_mode = hilbert_mode
        front_end = front_end or os.environ.get("WEFAX_FRONT_END", "exact")
        if front_end not in ("exact", "time-domain"):
            raise ValueError(f"front_end: 'exact' or 'time-domain', not {front_end!r}")
        self.front_end = front_end
        self.front_end_used = None          # after process(): which of the two ran
        self._fe_dec = None
        self._ctx = None
        self._job = None
        self._cache = {}
        if not self.quiet:
            print("#" * 10 + ' ' * 5 + str(self.filename).ljust(20) + ' ' * 5 + "#" * 10)

    def update_lines_per_minute(self, lpm):                                # wefax.py:42-44
        self.lines_per_minute = lpm
        self.time_for_one_frame = 1 / (lpm / 60)  # in s

    # ------------------------------------------------------------------ helpers
    def _say(self, text):
        if not self.quiet:
            print(text)

    def _send_websocket_packet(self, message: dict):                       # wefax.py:396-397
        self.websocket_stack.append(message)

    def _progress(self, title, percentage):
        if self.stream:
            self._send_websocket_packet({"data_type": "progress_bar",
                                         "progress_title": title,
                                         "percentage": percentage})

    # ------------------------------------------------------------------ process
    def process(self):
        """wefax.py:46-93.  Raises what the reference raises: ValueError when no
        phasing group closes (wefax.py:294) or when the envelope is constant
        (int(nan), wefax.py:216)."""
        self._cache = {}
        self._image_shape = None
        if self._ctx is None:
            self._ctx = _acquire_context(self._device)
        # the samples go from the page cache into the context's page-locked staging buffer (a few threads) and from there to the
        # device by DMA; `data` is a view of that buffer, used before process() returns.  A 16-bit PCM file on the exact route takes
        # the pipelined form of the two (DecodeJob.from_wav): only its header is read here
        layout = hp.wav_pcm16_layout(self.filepath) if os.environ.get("WEFAX_UPLOAD_PIPELINE", "1") != "0" else None
        if layout is not None and (self.front_end == "time-domain" and layout[0] != hp.TARGET_RATE or layout[3] < 2):
            layout = None
        if layout is not None:
            sample_rate, data = layout[0], _Shape(layout[3], layout[1])
        else:
            sample_rate, data = hp.read_wav(self.filepath, alloc=self._ctx.staging)    # wefax.py:349
        if data.ndim == 2:                                                  # wefax.py:351-355
            self._say("\033[0;33mWARNING: two channels audio detected. Program will try to merge audio to one channel\033[0m")
            self._say("MERGING AUDIO CHANNELS:")
            parts = len(data)
            if self.stream:
                for p in range(0, parts, 1000):                             # wefax.py:364-370
                    self._progress("merging channels", (p + 1) / parts * 100)
                if (parts - 1) % 1000 != 0:
                    self._progress("merging channels", (parts - 1 + 1) / parts * 100)
        if sample_rate != hp.TARGET_RATE:                                   # wefax.py:376-383
            self._say("\033[0;33mWARNING: audio sample rate is not 11025 samples per second. Program will try to resample audio\033[0m")
            self._say(f"RESAMPLING AUDIO FROM {round(sample_rate / 1000, 2)} KhZ TO 11.025 KHZ:")
            self._progress("resampling audio", 0)

        notch = hp.load_notch_settings()
        if self._fe_dec is not None:
            self._fe_dec.close()
            self._fe_dec = None
        job = self._time_domain_job(data, sample_rate, notch) if (self.front_end == "time-domain" and sample_rate != hp.TARGET_RATE) else None
        self.front_end_used = "time-domain" if job is not None else "exact"
        if job is None:
            if layout is not None:
                job = DecodeJob.from_wav(self._ctx, self.filepath, layout, self.lines_per_minute, notch, self._hilbert_mode)
            else:
                job = DecodeJob(self._ctx, data, sample_rate, self.lines_per_minute, notch,
                                self._hilbert_mode)
            job.run()
        self._job = job
        info = job.result()

        if sample_rate != hp.TARGET_RATE:
            self._progress("resampling audio", 100)                         # wefax.py:386-390
        self.sample_rate, self.length = job.sample_rate, job.length
        self._say(f"{int(notch[0])} {notch[1]} {self.sample_rate}")         # wefax.py:66
        self._say("DEMODULATING SIGNAL:")
        self._progress("demodulating signal", 0)                            # wefax.py:169-181
        self._progress("demodulating signal", 100)
        self._say("DIGITALIZING SIGNAL:")
        self._progress("digitalizing signal", 0)                            # wefax.py:188-214
        self._progress("digitalizing signal", 99)
        self._progress("digitalizing signal", 100)
        if info.nan_count:
            raise ValueError("cannot convert float NaN to integer")        # int(nan), wefax.py:216
        self._say("FINDING SYNC PULSE:")
        n = job.n
        for k in range(1, info.npeaks):                                     # wefax.py:242-246
            self._progress("finding sync pulse", (info.first_pos[k] / n) * 100)
        if info.hit_limit:
            self._progress("finding sync pulse", 100)                       # wefax.py:254-258
        self._low, self._high = info.low, info.high
        self.peaks = [int(info.peak_pos[k]) for k in range(info.npeaks)]
        if info.no_group:
            max([], key=len)            # raises ValueError("max() arg is an empty sequence"), wefax.py:294
        self.phasing_signals = [int(info.phasing[k]) for k in range(info.n_phasing)]
        self.start_frame = self.phasing_signals[-1] if self.phasing_signals else 0   # wefax.py:80
        assert self.start_frame == info.start_frame

        self._say("CONVERTING SIGNAL TO IMAGE:")
        w, h = info.width, info.height
        if h == 0:
            # Image.new('L', (w, 0)).putpixel((0, 0), .) in the reference's loop (wefax.py:304)
            raise IndexError("image index out of range")
        if self.stream:
            for py in range(0, h, 50):                                      # wefax.py:307-313
                self._progress("converting signal to image", (py + 1) / h * 100)
            self._progress("converting signal to image", 100)               # wefax.py:316-322
        # the image stays in HBM until somebody looks at it (``output_array`` / ``output_image`` fetch it on first access;
        # ``save_output_image`` writes the PNG straight from the device)
        self._image_shape = (4 * h, w)
        if self.stream:                                                     # wefax.py:87-90
            self._send_websocket_packet({"data_type": "message",
                                         "message_content": "convert_end"})

    def _time_domain_job(self, data, sample_rate, notch):
        """The opt-in route of ``front_end="time-domain"``: decimator chain on the device, then the fused exact decode attached to its
        output (sharded.FrontEndExactDecoder).  None when the capture is not one this route keeps the reference's grid for."""
        from . import polyphase, sharded
        if data.dtype != np.int16 or data.ndim > 2 or (data.ndim == 2 and data.shape[1] != 2):
            return None
        try:
            fe = _front_end_for(int(sample_rate))
            fe.n_out(int(data.shape[0]))
        except ValueError:
            return None
        if fe.n_target(int(data.shape[0])) < 2:
            return None
        dec = sharded.FrontEndExactDecoder(self._ctx, fe, data, lines_per_minute=self.lines_per_minute, notch=notch, hilbert_mode=self._hilbert_mode)
        self._fe_dec = dec
        dec.run()
        return dec.job

    # stage arrays, copied back on demand -----------------------------------------
    def _lazy(self, key):
        if key not in self._cache:
            if self._job is None:
                raise AttributeError(key)
            self._cache[key] = self._job.fetch(key)
        return self._cache[key]

    @property
    def output_array(self):          # uint8 [4h, w]: the pixels of ``output_image``
        if getattr(self, "_image_shape", None) is None:
            raise AttributeError("output_array")
        return self._lazy("image")

    @property
    def output_image(self):          # wefax.py:85: PIL Image, mode 'L', size (w, 4h)
        if "pil" not in self._cache:
            try:
                from PIL import Image
                self._cache["pil"] = Image.fromarray(self.output_array, mode="L")
            except ImportError:                                             # Pillow optional
                self._cache["pil"] = None
        return self._cache["pil"]

    @property
    def audio_data(self):            # after merge / resample / notch (wefax.py:72)
        return self._lazy("audio")

    @property
    def demodulated_data(self):      # wefax.py:74
        return self._lazy("envelope")

    @property
    def digitalized_data(self):      # wefax.py:76
        return self._lazy("digitalized")

    # ------------------------------------------------------------------ the rest
    def file_info(self):                                                    # wefax.py:342-346
        sample_rate, frames, ch = hp.wav_info(self.filepath)               # (the reference reads the samples for this)
        channels = 2 if ch > 1 else 1       # the reference reports ndim, not the channel count
        length = frames / sample_rate
        return {"filename": self.filename, "channels": channels, "sample_rate": sample_rate, "length": length}

    def show_output_image(self):                                            # wefax.py:403-405
        from matplotlib import pyplot as plt
        plt.imshow(self.output_image, cmap='gray')
        plt.show()

    def save_output_image(self, filepath: str, compress: int | None = None):   # wefax.py:407-408
        """Same picture the reference writes (8-bit gray PNG, w x 4h, identical pixels).  Encoding 27 MB on the host used to
        dominate the whole file-to-file time (PIL: 1.6 s; threaded zlib: 55 ms; the kernels: 0.35 ms), so by default the PNG is
        encoded ON THE DEVICE from the image still resident there (``wfx_decode_save_png_ex``: rows Up-filtered, dynamic-Huffman
        deflate blocks with distance-1 runs, Adler-32 / CRC-32 -- all by kernels; a file within a few percent of zlib level 6's on
        noisy pictures, 5 ms for the 10-minute capture).  WEFAX_PNG_STORED=1 in the environment selects stored (uncompressed)
        deflate blocks instead; ``compress`` = 1..9 (or WEFAX_PNG_COMPRESS) the threaded zlib encoder of wefax_amd/pngio.py on the
        host, ~50 ms.  Other formats go through PIL like the reference."""
        if filepath.lower().endswith(".png") and getattr(self, "_image_shape", None) is not None:
            if compress is None and os.environ.get("WEFAX_PNG_COMPRESS"):
                compress = int(os.environ["WEFAX_PNG_COMPRESS"])
            if not compress and self._ctx is not None and self._job is not None:
                try:
                    self._ctx.decode_save_png(filepath, deflate=os.environ.get("WEFAX_PNG_STORED") != "1")
                    return
                except nat.NativeError as exc:
                    # the context has moved on to another decode (or the device encoder failed): the host copy is encoded instead --
                    # said once per process, so that a broken device encoder does not hide behind a 10x slower save
                    global _PNG_FALLBACK_SAID
                    if not _PNG_FALLBACK_SAID:
                        _PNG_FALLBACK_SAID = True
                        import warnings
                        warnings.warn(f"wefax_amd: device PNG encoder not used ({exc}); falling back to the host encoder", RuntimeWarning, stacklevel=2)
            from .pngio import write_png_gray8
            write_png_gray8(filepath, self.output_array, level=compress or 1)
            return
        if self.output_image is None:
            raise RuntimeError("Pillow is not installed: cannot write " + filepath)
        self.output_image.save(filepath)

    def close(self):
        """Hand the context back (the image, if nobody has looked at it yet, goes with it)."""
        ctx, self._ctx, self._job = self._ctx, None, None
        if self._fe_dec is not None and ctx is not None:
            try:
                ctx.sync()
                self._fe_dec.fe.close()          # (the job inside it belongs to the context)
            except Exception:       # noqa: BLE001 -- a context that is already gone
                pass
        self._fe_dec = None
        if ctx is not None:
            _release_context(ctx, self._device)

    def __del__(self):
        try:
            self.close()
        except Exception:       # interpreter shutdown
            pass


def main(argv=None):
    argv = sys.argv if argv is None else argv
    filename = str(argv[1])
    lpm = int(argv[2])
    output = str(argv[3])
    demodulator = Demodulator(filename, lines_per_minute=lpm, tcp_stream=False, quiet=False)
    for key, value in demodulator.file_info().items():
        print(key, ":", value)
    demodulator.process()
    demodulator.save_output_image(output)


if __name__ == "__main__":
    main()
