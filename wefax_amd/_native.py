"""ctypes binding of libwefax_hip.so (include/wefax_hip.h).

There is NO CPU fallback: if the library is missing or no MI355X is visible the
calls raise.  ctypes releases the GIL for the duration of every call, so a thread
polling ``Demodulator.websocket_stack`` keeps running (main.py:63-83 does that).
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("WFX_LIB") or os.path.join(_HERE, "libwefax_hip.so")     # WFX_LIB: another build of the library (A/B runs)

WFX_IN_I16_MONO, WFX_IN_I16_STEREO, WFX_IN_F64_MONO, WFX_IN_F32_MONO = 0, 1, 2, 3
WFX_IN_U8_STEREO, WFX_IN_I32_STEREO, WFX_IN_F32_STEREO = 4, 5, 6
STEREO_KIND_OF = {np.dtype(np.int16): 1, np.dtype(np.uint8): 4, np.dtype(np.int32): 5, np.dtype(np.float32): 6}
WFX_HILBERT_FFT, WFX_HILBERT_BLUESTEIN, WFX_HILBERT_FFT_POW2, WFX_HILBERT_FMM = 0, 2, 3, 4      # (1 was the truncated FIR mode of rounds 1-2: removed)
WFX_BUF_AUDIO, WFX_BUF_ENVELOPE, WFX_BUF_DIGITAL, WFX_BUF_IMAGE = 0, 1, 2, 3
WFX_MAX_PEAKS = 100

# every symbol include/wefax_hip.h declares (tests check that all are exported)
SYMBOLS = [
    "wfx_device_count", "wfx_create", "wfx_destroy", "wfx_last_error", "wfx_sync",
    "wfx_version", "wfx_device_pci_bus_id", "wfx_merge_channels", "wfx_merge_channels_any", "wfx_resample", "wfx_notch_filtfilt", "wfx_notch_filtfilt_ext",
    "wfx_analytic_env", "wfx_order_stats", "wfx_quantise", "wfx_sync_corr",
    "wfx_sync_peaks", "wfx_lines_to_image", "wfx_packet_process", "wfx_packets_process", "wfx_packet_spectrum", "wfx_decode_upload", "wfx_decode_upload_fd", "wfx_decode_attach", "wfx_decode_run",
    "wfx_decode_result", "wfx_debug_counters", "wfx_decode_bind_image", "wfx_decode_fetch", "wfx_decode_device_ptr",
    "wfx_decode_copy_to_device", "wfx_stream_handle", "wfx_decode_export_async",
    "wfx_dev_malloc", "wfx_dev_free", "wfx_dev_upload", "wfx_dev_download", "wfx_dev_copy",
    "wfx_d_notch_fir", "wfx_d_notch_fir_f64", "wfx_d_decimate_fir64", "wfx_d_decimate_fir64_batch", "wfx_d_ingest_chain", "wfx_d_hilbert_fmm", "wfx_d_resample_fmm", "wfx_plan_resample_direct", "wfx_d_read_rate", "wfx_d_stream_rate", "wfx_d_median5", "wfx_d_select_hist",
    "wfx_d_quantise", "wfx_d_sync_search", "wfx_d_image_rows",
    "wfx_comm_unique_id", "wfx_comm_create", "wfx_comm_create_local", "wfx_comm_create_shm", "wfx_comm_selftest", "wfx_comm_info", "wfx_comm_destroy",
    "wfx_comm_barrier", "wfx_comm_allgather_host",
    "wfx_shard_layout_query", "wfx_shard_dry_run", "wfx_shard_wire_plan", "wfx_comm_wire_reset", "wfx_comm_wire_stats", "wfx_comm_wire_timing", "wfx_comm_wire_times", "wfx_comm_async_exchanges", "wfx_shard_create", "wfx_shard_upload", "wfx_shard_attach", "wfx_shard_phase_count", "wfx_shard_phase",
    "wfx_decode_sharded", "wfx_shard_result", "wfx_shard_fetch", "wfx_shard_destroy",
    "wfx_synth_frames", "wfx_synth_capture", "wfx_decode_png", "wfx_decode_save_png", "wfx_decode_png_ex", "wfx_decode_save_png_ex", "wfx_host_alloc", "wfx_host_free",
    "wfx_decode_reload", "wfx_decode_fetch_async", "wfx_plan_padded_length", "wfx_plan_describe",
    "wfx_timer_start", "wfx_timer_stop", "wfx_profile_enable", "wfx_profile_reset",
    "wfx_profile_kernel_count", "wfx_profile_kernel_name", "wfx_profile_get",
]


class DecodeParams(C.Structure):
    _fields_ = [
        ("in_kind", C.c_int),
        ("n0", C.c_uint64),
        ("n", C.c_uint64),
        ("resample", C.c_int),
        ("notch_b", C.c_double * 3),
        ("notch_a", C.c_double * 3),
        ("hilbert_mode", C.c_int),
        ("shard_plan", C.c_int),
        ("rank_lo", C.c_uint64 * 2),
        ("rank_hi", C.c_uint64 * 2),
        ("gamma_lo", C.c_double),
        ("gamma_hi", C.c_double),
        ("n1", C.c_int),
        ("n0_gap", C.c_int),
        ("mindistance", C.c_int64),
        ("frame_samples", C.c_double),
        ("width", C.c_int),
        ("has_ext", C.c_int),
        ("ext_left", C.c_double * 9),
        ("ext_right", C.c_double * 9),
    ]


class DecodeInfo(C.Structure):
    _fields_ = [
        ("n", C.c_uint64),
        ("low", C.c_double),
        ("high", C.c_double),
        ("nan_count", C.c_uint64),
        ("npeaks", C.c_int),
        ("hit_limit", C.c_int),
        ("no_group", C.c_int),
        ("n_phasing", C.c_int),
        ("start_frame", C.c_int64),
        ("width", C.c_int),
        ("height", C.c_int),
        ("peak_pos", C.c_int64 * (WFX_MAX_PEAKS + 1)),
        ("first_pos", C.c_int64 * (WFX_MAX_PEAKS + 1)),
        ("phasing", C.c_int64 * (WFX_MAX_PEAKS + 1)),
    ]


class SynthParams(C.Structure):
    _fields_ = [
        ("sample_rate", C.c_double),
        ("lines_per_minute", C.c_int),
        ("ioc", C.c_int),
        ("start_tone_s", C.c_double),
        ("phasing_lines", C.c_int),
        ("image_lines", C.c_int),
        ("stop_tone_s", C.c_double),
        ("black_tail_s", C.c_double),
        ("amplitude", C.c_double),
        ("noise", C.c_double),
        ("seed", C.c_uint64),
        ("iq", C.c_int),
    ]


class ShardLayout(C.Structure):
    _fields_ = [
        ("world", C.c_int),
        ("rank", C.c_int),
        ("first_radix", C.c_int * 2),
        ("in_lo", C.c_uint64),
        ("in_hi", C.c_uint64),
        ("own_lo", C.c_uint64),
        ("own_hi", C.c_uint64),
        ("nseg", C.c_int),
        ("in_halo", C.c_int),
        ("in_seg_len", C.c_uint64),
        ("in_seg_stride", C.c_uint64),
        ("own_seg_len", C.c_uint64),
        ("own_seg_stride", C.c_uint64),
        ("plan", C.c_int),
        ("plan_forced", C.c_int),
        ("model_single_s", C.c_double),
        ("model_dist_compute_s", C.c_double),
        ("model_dist_wire_s", C.c_double),
        ("model_wire_bytes", C.c_uint64),
        ("plan_reason", C.c_char * 160),
    ]

    @property
    def in_frames(self) -> int:
        """Frames the rank hands over (segments with their halos, back to back)."""
        if self.plan == 3:      # chunk-local multipole plan: one range with `in_halo` frames round the circle on either side
            return int(self.in_hi - self.in_lo) + 2 * int(self.in_halo)
        return int(self.nseg) * (int(self.in_seg_len) + 2 * int(self.in_halo)) if self.nseg > 1 else int(self.in_hi - self.in_lo)

    @property
    def own_samples(self) -> int:
        return int(self.nseg) * int(self.own_seg_len) if self.nseg > 1 else int(self.own_hi - self.own_lo)

    def own_index(self) -> "np.ndarray":
        """Global sample indices of the rank's own samples, in the order of its stage buffers."""
        if self.nseg <= 1:
            return np.arange(int(self.own_lo), int(self.own_hi), dtype=np.int64)
        s = np.arange(int(self.nseg), dtype=np.int64)[:, None] * int(self.own_seg_stride)
        return (int(self.own_lo) + s + np.arange(int(self.own_seg_len), dtype=np.int64)[None, :]).reshape(-1)

    def in_index(self) -> "np.ndarray":
        """Global frame indices of the frames the rank hands over (may lie outside the capture: provide anything there)."""
        if self.plan == 3:      # (negative / beyond the capture: take them modulo its length -- the halo wraps)
            return np.arange(int(self.in_lo) - int(self.in_halo), int(self.in_hi) + int(self.in_halo), dtype=np.int64)
        if self.nseg <= 1:
            return np.arange(int(self.in_lo), int(self.in_hi), dtype=np.int64)
        h = int(self.in_halo)
        s = np.arange(int(self.nseg), dtype=np.int64)[:, None] * int(self.in_seg_stride)
        return (int(self.in_lo) - h + s + np.arange(int(self.in_seg_len) + 2 * h, dtype=np.int64)[None, :]).reshape(-1)


class WireEntry(C.Structure):
    _fields_ = [("name", C.c_char * 24), ("total_bytes", C.c_uint64), ("max_rank_bytes", C.c_uint64), ("max_link_bytes", C.c_uint64)]


class WireTime(C.Structure):
    _fields_ = [("us", C.c_double), ("wait_us", C.c_double), ("on_comm_stream", C.c_int), ("timed", C.c_int)]


WFX_COMM_ID_BYTES = 128
WFX_ERR_COMM = -5
WFX_ERR_SHORT_FILE = -6


class NativeError(RuntimeError):
    """A libwefax_hip.so call failed (bad argument, HIP error, out of memory)."""


_lib = None


def load():
    """Load libwefax_hip.so; raises if it has not been built (python -m wefax_amd.build)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise NativeError(
            f"{LIB_PATH} is missing: build it with `python -m wefax_amd.build` "
            "(there is no CPU fallback for the WEFAX hot path)")
    lib = C.CDLL(LIB_PATH)
    vp, sz, i, dp = C.c_void_p, C.c_size_t, C.c_int, C.POINTER(C.c_double)
    lib.wfx_device_count.restype = i
    lib.wfx_create.argtypes = [i, i]
    lib.wfx_create.restype = vp
    lib.wfx_destroy.argtypes = [vp]
    lib.wfx_destroy.restype = None
    lib.wfx_last_error.argtypes = [vp]
    lib.wfx_last_error.restype = C.c_char_p
    lib.wfx_version.restype = C.c_char_p
    lib.wfx_sync.argtypes = [vp]
    lib.wfx_device_pci_bus_id.argtypes = [vp, C.c_char_p, i]
    lib.wfx_merge_channels.argtypes = [vp, vp, sz, vp]
    lib.wfx_merge_channels_any.argtypes = [vp, vp, i, sz, vp]
    lib.wfx_resample.argtypes = [vp, vp, sz, sz, vp]
    lib.wfx_notch_filtfilt.argtypes = [vp, vp, i, sz, dp, dp, vp]
    lib.wfx_notch_filtfilt_ext.argtypes = [vp, vp, i, sz, dp, dp, dp, dp, vp]
    lib.wfx_analytic_env.argtypes = [vp, vp, sz, i, vp]
    lib.wfx_order_stats.argtypes = [vp, vp, sz, vp, i, vp]
    lib.wfx_quantise.argtypes = [vp, vp, sz, C.c_double, C.c_double, vp, C.POINTER(C.c_uint64)]
    lib.wfx_sync_corr.argtypes = [vp, vp, sz, i, i, vp]
    lib.wfx_sync_peaks.argtypes = [vp, vp, sz, i, i, C.c_int64, vp, vp, C.POINTER(i), C.POINTER(i)]
    lib.wfx_lines_to_image.argtypes = [vp, vp, sz, sz, i, vp]
    lib.wfx_decode_upload.argtypes = [vp, vp, C.POINTER(DecodeParams)]
    lib.wfx_decode_upload_fd.argtypes = [vp, i, C.c_uint64, vp, sz, C.POINTER(DecodeParams)]
    lib.wfx_decode_run.argtypes = [vp]
    lib.wfx_decode_result.argtypes = [vp, C.POINTER(DecodeInfo)]
    lib.wfx_debug_counters.argtypes = [vp, C.POINTER(C.c_longlong)]
    lib.wfx_decode_bind_image.argtypes = [vp, vp, sz]
    lib.wfx_decode_fetch.argtypes = [vp, i, vp, sz]
    lib.wfx_decode_device_ptr.argtypes = [vp, i, C.POINTER(vp), C.POINTER(sz)]
    lib.wfx_decode_copy_to_device.argtypes = [vp, i, vp, sz, C.POINTER(sz)]
    lib.wfx_packet_process.argtypes = [vp, vp, i, sz, dp, dp, C.POINTER(C.c_uint64), C.c_double, C.c_double, vp,
                                       C.POINTER(C.c_double), C.POINTER(C.c_double)]
    lib.wfx_packet_spectrum.argtypes = [vp, vp, i, sz, vp]
    lib.wfx_packets_process.argtypes = [vp, vp, i, sz, sz, dp, dp, C.POINTER(C.c_uint64), C.c_double, C.c_double, vp, vp, vp]
    lib.wfx_decode_attach.argtypes = [vp, vp, C.POINTER(DecodeParams)]
    lib.wfx_stream_handle.argtypes = [vp, C.POINTER(vp)]
    lib.wfx_decode_export_async.argtypes = [vp, i, vp, sz]
    lib.wfx_dev_malloc.argtypes = [vp, sz, C.POINTER(vp)]
    lib.wfx_dev_free.argtypes = [vp, vp]
    lib.wfx_dev_upload.argtypes = [vp, vp, vp, sz]
    lib.wfx_dev_download.argtypes = [vp, vp, vp, sz]
    lib.wfx_dev_copy.argtypes = [vp, vp, vp, sz]
    lib.wfx_d_notch_fir.argtypes = [vp, vp, sz, dp, dp, vp, i]
    lib.wfx_d_notch_fir_f64.argtypes = [vp, vp, sz, dp, dp, vp, i]
    lib.wfx_d_decimate_fir64.argtypes = [vp, vp, i, sz, C.c_int64, i, vp, i, vp, sz, i, C.POINTER(C.c_int)]
    lib.wfx_d_decimate_fir64_batch.argtypes = [vp, vp, i, sz, C.c_int64, i, vp, i, vp, sz, i, C.POINTER(C.c_int), i, sz, sz]
    lib.wfx_d_hilbert_fmm.argtypes = [vp, vp, sz, vp, i, C.POINTER(C.c_int)]
    lib.wfx_d_resample_fmm.argtypes = [vp, vp, sz, sz, vp, C.POINTER(C.c_int)]
    lib.wfx_plan_resample_direct.argtypes = [C.c_uint64, C.c_uint64]
    lib.wfx_d_read_rate.argtypes = [vp, vp, sz, i, C.POINTER(C.c_double)]
    lib.wfx_d_stream_rate.argtypes = [vp, vp, sz, vp, i, C.POINTER(C.c_double)]
    lib.wfx_d_ingest_chain.argtypes = [vp, vp, i, sz, i, vp, i, i, i, vp, i, vp, sz, i, sz, sz, C.POINTER(C.c_int)]
    lib.wfx_d_median5.argtypes = [vp, vp, sz, vp]
    lib.wfx_d_select_hist.argtypes = [vp, vp, sz, i, C.POINTER(C.c_uint64), vp]
    lib.wfx_d_quantise.argtypes = [vp, vp, sz, C.c_double, C.c_double, vp, C.POINTER(C.c_uint64)]
    lib.wfx_d_sync_search.argtypes = [vp, vp, sz, sz, i, i, C.c_int64, C.c_double, i, C.POINTER(DecodeInfo)]
    lib.wfx_d_image_rows.argtypes = [vp, vp, sz, C.c_uint64, C.c_uint64, i, i, i, i, vp]
    lib.wfx_comm_unique_id.argtypes = [vp]
    lib.wfx_comm_create.argtypes = [vp, vp, i, i, C.POINTER(vp)]
    lib.wfx_comm_create_local.argtypes = [i, C.POINTER(vp)]
    lib.wfx_comm_create_shm.argtypes = [vp, C.c_char_p, i, i, C.c_double, C.POINTER(vp)]
    lib.wfx_comm_selftest.argtypes = [vp, vp, i, C.c_uint64]
    lib.wfx_comm_info.argtypes = [vp, C.POINTER(i), C.POINTER(i), C.POINTER(i)]
    lib.wfx_comm_destroy.argtypes = [vp]
    lib.wfx_comm_barrier.argtypes = [vp, vp]
    lib.wfx_comm_allgather_host.argtypes = [vp, vp, vp, vp, sz]
    lib.wfx_shard_layout_query.argtypes = [C.POINTER(DecodeParams), i, i, C.POINTER(ShardLayout)]
    lib.wfx_shard_dry_run.argtypes = [C.POINTER(DecodeParams), i]
    lib.wfx_shard_wire_plan.argtypes = [C.POINTER(DecodeParams), i, C.POINTER(WireEntry), i]
    lib.wfx_comm_wire_reset.argtypes = [vp]
    lib.wfx_comm_async_exchanges.argtypes = [vp]
    lib.wfx_comm_async_exchanges.restype = C.c_uint64
    lib.wfx_comm_wire_stats.argtypes = [vp, C.POINTER(WireEntry), i]
    lib.wfx_comm_wire_timing.argtypes = [vp, i]
    lib.wfx_comm_wire_times.argtypes = [vp, C.POINTER(WireTime), i]
    lib.wfx_shard_create.argtypes = [vp, vp, C.POINTER(DecodeParams), C.POINTER(vp)]
    lib.wfx_shard_upload.argtypes = [vp, vp]
    lib.wfx_shard_attach.argtypes = [vp, vp]
    lib.wfx_shard_phase_count.argtypes = [vp]
    lib.wfx_shard_phase.argtypes = [vp, i]
    lib.wfx_decode_sharded.argtypes = [vp]
    lib.wfx_shard_result.argtypes = [vp, C.POINTER(DecodeInfo)]
    lib.wfx_shard_fetch.argtypes = [vp, i, vp, sz]
    lib.wfx_shard_destroy.argtypes = [vp]
    lib.wfx_decode_png.argtypes = [vp, C.POINTER(vp), C.POINTER(sz)]
    lib.wfx_decode_save_png.argtypes = [vp, C.c_char_p, C.POINTER(sz)]
    lib.wfx_decode_png_ex.argtypes = [vp, i, C.POINTER(vp), C.POINTER(sz)]
    lib.wfx_decode_save_png_ex.argtypes = [vp, C.c_char_p, i, C.POINTER(sz)]
    lib.wfx_host_alloc.argtypes = [sz]
    lib.wfx_host_alloc.restype = vp
    lib.wfx_host_free.argtypes = [vp]
    lib.wfx_host_free.restype = None
    lib.wfx_decode_reload.argtypes = [vp, vp, sz, C.POINTER(C.c_double), C.POINTER(C.c_double)]
    lib.wfx_decode_fetch_async.argtypes = [vp, i, vp, sz]
    lib.wfx_plan_padded_length.argtypes = [C.c_uint64]
    lib.wfx_plan_padded_length.restype = C.c_uint64
    lib.wfx_plan_describe.argtypes = [C.c_uint64, C.c_char_p, i]
    lib.wfx_synth_frames.argtypes = [C.POINTER(SynthParams)]
    lib.wfx_synth_frames.restype = C.c_uint64
    lib.wfx_synth_capture.argtypes = [vp, C.POINTER(SynthParams), C.c_uint64, C.c_uint64, vp]
    lib.wfx_timer_start.argtypes = [vp]
    lib.wfx_timer_stop.argtypes = [vp, C.POINTER(C.c_float)]
    lib.wfx_profile_enable.argtypes = [vp, i]
    lib.wfx_profile_reset.argtypes = [vp]
    lib.wfx_profile_kernel_count.restype = i
    lib.wfx_profile_kernel_name.argtypes = [i]
    lib.wfx_profile_kernel_name.restype = C.c_char_p
    lib.wfx_profile_get.argtypes = [vp, i, C.POINTER(C.c_uint64), C.POINTER(C.c_double)]
    for name in SYMBOLS:
        fn = getattr(lib, name)
        if fn.restype is C.c_int and name not in ("wfx_device_count", "wfx_profile_kernel_count", "wfx_synth_frames", "wfx_host_alloc", "wfx_host_free", "wfx_plan_padded_length"):
            fn.restype = C.c_int
    _lib = lib
    return lib


def resample_direct(n0: int, num: int) -> bool:
    """True: the transform-based resampler takes n0 -> num with mixed-radix passes; False: it needs the chirp-z form (any length, ~3.5x the time)."""
    return bool(load().wfx_plan_resample_direct(int(n0), int(num)))


def padded_length(min_len: int) -> int:
    """13-smooth transform length for an any-length analytic-signal convolution (include/wefax_hip.h wfx_plan_padded_length)."""
    return int(load().wfx_plan_padded_length(int(min_len)))


def plan_describe(length: int) -> str:
    buf = C.create_string_buffer(256)
    load().wfx_plan_describe(int(length), buf, 256)
    return buf.value.decode()


def _ptr(a: np.ndarray):
    return a.ctypes.data_as(C.c_void_p)


class Context:
    """One opaque native context (device buffers + stream) per decoder."""

    def __init__(self, device: int | None = None):
        self.lib = load()
        if device is None:
            device = int(os.environ.get("WEFAX_DEVICE", os.environ.get("LOCAL_RANK", "0")))
        self.device = device
        self.h = self.lib.wfx_create(device, 0)
        if not self.h:
            msg = self.lib.wfx_last_error(None)
            raise NativeError("wfx_create failed: " + (msg.decode() if msg else "unknown error"))

    def close(self):
        if getattr(self, "h", None):
            self.lib.wfx_destroy(self.h)
            self.h = None
        self._staging = None
        self._keep = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def _check(self, rc: int):
        if rc != 0:
            msg = self.lib.wfx_last_error(self.h)
            raise NativeError(f"libwefax_hip error {rc}: {msg.decode() if msg else ''}")

    # ---- stage entry points -------------------------------------------------
    def merge_channels(self, lr: np.ndarray) -> np.ndarray:
        """wefax.py:360-373 on the first two channels, in the file's own sample format (int16 / uint8 / int32 / float32)."""
        lr = np.asarray(lr)
        if lr.dtype not in STEREO_KIND_OF:
            lr = lr.astype(np.int16)
        lr = np.ascontiguousarray(lr[:, :2])
        out = np.empty(lr.shape[0], dtype=np.float64)
        if lr.dtype == np.int16:
            self._check(self.lib.wfx_merge_channels(self.h, _ptr(lr), lr.shape[0], _ptr(out)))
        else:
            self._check(self.lib.wfx_merge_channels_any(self.h, _ptr(lr), STEREO_KIND_OF[lr.dtype], lr.shape[0], _ptr(out)))
        return out

    def resample(self, x: np.ndarray, num: int) -> np.ndarray:
        x = np.ascontiguousarray(x, dtype=np.float64)
        out = np.empty(num, dtype=np.float64)
        self._check(self.lib.wfx_resample(self.h, _ptr(x), x.shape[0], num, _ptr(out)))
        return out

    def notch_filtfilt(self, x: np.ndarray, b, a, ext=None) -> np.ndarray:
        """``ext``: (left[9], right[9]) of filtfilt's odd extension when it was evaluated in the capture's own dtype."""
        if x.dtype == np.int16:
            x = np.ascontiguousarray(x)
            kind = WFX_IN_I16_MONO
        else:
            x = np.ascontiguousarray(x, dtype=np.float64)
            kind = WFX_IN_F64_MONO
        bb = (C.c_double * 3)(*[float(v) for v in b])
        aa = (C.c_double * 3)(*[float(v) for v in a])
        out = np.empty(x.shape[0], dtype=np.float64)
        if ext is not None:
            el = (C.c_double * 9)(*[float(v) for v in ext[0]])
            er = (C.c_double * 9)(*[float(v) for v in ext[1]])
            self._check(self.lib.wfx_notch_filtfilt_ext(self.h, _ptr(x), kind, x.shape[0], bb, aa, el, er, _ptr(out)))
        else:
            self._check(self.lib.wfx_notch_filtfilt(self.h, _ptr(x), kind, x.shape[0], bb, aa, _ptr(out)))
        return out

    def analytic_env(self, x: np.ndarray, mode: int = WFX_HILBERT_FFT) -> np.ndarray:
        x = np.ascontiguousarray(x, dtype=np.float64)
        out = np.empty(x.shape[0], dtype=np.float64)
        self._check(self.lib.wfx_analytic_env(self.h, _ptr(x), x.shape[0], mode, _ptr(out)))
        return out

    def order_stats(self, env: np.ndarray, ranks) -> np.ndarray:
        env = np.ascontiguousarray(env, dtype=np.float64)
        r = np.ascontiguousarray(ranks, dtype=np.uint64)
        out = np.empty(r.shape[0], dtype=np.float64)
        self._check(self.lib.wfx_order_stats(self.h, _ptr(env), env.shape[0], _ptr(r), r.shape[0], _ptr(out)))
        return out

    def quantise(self, env: np.ndarray, low: float, high: float):
        env = np.ascontiguousarray(env, dtype=np.float64)
        out = np.empty(env.shape[0], dtype=np.uint8)
        nan = C.c_uint64(0)
        self._check(self.lib.wfx_quantise(self.h, _ptr(env), env.shape[0], low, high, _ptr(out), C.byref(nan)))
        return out, int(nan.value)

    def sync_corr(self, d: np.ndarray, n1: int, n0: int) -> np.ndarray:
        d = np.ascontiguousarray(d, dtype=np.uint8)
        ncorr = max(0, d.shape[0] - (2 * n1 + n0))
        out = np.empty(ncorr, dtype=np.int32)
        self._check(self.lib.wfx_sync_corr(self.h, _ptr(d), d.shape[0], n1, n0, _ptr(out)))
        return out

    def sync_peaks(self, d: np.ndarray, n1: int, n0: int, mindistance: int):
        d = np.ascontiguousarray(d, dtype=np.uint8)
        pos = np.zeros(WFX_MAX_PEAKS + 1, dtype=np.int64)
        first = np.zeros(WFX_MAX_PEAKS + 1, dtype=np.int64)
        k, hit = C.c_int(0), C.c_int(0)
        self._check(self.lib.wfx_sync_peaks(self.h, _ptr(d), d.shape[0], n1, n0, mindistance,
                                            _ptr(pos), _ptr(first), C.byref(k), C.byref(hit)))
        return pos[:k.value].tolist(), first[:k.value].tolist(), bool(hit.value)

    def lines_to_image(self, d: np.ndarray, start: int, w: int) -> np.ndarray:
        d = np.ascontiguousarray(d, dtype=np.uint8)
        h = (d.shape[0] - start) // w
        img = np.empty((4 * h, w), dtype=np.uint8)
        self._check(self.lib.wfx_lines_to_image(self.h, _ptr(d), d.shape[0], start, w, _ptr(img)))
        return img

    # ---- fused decode ---------------------------------------------------------
    def staging(self, nbytes: int) -> np.ndarray:
        """This context's page-locked staging buffer (uint8, at least ``nbytes``; grown when needed, freed with the context):
        where ``hostparams.read_wav`` puts a file's samples so that the upload is a DMA.  Its contents are valid until the next
        call."""
        buf = getattr(self, "_staging", None)
        if buf is None or buf.nbytes < nbytes:
            self._staging = None
            buf = self._staging = pinned_empty((max(int(nbytes) * 5 // 4, 1 << 20),), np.uint8)
        return buf

    def decode_upload(self, data: np.ndarray, params: DecodeParams):
        data = np.ascontiguousarray(data)
        self._keep = data
        self._check(self.lib.wfx_decode_upload(self.h, _ptr(data), C.byref(params)))

    def decode_upload_fd(self, fd: int, file_offset: int, nbytes: int, params: DecodeParams):
        """The capture straight from an open file (16-bit PCM): slices read into this context's staging buffer by a few threads, each
        on its way to the device as soon as it is complete (include/wefax_hip.h: wfx_decode_upload_fd)."""
        buf = self.staging(nbytes)
        self._keep = buf
        rc = self.lib.wfx_decode_upload_fd(self.h, int(fd), int(file_offset), _ptr(buf), buf.nbytes, C.byref(params))
        if rc == WFX_ERR_SHORT_FILE:        # the same exception as hostparams.read_wav (= scipy.io.wavfile.read, wefax.py:349)
            msg = self.lib.wfx_last_error(self.h)
            raise ValueError(msg.decode() if msg else "Incomplete wav file")
        self._check(rc)

    def decode_run(self):
        self._check(self.lib.wfx_decode_run(self.h))

    def decode_result(self) -> DecodeInfo:
        info = DecodeInfo()
        self._check(self.lib.wfx_decode_result(self.h, C.byref(info)))
        return info

    def decode_reload(self, data: np.ndarray, ext=None):
        """A new capture of the same description into the context (DMA when ``data`` is pinned: ``pinned_empty``).  The library
        checks the byte count against the uploaded capture's; ``ext`` = (left, right) odd-extension values for float64
        hand-overs of uint8 / int32 / float32 files, None when the capture has none."""
        data = np.ascontiguousarray(data)
        self._keep = data
        if ext is None:
            self._check(self.lib.wfx_decode_reload(self.h, _ptr(data), data.nbytes, None, None))
        else:
            left = (C.c_double * 9)(*[float(v) for v in ext[0]])
            right = (C.c_double * 9)(*[float(v) for v in ext[1]])
            self._check(self.lib.wfx_decode_reload(self.h, _ptr(data), data.nbytes, left, right))

    def decode_fetch_async(self, buffer_id: int, out: np.ndarray):
        """Enqueue the copy of a stage buffer into ``out`` (pinned) without waiting; ``sync()`` / ``decode_result()`` waits."""
        self._check(self.lib.wfx_decode_fetch_async(self.h, buffer_id, _ptr(out), out.nbytes))

    def decode_png(self, deflate: bool = False) -> bytes:
        """The PNG file of the decode's image (8-bit gray), assembled on the device: stored deflate blocks, or with ``deflate`` the
        compressed form (Up filter + dynamic-Huffman blocks encoded by kernels)."""
        p, n = C.c_void_p(0), C.c_size_t(0)
        self._check(self.lib.wfx_decode_png_ex(self.h, 1 if deflate else 0, C.byref(p), C.byref(n)))
        return C.string_at(p.value, n.value)

    def decode_save_png(self, path: str, deflate: bool = False) -> int:
        n = C.c_size_t(0)
        self._check(self.lib.wfx_decode_save_png_ex(self.h, os.fsencode(path), 1 if deflate else 0, C.byref(n)))
        return int(n.value)

    def decode_bind_image(self, dst_ptr: int, capacity: int):
        """The next decodes write {header, image} straight to this device address (0 unbinds)."""
        self._check(self.lib.wfx_decode_bind_image(self.h, C.c_void_p(dst_ptr or None), capacity))

    def debug_counters(self):
        """Diagnostics of the last decode: [7] = form of the peak scan (1 joined segments, -1 sequential after a failed join, 0 sequential)."""
        out = (C.c_longlong * 8)()
        self._check(self.lib.wfx_debug_counters(self.h, out))
        return list(out)

    def decode_fetch(self, buffer_id: int, shape, dtype) -> np.ndarray:
        out = np.empty(shape, dtype=dtype)
        self._check(self.lib.wfx_decode_fetch(self.h, buffer_id, _ptr(out), out.nbytes))
        return out

    def decode_device_ptr(self, buffer_id: int):
        p, nb = C.c_void_p(0), C.c_size_t(0)
        self._check(self.lib.wfx_decode_device_ptr(self.h, buffer_id, C.byref(p), C.byref(nb)))
        return p.value, nb.value

    def decode_copy_to_device(self, buffer_id: int, dst_dev_ptr: int, capacity: int) -> int:
        nb = C.c_size_t(0)
        self._check(self.lib.wfx_decode_copy_to_device(self.h, buffer_id, C.c_void_p(dst_dev_ptr), capacity, C.byref(nb)))
        return nb.value

    # ---- device-resident building blocks (wefax_amd/sharded.py) ----------------------
    def dev_malloc(self, nbytes: int) -> int:
        p = C.c_void_p(0)
        self._check(self.lib.wfx_dev_malloc(self.h, nbytes, C.byref(p)))
        return p.value

    def dev_free(self, ptr: int):
        self._check(self.lib.wfx_dev_free(self.h, C.c_void_p(ptr)))

    def dev_upload(self, ptr: int, a: np.ndarray):
        a = np.ascontiguousarray(a)
        self._check(self.lib.wfx_dev_upload(self.h, C.c_void_p(ptr), _ptr(a), a.nbytes))

    def dev_download(self, ptr: int, shape, dtype) -> np.ndarray:
        out = np.empty(shape, dtype=dtype)
        self._check(self.lib.wfx_dev_download(self.h, _ptr(out), C.c_void_p(ptr), out.nbytes))
        return out

    def packet_process(self, samples: np.ndarray, b, a, ranks, gamma_lo: float, gamma_hi: float):
        """One audio packet of the live path (data_packet.py:408-464): (uint8 samples, low, high)."""
        x = np.ascontiguousarray(samples)
        if x.dtype == np.int16:
            kind = WFX_IN_I16_MONO
        else:
            x = np.ascontiguousarray(x, dtype=np.float64)
            kind = WFX_IN_F64_MONO
        out = np.empty(x.shape[0], dtype=np.uint8)
        bb = (C.c_double * 3)(*[float(v) for v in b])
        aa = (C.c_double * 3)(*[float(v) for v in a])
        rr = (C.c_uint64 * 4)(*[int(v) for v in ranks])
        lo, hi = C.c_double(), C.c_double()
        self._check(self.lib.wfx_packet_process(self.h, _ptr(x), kind, x.shape[0], bb, aa, rr, gamma_lo, gamma_hi, _ptr(out),
                                                C.byref(lo), C.byref(hi)))
        return out, lo.value, hi.value

    def packets_process(self, samples2d: np.ndarray, b, a, ranks, gamma_lo: float, gamma_hi: float):
        """[count, n] packets decoded back to back: (uint8 [count, n], low [count], high [count])."""
        x = np.ascontiguousarray(samples2d)
        if x.ndim != 2:
            raise ValueError("packets_process expects a [count, n] array")
        if x.dtype == np.int16:
            kind = WFX_IN_I16_MONO
        else:
            x = np.ascontiguousarray(x, dtype=np.float64)
            kind = WFX_IN_F64_MONO
        count, n = x.shape
        out = np.empty((count, n), dtype=np.uint8)
        lo, hi = np.empty(count), np.empty(count)
        bb = (C.c_double * 3)(*[float(v) for v in b])
        aa = (C.c_double * 3)(*[float(v) for v in a])
        rr = (C.c_uint64 * 4)(*[int(v) for v in ranks])
        self._check(self.lib.wfx_packets_process(self.h, _ptr(x), kind, n, count, bb, aa, rr, gamma_lo, gamma_hi, _ptr(out), _ptr(lo), _ptr(hi)))
        return out, lo, hi

    def packet_spectrum(self, samples: np.ndarray) -> np.ndarray:
        """|FFT(samples)[:n // 2] / (n // 2)| (data_packet.py:388-406, before the normalisation)."""
        x = np.ascontiguousarray(samples)
        if x.dtype == np.int16:
            kind = WFX_IN_I16_MONO
        else:
            x = np.ascontiguousarray(x, dtype=np.float64)
            kind = WFX_IN_F64_MONO
        out = np.empty(x.shape[0] // 2, dtype=np.float64)
        self._check(self.lib.wfx_packet_spectrum(self.h, _ptr(x), kind, x.shape[0], _ptr(out)))
        return out

    def decode_attach(self, dev_ptr: int, params: "DecodeParams"):
        """Fused decode of a capture that already sits in device memory (caller-owned; nothing is copied)."""
        self._check(self.lib.wfx_decode_attach(self.h, C.c_void_p(dev_ptr), C.byref(params)))

    def stream_handle(self) -> int:
        """The context's hipStream_t as an integer (e.g. for torch.cuda.ExternalStream)."""
        out = C.c_void_p()
        self._check(self.lib.wfx_stream_handle(self.h, C.byref(out)))
        return int(out.value or 0)

    def decode_export_async(self, buffer_id: int, dst_ptr: int, capacity: int):
        """Enqueue {int64 bytes, int64 width} + the stage buffer of the decode in flight into device memory; no wait."""
        self._check(self.lib.wfx_decode_export_async(self.h, buffer_id, C.c_void_p(dst_ptr), capacity))

    def dev_copy(self, dst_ptr: int, src_ptr: int, nbytes: int):
        self._check(self.lib.wfx_dev_copy(self.h, C.c_void_p(dst_ptr), C.c_void_p(src_ptr), nbytes))

    def d_notch_fir(self, in_ptr: int, n: int, b, a, out_ptr: int, edge_flags: int = 0):
        bb = (C.c_double * 3)(*[float(v) for v in b])
        aa = (C.c_double * 3)(*[float(v) for v in a])
        self._check(self.lib.wfx_d_notch_fir(self.h, C.c_void_p(in_ptr), n, bb, aa, C.c_void_p(out_ptr), edge_flags))

    def d_notch_fir_f64(self, in_ptr: int, n: int, b, a, out_ptr: int, edge_flags: int = 0):
        bb = (C.c_double * 3)(*[float(v) for v in b])
        aa = (C.c_double * 3)(*[float(v) for v in a])
        self._check(self.lib.wfx_d_notch_fir_f64(self.h, C.c_void_p(in_ptr), n, bb, aa, C.c_void_p(out_ptr), edge_flags))

    def d_decimate_fir64(self, in_ptr: int, in_kind: int, n_in: int, first: int, factor: int, coef: np.ndarray, out_ptr: int, n_out: int,
                         fix_shift: int = 0, nbatch: int = 1, in_stride: int = 0, out_stride: int = 0) -> bool:
        """float64 taps, float64 result; True when the integer-exact form ran (taps on the grid 2**-fix_shift; include/wefax_hip.h).
        ``nbatch`` > 1: that many equally shaped jobs in one launch, ``in_stride`` frames / ``out_stride`` samples apart."""
        c = np.ascontiguousarray(coef, dtype=np.float64)
        ex = C.c_int(0)
        if nbatch > 1:
            self._check(self.lib.wfx_d_decimate_fir64_batch(self.h, C.c_void_p(in_ptr), in_kind, n_in, first, factor, _ptr(c), c.shape[0],
                                                            C.c_void_p(out_ptr), n_out, int(fix_shift), C.byref(ex), int(nbatch), int(in_stride), int(out_stride)))
        else:
            self._check(self.lib.wfx_d_decimate_fir64(self.h, C.c_void_p(in_ptr), in_kind, n_in, first, factor, _ptr(c), c.shape[0],
                                                      C.c_void_p(out_ptr), n_out, int(fix_shift), C.byref(ex)))
        return bool(ex.value)

    def d_ingest_chain(self, in_ptr: int, in_kind: int, n_in: int, factor: int, coef1: np.ndarray, fix_shift: int, factor2: int, coef2,
                       out_ptr: int, n_out: int, nbatch: int = 1, in_stride: int = 0, out_stride: int = 0) -> bool:
        """The integer-exact ingest (``factor`` 32) and the float64 stage behind it (``factor2`` 2 or 3; 0: none) in one streaming
        kernel (csrc/wfx_ingest.hip); bit-identical to the two ``d_decimate_fir64`` calls it replaces.  False: not a shape that
        kernel takes, nothing was enqueued."""
        c1 = np.ascontiguousarray(coef1, dtype=np.float64)
        c2 = np.ascontiguousarray(coef2 if coef2 is not None else np.zeros(1), dtype=np.float64)
        handled = C.c_int(0)
        self._check(self.lib.wfx_d_ingest_chain(self.h, C.c_void_p(in_ptr), in_kind, n_in, int(factor), _ptr(c1), c1.shape[0], int(fix_shift),
                                                int(factor2), _ptr(c2), c2.shape[0] if factor2 else 0, C.c_void_p(out_ptr), n_out,
                                                int(nbatch), int(in_stride), int(out_stride), C.byref(handled)))
        return bool(handled.value)

    def d_hilbert_fmm(self, x_ptr: int, n: int, out_ptr: int, out_env=False) -> bool:
        """H = imag(scipy.signal.hilbert(x)) (``out_env`` True / 1: |x + i H|; 2: its 5-tap median, wefax.py:174-175) by near field + fast
        multipole far field (csrc/wfx_fmm.hip); False: a length that form does not take, nothing was enqueued."""
        handled = C.c_int(0)
        self._check(self.lib.wfx_d_hilbert_fmm(self.h, C.c_void_p(x_ptr), int(n), C.c_void_p(out_ptr), int(out_env), C.byref(handled)))
        return bool(handled.value)

    def d_resample_fmm(self, x_ptr: int, n0: int, num: int, y_ptr: int) -> bool:
        """scipy.signal.resample(x, num) (wefax.py:160-161) of n0 float64 samples in device memory by the multipole form of the periodic sinc
        sum (csrc/wfx_fmm.hip); False: lengths that form does not take (upsampling, odd counts, short captures), nothing was enqueued."""
        handled = C.c_int(0)
        self._check(self.lib.wfx_d_resample_fmm(self.h, C.c_void_p(x_ptr), int(n0), int(num), C.c_void_p(y_ptr), C.byref(handled)))
        return bool(handled.value)

    def d_read_rate(self, ptr: int, nbytes: int, reps: int = 3) -> float:
        """GB/s of a plain read of device memory (include/wefax_hip.h: wfx_d_read_rate): the box's read-stream ceiling."""
        g = C.c_double(0.0)
        self._check(self.lib.wfx_d_read_rate(self.h, C.c_void_p(ptr), int(nbytes), int(reps), C.byref(g)))
        return float(g.value)

    def d_stream_rate(self, ptr: int, nbytes: int, reps: int = 2, out_ptr: int = 0) -> float:
        """GB/s at which the streaming ingest kernel works through the buffer taken as an IQ capture (include/wefax_hip.h: wfx_d_stream_rate)."""
        g = C.c_double(0.0)
        self._check(self.lib.wfx_d_stream_rate(self.h, C.c_void_p(ptr), int(nbytes), C.c_void_p(out_ptr or None), int(reps), C.byref(g)))
        return float(g.value)

    def dev_malloc_placed(self, nbytes: int, tries: int | None = None, good_gbs: float = 6000.0, probe=None):
        """Device memory for a capture the streaming ingest will read many times (or, with ``probe``, for the buffer it writes): the same
        bytes stream up to 15 % slower through one allocation than through another of the same process (docs/history/EXPERIMENTS_rounds1-5.md §9.2), so up
        to ``tries`` allocations (``WFX_PLACE_TRIES``, default 4 since round 6 -- what bench.py always used, so that the product and the
        bench line agree; 1 = take the first) are held side by side and timed (~4 ms each at 22 GB) -- ``probe(ptr)`` ->
        GB/s, default ``d_stream_rate`` on the allocation as the capture -- until one reaches ``good_gbs``; the best is kept, the others
        are freed.  Returns (pointer, [GB/s of every candidate, in order]); without a probe, captures under 1 GiB are not timed."""
        if tries is None:
            tries = int(os.environ.get("WFX_PLACE_TRIES", "4"))
        if tries <= 1 or (probe is None and nbytes < (1 << 30)):
            return self.dev_malloc(nbytes), []
        if probe is None:
            probe = lambda p: self.d_stream_rate(p, nbytes)      # noqa: E731
        cands, rates = [], []
        for _ in range(tries):
            p = self.dev_malloc(nbytes)
            cands.append(p)
            rates.append(probe(p))
            if rates[-1] >= good_gbs:
                break
        best = max(range(len(cands)), key=lambda k: rates[k])
        for k, p in enumerate(cands):
            if k != best:
                self.dev_free(p)
        return cands[best], rates

    def d_median5(self, in_ptr: int, n: int, out_ptr: int):
        self._check(self.lib.wfx_d_median5(self.h, C.c_void_p(in_ptr), n, C.c_void_p(out_ptr)))

    def d_select_hist(self, env_ptr: int, n: int, level: int, prefixes, hist_ptr: int):
        pf = (C.c_uint64 * 4)(*[int(v) for v in prefixes])
        self._check(self.lib.wfx_d_select_hist(self.h, C.c_void_p(env_ptr), n, level, pf, C.c_void_p(hist_ptr)))

    def d_quantise(self, env_ptr: int, n: int, low: float, high: float, out_ptr: int) -> int:
        nan = C.c_uint64(0)
        self._check(self.lib.wfx_d_quantise(self.h, C.c_void_p(env_ptr), n, low, high, C.c_void_p(out_ptr), C.byref(nan)))
        return int(nan.value)

    def d_sync_search(self, d_ptr: int, n: int, n_total: int, n1: int, n0: int, mindistance: int, frame_samples: float,
                      width: int) -> DecodeInfo:
        info = DecodeInfo()
        self._check(self.lib.wfx_d_sync_search(self.h, C.c_void_p(d_ptr), n, n_total, n1, n0, mindistance, frame_samples,
                                               width, C.byref(info)))
        return info

    def d_image_rows(self, d_ptr: int, n: int, g0: int, start: int, width: int, h_total: int, y0: int, rows: int, img_ptr: int):
        self._check(self.lib.wfx_d_image_rows(self.h, C.c_void_p(d_ptr), n, g0, start, width, h_total, y0, rows,
                                              C.c_void_p(img_ptr)))

    def synth_capture(self, params: "SynthParams", lo: int, hi: int, out_ptr: int):
        """Frames [lo, hi) of the synthetic transmission described by ``params`` into device memory (asynchronous)."""
        self._check(self.lib.wfx_synth_capture(self.h, C.byref(params), lo, hi, C.c_void_p(out_ptr)))

    def sync(self):
        self._check(self.lib.wfx_sync(self.h))

    def pci_bus_id(self) -> str:
        """PCI address of this context's GPU as sysfs names it (include/wefax_hip.h: wfx_device_pci_bus_id)."""
        buf = C.create_string_buffer(32)
        self._check(self.lib.wfx_device_pci_bus_id(self.h, buf, 32))
        return buf.value.decode()

    # ---- measurement ------------------------------------------------------------
    def timer_start(self):
        self._check(self.lib.wfx_timer_start(self.h))

    def timer_stop(self) -> float:
        ms = C.c_float(0)
        self._check(self.lib.wfx_timer_stop(self.h, C.byref(ms)))
        return float(ms.value)

    def profile_enable(self, on: bool):
        self._check(self.lib.wfx_profile_enable(self.h, 1 if on else 0))

    def profile_reset(self):
        self._check(self.lib.wfx_profile_reset(self.h))

    def profile(self) -> dict:
        out = {}
        for k in range(self.lib.wfx_profile_kernel_count()):
            cnt, ms = C.c_uint64(0), C.c_double(0)
            self._check(self.lib.wfx_profile_get(self.h, k, C.byref(cnt), C.byref(ms)))
            if cnt.value:
                out[self.lib.wfx_profile_kernel_name(k).decode()] = (int(cnt.value), float(ms.value))
        return out


def _global_error(lib, rc: int) -> NativeError:
    msg = lib.wfx_last_error(None)
    return NativeError(f"libwefax_hip error {rc}: {msg.decode() if msg else ''}")


def shard_layout(params: DecodeParams, world: int, rank: int) -> ShardLayout:
    """Host-only: how the capture described by ``params`` is cut for ``world`` ranks (no GPU needed)."""
    lib = load()
    out = ShardLayout()
    rc = lib.wfx_shard_layout_query(C.byref(params), world, rank, C.byref(out))
    if rc != 0:
        raise _global_error(lib, rc)
    return out


def _wire_list(arr, n) -> list:
    return [{"name": arr[k].name.decode(), "bytes": int(arr[k].total_bytes), "max_rank_bytes": int(arr[k].max_rank_bytes),
             "max_link_bytes": int(arr[k].max_link_bytes)} for k in range(n)]


def shard_wire_plan(params: DecodeParams, world: int) -> list:
    """Host-only: the collectives of one sharded decode in order, with the bytes they put on the wire (all ranks together, the
    busiest rank, the busiest directed link)."""
    lib = load()
    arr = (WireEntry * 64)()
    n = lib.wfx_shard_wire_plan(C.byref(params), world, arr, 64)
    if n < 0:
        raise _global_error(lib, n)
    return _wire_list(arr, min(n, 64))


def shard_dry_run(params: DecodeParams, world: int):
    """Host-only consistency check of every rank's exchange lists for ``world`` ranks; raises NativeError with the reason."""
    lib = load()
    rc = lib.wfx_shard_dry_run(C.byref(params), world)
    if rc != 0:
        raise _global_error(lib, rc)


def comm_unique_id() -> bytes:
    """Rank 0: the 128-byte RCCL unique id the other ranks need for ``Comm.rccl``."""
    lib = load()
    buf = C.create_string_buffer(WFX_COMM_ID_BYTES)
    rc = lib.wfx_comm_unique_id(buf)
    if rc != 0:
        raise _global_error(lib, rc)
    return buf.raw


class Comm:
    """Communicator handle of the sharded decode (include/wefax_hip.h): RCCL, or all ranks in this process."""

    def __init__(self, handle: int, lib):
        self.h, self.lib = handle, lib
        w, r, k = C.c_int(0), C.c_int(0), C.c_int(0)
        lib.wfx_comm_info(self.h, C.byref(w), C.byref(r), C.byref(k))
        self.world, self.rank, self.is_rccl = w.value, r.value, bool(k.value)
        self.is_shm = False

    def wire_reset(self):
        self.lib.wfx_comm_wire_reset(self.h)

    @property
    def async_exchanges(self) -> int:
        """Exchanges run on the communicator's own stream so far (RCCL only: overlapping the caller's kernels)."""
        return int(self.lib.wfx_comm_async_exchanges(self.h))

    def wire_stats(self) -> list:
        """This rank's collectives since the last reset: name, bytes sent to / received from other ranks, largest message."""
        arr = (WireEntry * 256)()
        n = self.lib.wfx_comm_wire_stats(self.h, arr, 256)
        out = _wire_list(arr, max(0, min(n, 256)))
        for e in out:
            e["sent"], e["received"], e["largest_message"] = e.pop("bytes"), e.pop("max_rank_bytes"), e.pop("max_link_bytes")
        return out

    def wire_timing(self, on: bool = True):
        """Reset the records and bracket every collective from now on with a HIP-event pair (RCCL) or the host clock (shm / local)."""
        self.lib.wfx_comm_wire_timing(self.h, 1 if on else 0)

    def wire_times(self) -> list:
        """Per collective, parallel to ``wire_stats()`` (call after the streams were synchronised): ``us`` the collective itself,
        ``wait_us`` what the compute stream stood still for it, ``hidden_us`` the rest (only an exchange on the communicator's own
        stream can hide anything), ``clock`` = events / host / None (include/wefax_hip.h: wfx_comm_wire_times)."""
        arr = (WireTime * 256)()
        n = max(0, min(self.lib.wfx_comm_wire_times(self.h, arr, 256), 256))
        out = []
        for k in range(n):
            t = arr[k]
            ok = t.timed != 0 and t.us >= 0
            wait = t.wait_us if (ok and t.wait_us >= 0) else None
            out.append({"us": round(t.us, 2) if ok else None, "wait_us": round(wait, 2) if wait is not None else None,
                        "hidden_us": round(max(0.0, t.us - wait), 2) if (ok and wait is not None) else None,
                        "stream": "communicator" if t.on_comm_stream else "context", "clock": {0: None, 1: "events", 2: "host"}[int(t.timed)]})
        return out

    @classmethod
    def rccl(cls, ctx: "Context", unique_id: bytes, world: int, rank: int) -> "Comm":
        if len(unique_id) != WFX_COMM_ID_BYTES:
            raise ValueError("RCCL unique id must be 128 bytes")
        h = C.c_void_p(0)
        ctx._check(ctx.lib.wfx_comm_create(ctx.h, unique_id, world, rank, C.byref(h)))
        return cls(h.value, ctx.lib)

    @classmethod
    def local(cls, world: int):
        """``world`` communicators whose ranks all live in this process (emulation on one GPU, tests)."""
        lib = load()
        arr = (C.c_void_p * world)()
        rc = lib.wfx_comm_create_local(world, arr)
        if rc != 0:
            raise _global_error(lib, rc)
        return [cls(arr[r], lib) for r in range(world)]

    @classmethod
    def shm(cls, ctx: "Context | None", job: str, world: int, rank: int, timeout: float = 120.0) -> "Comm":
        """One process per rank on this host, messages staged through shared memory (any number of ranks per GPU).  ``ctx`` None:
        the collectives move host memory (``selftest`` on a machine without a GPU)."""
        lib = ctx.lib if ctx is not None else load()
        h = C.c_void_p(0)
        rc = lib.wfx_comm_create_shm(ctx.h if ctx is not None else None, job.encode(), world, rank, float(timeout), C.byref(h))
        if rc != 0:
            if ctx is not None:
                ctx._check(rc)
            raise _global_error(lib, rc)
        c = cls(h.value, lib)
        c.is_shm = True
        return c

    def selftest(self, ctx: "Context | None", rounds: int = 8, seed: int = 1):
        """Randomised collectives with known answers (wfx_comm_selftest)."""
        rc = self.lib.wfx_comm_selftest(self.h, ctx.h if ctx is not None else None, int(rounds), int(seed))
        if rc != 0:
            if ctx is not None:
                ctx._check(rc)
            raise _global_error(self.lib, rc)

    def barrier(self, ctx: "Context"):
        """Every rank has arrived and this rank's stream is idle."""
        if ctx is None:
            rc = self.lib.wfx_comm_barrier(self.h, None)
            if rc != 0:
                raise _global_error(self.lib, rc)
            return
        ctx._check(self.lib.wfx_comm_barrier(self.h, ctx.h))

    def allgather(self, ctx: "Context", values: np.ndarray) -> np.ndarray:
        """Small host array of every rank -> [world, ...] on every rank."""
        a = np.ascontiguousarray(values)
        out = np.empty((self.world,) + a.shape, dtype=a.dtype)
        ctx._check(self.lib.wfx_comm_allgather_host(self.h, ctx.h, _ptr(a), _ptr(out), a.nbytes))
        return out

    def close(self):
        if getattr(self, "h", None):
            self.lib.wfx_comm_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Shard:
    """One rank's part of a sharded exact decode (wfx_shard_*)."""

    def __init__(self, ctx: "Context", comm: Comm, params: DecodeParams):
        self.ctx, self.comm, self.params = ctx, comm, params
        h = C.c_void_p(0)
        ctx._check(ctx.lib.wfx_shard_create(ctx.h, comm.h, C.byref(params), C.byref(h)))
        self.h = h.value
        self.layout = shard_layout(params, comm.world, comm.rank)
        self._keep = None

    def upload(self, frames: np.ndarray):
        """``frames``: this rank's input frames [layout.in_lo, layout.in_hi)."""
        a = np.ascontiguousarray(frames)
        if a.shape[0] != self.layout.in_frames:
            raise ValueError(f"rank {self.comm.rank} needs {self.layout.in_frames} frames ({self.layout.nseg} segment(s) from frame {self.layout.in_lo}), got {a.shape[0]}")
        self.ctx._check(self.ctx.lib.wfx_shard_upload(self.h, _ptr(a)))

    def attach(self, dev_ptr: int):
        self.ctx._check(self.ctx.lib.wfx_shard_attach(self.h, C.c_void_p(dev_ptr)))

    @property
    def phases(self) -> int:
        return int(self.ctx.lib.wfx_shard_phase_count(self.h))

    def phase(self, k: int):
        self.ctx._check(self.ctx.lib.wfx_shard_phase(self.h, k))

    def run(self):
        """Enqueue every phase (RCCL communicator or world 1); asynchronous."""
        self.ctx._check(self.ctx.lib.wfx_decode_sharded(self.h))

    def result(self) -> DecodeInfo:
        info = DecodeInfo()
        self.ctx._check(self.ctx.lib.wfx_shard_result(self.h, C.byref(info)))
        return info

    def fetch(self, buffer_id: int, shape, dtype) -> np.ndarray:
        out = np.empty(shape, dtype=dtype)
        self.ctx._check(self.ctx.lib.wfx_shard_fetch(self.h, buffer_id, _ptr(out), out.nbytes))
        return out

    def close(self):
        if getattr(self, "h", None):
            self.ctx.lib.wfx_shard_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class _Pinned:
    def __init__(self, lib, ptr):
        self.lib, self.ptr = lib, ptr

    def __del__(self):
        try:
            self.lib.wfx_host_free(self.ptr)
        except Exception:
            pass


def pinned_empty(shape, dtype) -> np.ndarray:
    """NumPy array in page-locked host memory (wfx_host_alloc): uploads from it and fetches into it run at DMA speed."""
    lib = load()
    dt = np.dtype(dtype)
    n = int(np.prod(shape)) * dt.itemsize
    ptr = lib.wfx_host_alloc(max(n, 1))
    if not ptr:
        raise _global_error(lib, -3)
    buf = (C.c_char * max(n, 1)).from_address(ptr)
    buf._owner = _Pinned(lib, ptr)                 # freed when the array (and every view of it) is gone
    return np.frombuffer(buf, dtype=dt, count=int(np.prod(shape))).reshape(shape)


def device_count() -> int:
    return int(load().wfx_device_count())
