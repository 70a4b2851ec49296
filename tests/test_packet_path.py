"""Live path, one audio packet (data_packet.py:408-464): the oracle against goldens produced by the
reference's own DataPacket (tests/golden/make_packet_golden.py), and -- on a GPU -- the HIP path against both."""
import os

import numpy as np
import pytest

from conftest import GOLDEN
from oracle import wefax_oracle as wo


def _cases():
    z = np.load(os.path.join(GOLDEN, "packets.npz"))
    return [(str(n), int(z[f"{n}__sr"]), z[f"{n}__in"], z[f"{n}__out"]) for n in z["names"]]


CASES = _cases()


@pytest.mark.parametrize("name,sr,x,want", CASES, ids=[c[0] for c in CASES])
def test_oracle_packet_matches_the_reference(name, sr, x, want):
    got = wo.process_packet(x, sr)["samples"]
    assert got.dtype == np.uint8 and np.array_equal(got, want)


@pytest.fixture(scope="module")
def ctx():
    from wefax_amd import _native as nat
    c = nat.Context(0)
    yield c
    c.close()


@pytest.mark.gpu
@pytest.mark.parametrize("name,sr,x,want", CASES, ids=[c[0] for c in CASES])
def test_hip_packet_matches_the_reference(ctx, name, sr, x, want):
    from wefax_amd.packet import DataPacket
    pkt = DataPacket(sr, x, 120, "/tmp/", 1, 0, ctx=ctx)
    assert pkt.samples.dtype == np.int64 and pkt.samples.shape == want.shape        # the reference's astype(int)
    assert np.array_equal(pkt.samples, want.astype(np.int64))
    ref = wo.process_packet(x, sr)
    assert abs(pkt.low - ref["low"]) <= 1e-9 * abs(ref["high"]) and abs(pkt.high - ref["high"]) <= 1e-9 * abs(ref["high"])


@pytest.mark.gpu
def test_hip_packet_batch_and_float_input(ctx):
    """A run of packets through one context, and float64 input (what a caller passing a float array gets)."""
    from wefax_amd.packet import DataPacket, process_packets
    rng = np.random.default_rng(9)
    t = np.arange(11025 * 6)
    x = (7000 * np.sin(2 * np.pi * 1900 / 11025 * t + 2 * np.sin(t / 500.0)) + 400 * rng.standard_normal(t.shape[0])).astype(np.int16)
    outs = process_packets(ctx, 11025, x.reshape(6, 11025))
    for k in range(6):
        assert np.array_equal(outs[k], wo.process_packet(x[k * 11025:(k + 1) * 11025], 11025)["samples"])
    xf = x[:11025].astype(np.float64) * 0.37
    assert np.array_equal(DataPacket(11025, xf, 120, "/tmp/", 1, 0, ctx=ctx).samples, wo.process_packet(xf, 11025)["samples"])


@pytest.mark.gpu
def test_hip_packets_through_one_context(ctx):
    from wefax_amd.packet import process_packets
    z = np.load(os.path.join(GOLDEN, "packets.npz"))
    names = [str(n) for n in z["names"] if str(n).startswith("mono_noisy_120")]
    outs = process_packets(ctx, 11025, [z[f"{n}__in"] for n in names])
    for n, got in zip(names, outs):
        assert got.dtype == np.uint8 and np.array_equal(got, z[f"{n}__out"])


@pytest.mark.gpu
@pytest.mark.parametrize("sr", [8000, 16000, 22050, 44100, 48000, 96000, 192000])
@pytest.mark.parametrize("dtype", [np.int16, np.float64])
def test_notch_at_the_sound_cards_rate(ctx, sr, dtype):
    """data_packet.py:430-432 designs the notch at the packet's rate: pole radius 0.2 (11 025 Hz) .. 0.96 (192 kHz).
    The chunked recurrence has to reproduce scipy's sequential filtfilt; tolerance: 1e-12 of the signal's scale
    (rounding-level; the uint8 stream is compared bit for bit in the golden cases above)."""
    from wefax_amd import hostparams as hp
    rng = np.random.default_rng(sr)
    t = np.arange(sr + 37) / sr
    x = 9000 * np.sin(2 * np.pi * 1900 * t) + 4000 * np.sin(2 * np.pi * 2600 * t) + rng.normal(0, 2000, t.size)
    x = np.clip(np.rint(x), -32768, 32767).astype(dtype)
    b, a = hp.iirnotch(2600, 1, sr)
    want = wo.filtfilt_biquad(b, a, x)
    got = ctx.notch_filtfilt(x, b, a)
    assert got.shape == want.shape
    assert np.abs(got - want).max() <= 1e-12 * np.abs(want).max()
    # packet end to end at this rate against the oracle
    from wefax_amd.packet import DataPacket
    pkt = DataPacket(sr, x, 120, "/tmp/", 1, 0, ctx=ctx)
    o = wo.process_packet(x, sr)
    assert np.array_equal(pkt.samples, o["samples"].astype(np.int64))
    assert abs(pkt.low - o["low"]) <= 1e-9 * o["high"] and abs(pkt.high - o["high"]) <= 1e-9 * o["high"]


@pytest.mark.gpu
def test_packet_rejects_what_the_reference_rejects(ctx):
    from wefax_amd.packet import DataPacket
    with pytest.raises(ValueError, match="padlen"):
        DataPacket(11025, np.zeros(9, dtype=np.int16), 120, "/tmp/", 1, 0, ctx=ctx)


def test_packet_module_needs_the_library(monkeypatch):
    """No CPU fallback: without a GPU the context constructor raises."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from wefax_amd.packet import DataPacket
    with pytest.raises(Exception):
        DataPacket(11025, np.zeros(11025, dtype=np.int16), 120, "/tmp/", 1, 0)


# ---- detectors of the live path (data_packet.py:301-406) ---------------------------------------------
def _det_cases():
    z = np.load(os.path.join(GOLDEN, "packets.npz"))
    out = []
    for n in z["names"]:
        n = str(n)
        out.append((n, int(z[f"{n}__sr"]), z[f"{n}__in"], z[f"{n}__out"], z[f"{n}__amp"], bool(z[f"{n}__start_tone"]), bool(z[f"{n}__stop_tone"]),
                    z[f"{n}__sp_flags"].tolist(), z[f"{n}__sp_fft_freq"], z[f"{n}__sp_fft_height"], z[f"{n}__sp_samples"].tolist()))
    return out


DET = _det_cases()


@pytest.mark.parametrize("case", DET, ids=[c[0] for c in DET])
def test_oracle_detectors_match_the_reference(case):
    name, sr, raw, dig, amp, start, stop, flags, pf, ph, ps = case
    f, a = wo.packet_spectrum(raw, sr)
    assert np.array_equal(a, amp)                                  # same numpy calls
    assert wo.contain_start_tone(raw, sr) == start
    assert wo.contain_stop_tone(raw, sr) == stop
    sp = wo.packet_find_sync_pulse(raw, dig, sr)
    assert [sp["frequency_peak_found"], sp["samples_peak_found"], sp["pulse_found"]] == flags
    assert np.array_equal(sp["peaks_fft"][0], pf) and np.array_equal(sp["peaks_fft"][1], ph)
    assert sp["peaks_samples"] == ps


@pytest.mark.parametrize("case", DET, ids=[c[0] for c in DET])
def test_host_detector_logic_matches_the_reference(case):
    """wefax_amd/detect.py (the product's host side) on the reference's own spectra and digitised samples."""
    from wefax_amd import detect
    name, sr, raw, dig, amp, start, stop, flags, pf, ph, ps = case
    f = detect.frequencies(len(raw), sr)
    assert detect.contain_tone(f, amp, detect.TONES["start_distance"]) == start
    assert detect.contain_tone(f, amp, detect.TONES["stop_distance"]) == stop
    sp = detect.find_sync_pulse(f, amp, dig, sr)
    assert [sp["frequency_peak_found"], sp["samples_peak_found"], sp["pulse_found"]] == flags
    assert np.array_equal(sp["peaks_fft"][0], pf) and np.array_equal(sp["peaks_fft"][1], ph)
    assert sp["peaks_samples"] == ps


@pytest.mark.parametrize("seed", range(6))
def test_host_find_peaks_and_pattern_search_against_the_oracle(seed):
    from wefax_amd import detect
    rng = np.random.default_rng(seed)
    x = rng.random(4000)
    x[rng.integers(0, 4000, 300)] = 0.5                         # plateaus and ties
    x = np.round(x, 2 if seed % 2 else 6)
    for kw in (dict(height=0.3), dict(height=0.05, distance=25), dict(height=0.2, prominence=0.3), dict(height=0.05, distance=7.5, prominence=0.1)):
        got, want = detect.find_peaks(x, **kw), wo.find_peaks(x, **kw)
        if "distance" in kw and len(np.unique(x[wo._local_maxima(x)])) < len(wo._local_maxima(x)):
            continue                                            # equal heights: scipy's own order is unspecified (argsort)
        assert np.array_equal(got[0], want[0]) and np.array_equal(got[1], want[1]), kw
    s = rng.integers(0, 256, size=11025 + 100 * seed).astype(np.uint8)
    s[rng.integers(0, 9000)::2756][:3] = 255
    assert detect.pattern_search(s, 11025) == wo.packet_pattern_search(s, 11025)


@pytest.mark.gpu
@pytest.mark.parametrize("case", DET, ids=[c[0] for c in DET])
def test_hip_packet_detectors_match_the_reference(ctx, case):
    from wefax_amd import detect
    from wefax_amd.packet import DataPacket
    name, sr, raw, dig, amp, start, stop, flags, pf, ph, ps = case
    pkt = DataPacket(sr, raw, 120, "/tmp/", 1, 0, ctx=ctx)
    f, a = pkt._spectrum()
    assert np.abs(a - amp).max() <= 1e-12                       # amplitudes are normalised to <= 1
    assert pkt.contain_start_tone() == start and pkt.contain_stop_tone() == stop
    sp = pkt.find_sync_pulse()
    assert [sp["frequency_peak_found"], sp["samples_peak_found"], sp["pulse_found"]] == flags
    assert np.array_equal(sp["peaks_fft"][0], pf) and np.abs(sp["peaks_fft"][1] - ph).max(initial=0.0) <= 1e-12
    assert sp["peaks_samples"] == ps


@pytest.mark.gpu
@pytest.mark.parametrize("rate,lpm,frames", [(11025, 120, 10), (11025, 240, 10), (8000, 120, 3)])
def test_live_strip_equals_pillow(ctx, rate, lpm, frames):
    """wefax_live.py:124-148 rendered with Pillow exactly as the reference does (Image.new / putpixel / resize)."""
    from PIL import Image
    from wefax_amd.packet import frames_to_image
    t_frame = 1 / (lpm / 60)
    w = int(t_frame * rate)
    rng = np.random.default_rng(rate + lpm)
    pts = rng.integers(0, 256, size=w * frames + 777)
    img = Image.new("L", (w, frames))
    px = py = 0
    for p in range(w * frames):
        img.putpixel((px, py), 255 - int(pts[p]))
        px += 1
        if px >= w:
            px = 0
            py += 1
    want = np.asarray(img.resize((w, 4 * frames)))
    got = frames_to_image(ctx, pts, rate, t_frame, frames)
    assert got.shape == want.shape and np.array_equal(got, want)


@pytest.mark.gpu
def test_hip_packets_back_to_back_equal_one_by_one(ctx):
    """wfx_packets_process: the same decode per packet, one upload / download for the whole stack."""
    from wefax_amd import synth
    from wefax_amd.packet import _process, process_packets
    x = synth.config_c2(noise=0.02, seed=2)[:12 * 11025].reshape(12, 11025)
    many = process_packets(ctx, 11025, x)
    for p, got in zip(x, many):
        assert np.array_equal(got, _process(ctx, 11025, p)[0])
        assert np.array_equal(got, wo.process_packet(p, 11025)["samples"])
    ragged = process_packets(ctx, 11025, [x[0], x[1][:9000]])            # different lengths: decoded one by one
    assert np.array_equal(ragged[0], many[0]) and ragged[1].shape == (9000,)


@pytest.mark.gpu
def test_packet_entry_points_reject_bad_input(ctx):
    from wefax_amd import _native as nat
    with pytest.raises(nat.NativeError):
        ctx.packet_spectrum(np.zeros(1, dtype=np.int16))                 # fewer than two samples
    amp = ctx.packet_spectrum(np.array([3, -3, 3, -3, 3], dtype=np.int16))      # odd length: n // 2 bins
    f, a = wo.packet_spectrum(np.array([3, -3, 3, -3, 3], dtype=np.int16), 5)
    assert amp.shape == (2,) and np.abs(amp / (amp.max() + 0.0001) - a).max() < 1e-12
    with pytest.raises(ValueError):
        ctx.packets_process(np.zeros(11025, dtype=np.int16), [1, 0, 0], [1, 0, 0], (0, 0, 1, 1), 0.0, 0.0)   # not [count, n]
