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
