"""Time-domain front end (wefax_amd/polyphase.py + csrc/wfx_polyphase.hip): the halo-local
counterpart of the reference's FFT resampler (wefax.py:375-394) for oversampled captures.

CPU: filter design, index bookkeeping and the orchestration with the NumPy stage backend,
checked against the oracle's FFT resampler.  GPU (-m gpu): the two stencil kernels against
their float64 models, slice invariance, and the whole sharded decode of 48 kHz and
1.536 MS/s IQ captures against the oracle."""
import os

import numpy as np
import pytest

from conftest import GOLDEN
from oracle import wefax_oracle as wo
from wefax_amd import polyphase as pp
from wefax_amd import sharded, synth
from sharded_numpy_backend import NumpyStages, decimate_model, front_end_model, rational_model, to_real

TAPS = 255


def _response(st, freqs_hz):
    """|H(f)| of a stage at its input rate (rational: of the prototype the phases are cut from)."""
    fs = float(st.fs_in)
    if st.kind == "decimate":
        h, rate = st.coef64, fs
    else:   # interleave the phase rows back into the prototype at rate fs*q; row r holds h(j - left - r/q)
        proto = np.zeros(st.taps * st.q)
        for r in range(st.q):
            for j in range(st.taps):
                k = (j + 1) * st.q - r - 1
                proto[k] = st.table64[r, j]
        h, rate = proto / st.q, fs * st.q
    w = np.exp(-2j * np.pi * np.outer(np.asarray(freqs_hz) / rate, np.arange(h.shape[0])))
    return np.abs(w @ h)


@pytest.mark.parametrize("fs", [1536000, 48000, 44100, 96000, 2400000])
def test_stage_filters_meet_their_specification(fs):
    fe = pp.FrontEnd(fs)
    assert float(fe.stages[-1].fs_out) == 11025.0
    rate = pp.Fraction(fs)
    for st in fe.stages:
        assert st.fs_in == rate
        rate = st.fs_out
        last = st is fe.stages[-1]
        passband = np.linspace(0, 5000.0 if last else 5512.5, 40)
        assert np.max(np.abs(_response(st, passband) - 1.0)) < 2e-4
        # everything that would alias into 0..5512.5 Hz at the stage's output rate is >= 85 dB down
        fo = float(st.fs_out)
        k = np.arange(1, 4)[:, None]
        alias = (k * fo + np.linspace(-5512.5, 5512.5, 25)[None, :]).ravel()
        alias = alias[alias < float(st.fs_in) * (st.q if st.kind == "rational" else 1) / 2]
        if alias.size:
            assert np.max(_response(st, alias)) < 10 ** (-85 / 20)
        if st.kind == "decimate":
            assert st.ntaps % 2 == 1 and abs(st.coef64.sum() - 1) < 1e-12
            assert np.allclose(st.coef64, st.coef64[::-1])                    # linear phase, zero delay
        else:
            assert np.allclose(st.table64.sum(axis=1), 1.0)
    with pytest.raises(ValueError):
        pp.FrontEnd(8000)


def test_chain_ranges_are_consistent_and_slice_invariant():
    fe = pp.FrontEnd(1536000)
    ch = fe.chain(-700, 1900)
    assert [st.kind for st, _, _ in ch] == ["decimate", "rational", "decimate", "decimate"]
    for (s0, o0, i0), (s1, o1, i1) in zip(ch[:-1], ch[1:]):
        assert o0 == i1                                    # a stage's output range is the next one's input range
    assert ch[-1][1] == (-700, 1900)
    rng = np.random.default_rng(5)
    n0 = 40 * 20480                                        # 40 phase periods: 5880 output samples
    x = rng.integers(-20000, 20000, size=(n0, 2)).astype(np.int16)
    n = fe.n_out(n0)
    assert n == n0 * 147 // 20480
    ia, ib = fe.input_range(0, n)
    full = front_end_model(x[np.arange(ia, ib) % n0], fe.chain(0, n))
    for lo, hi in ((0, 100), (-300, 50), (n - 64, n + 200), (1234, 2345)):
        ja, jb = fe.input_range(lo, hi)
        part = front_end_model(x[np.arange(ja, jb) % n0], fe.chain(lo, hi))
        assert np.allclose(part, full[np.arange(lo, hi) % n], rtol=0, atol=1e-9 * np.abs(full).max())


@pytest.mark.parametrize("fs,stereo", [(48000, False), (1536000, True)])
def test_front_end_model_tracks_the_fft_resampler(fs, stereo):
    """Band-limited content: the stencil chain and scipy-style FFT resampling agree to the filters' ripple."""
    seconds = 0.52 if fs > 100000 else 4                  # 11025 * seconds is an integer: same rate ratio in both resamplers
    n0 = int(fs * seconds)
    t = np.arange(n0) / fs
    sig = 6000 * np.sin(2 * np.pi * 1500 * t) + 5000 * np.sin(2 * np.pi * 2300 * t + 1) + 3000 * np.sin(2 * np.pi * 3900 * t + 2)
    sig *= np.hanning(n0)                                  # no wrap-around discontinuity
    x = np.rint(sig).astype(np.int16)
    raw = np.stack([x, x], axis=1) if stereo else x
    fe = pp.FrontEnd(fs)
    n = fe.n_out(n0)
    ia, ib = fe.input_range(0, n)
    idx = np.arange(ia, ib) % n0
    got = front_end_model(raw[idx], fe.chain(0, n))
    ref = wo.resample_fft(to_real(raw), n)
    assert np.max(np.abs(got - ref)) < 3e-4 * np.max(np.abs(ref))
    # a tone the reference's brick wall removes (7 kHz) is removed here too
    tone = np.rint(8000 * np.sin(2 * np.pi * 7000 * t) * np.hanning(n0)).astype(np.int16)
    rawt = np.stack([tone, tone], axis=1) if stereo else tone
    out = front_end_model(rawt[idx], fe.chain(0, n))
    assert np.max(np.abs(out)) < 8000 * 10 ** (-80 / 20)


def _capture(fs, noise, seed=1, lpm=240, seconds=61.0, iq=False):
    """Short capture that still holds 100 sync peaks in its first half (240 LPM: a line is 0.25 s).  Whole
    seconds: 11025 * length is then an integer and the reference's n0/num equals the nominal rate ratio (for
    other lengths the two differ by up to one output sample over the whole capture, see polyphase.py)."""
    t_line = 60.0 / lpm
    phasing = 40 if lpm == 240 else 20
    lines = int(round((seconds - 3.0) / t_line)) - phasing
    return synth.synth_capture(float(fs), noise=noise, seed=seed, lpm=lpm, phasing_lines=phasing, image_lines=lines,
                               start_tone_s=1.0, stop_tone_s=1.0, black_tail_s=1.0, iq=iq)


def test_sharded_decode_with_front_end_cpu():
    """Orchestration with the NumPy backend: world-size invariance and closeness to the oracle's FFT resampler."""
    x = _capture(48000, 0.05)
    fe = pp.FrontEnd(48000)
    one = sharded.decode_emulated(NumpyStages, x, 1, lines_per_minute=240, taps=TAPS, frontend=fe)
    two = sharded.decode_emulated(NumpyStages, x, 2, lines_per_minute=240, taps=TAPS, frontend=fe)
    assert one["sync"]["start_frame"] == two["sync"]["start_frame"]
    assert np.array_equal(one["image"], two["image"]) and np.array_equal(one["digitalized"], two["digitalized"])
    n = fe.n_out(x.shape[0])
    ref_audio = wo.resample_fft(x.astype(np.float64), n)
    assert one["audio"].shape[0] == n
    # the noise between pass_hz and 5512.5 Hz, which the reference keeps and the front end drops, is the difference
    assert np.max(np.abs(one["audio"] - ref_audio)) < 0.08 * np.max(np.abs(ref_audio))
    assert np.sqrt(np.mean((one["audio"] - ref_audio) ** 2)) < 0.02 * np.max(np.abs(ref_audio))


def _free_port():
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def _gloo_worker(rank, world, port, out_dir):
    import torch
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        x = _capture(48000, 0.05)
        fe = pp.FrontEnd(48000)
        calls = []

        def raw_loader(lo, hi):             # every rank asks for its own slice of the oversampled capture only
            calls.append((lo, hi))
            return x[np.arange(lo, hi) % x.shape[0]]

        dec = sharded.ShardedDecoder(NumpyStages(), None, fe.n_out(x.shape[0]), world, rank, 240, TAPS, frontend=fe,
                                     n_in_total=x.shape[0], in_kind=0, raw_loader=raw_loader)
        assert len(calls) == 1 and calls[0][1] - calls[0][0] < 0.6 * x.shape[0]
        res = dec.run(sharded.TorchComm(dist, torch, "cpu"))
        if rank == 0:
            img, sync, low, high = res
            np.savez(os.path.join(out_dir, "root.npz"), image=img, start=sync["start_frame"], low=low, high=high)
        else:
            assert res is None
    finally:
        dist.destroy_process_group()


def test_two_gloo_ranks_with_front_end_equal_one_rank(tmp_path):
    pytest.importorskip("torch")
    import torch.multiprocessing as mp
    mp.spawn(_gloo_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    got = np.load(os.path.join(tmp_path, "root.npz"))
    ref = sharded.decode_emulated(NumpyStages, _capture(48000, 0.05), 1, lines_per_minute=240, taps=TAPS, frontend=pp.FrontEnd(48000))
    assert int(got["start"]) == ref["sync"]["start_frame"]
    assert float(got["low"]) == ref["low"] and float(got["high"]) == ref["high"]
    assert np.array_equal(got["image"], ref["image"])


# ---------------------------------------------------------------------------------------
# GPU
# ---------------------------------------------------------------------------------------
@pytest.fixture(scope="module")
def ctx():
    from wefax_amd import _native as nat
    c = nat.Context(0)
    yield c
    c.close()


def _dev(ctx, a):
    p = ctx.dev_malloc(max(a.nbytes, 16))
    ctx.dev_upload(p, np.ascontiguousarray(a))
    return p


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["i16", "iq", "f32"])
@pytest.mark.parametrize("factor,ntaps", [(32, 239), (4, 493), (2, 17), (64, 301), (8, 5)])
def test_decimate_kernel_matches_model(ctx, kind, factor, ntaps):
    from wefax_amd import _native as nat
    rng = np.random.default_rng(factor * 1000 + ntaps)
    n_out = 5000 + factor
    first = 37
    n_in = first + (n_out - 1) * factor + ntaps + 11
    coef = (rng.standard_normal(ntaps) / ntaps).astype(np.float32)
    if kind == "f32":
        raw = (rng.standard_normal(n_in) * 1000).astype(np.float32)
        real, k = raw.astype(np.float64), nat.WFX_IN_F32_MONO
    elif kind == "iq":
        raw = rng.integers(-32768, 32767, size=(n_in, 2)).astype(np.int16)      # sums overflow int16: the wrap is part of the contract
        real, k = to_real(raw), nat.WFX_IN_I16_STEREO
    else:
        raw = rng.integers(-32768, 32767, size=n_in).astype(np.int16)
        real, k = raw.astype(np.float64), nat.WFX_IN_I16_MONO
    p_in = _dev(ctx, raw)
    for f64, dt in ((False, np.float32), (True, np.float64)):
        for off in (0, 1, 3):                # misaligned base pointers and an input that ends inside the window
            esz = raw.nbytes // n_in
            p_out = ctx.dev_malloc(n_out * 8)
            ctx.d_decimate_fir(p_in + off * esz, k, n_in - off - 5, first - off, factor, coef, p_out, f64, n_out)
            got = ctx.dev_download(p_out, (n_out,), dt).astype(np.float64)
            want = decimate_model(real[off:n_in - 5], first - off, factor, coef, n_out)
            scale = np.sum(np.abs(coef)) * np.max(np.abs(real))
            assert np.max(np.abs(got - want)) < 3e-6 * scale
            ctx.dev_free(p_out)
    ctx.dev_free(p_in)


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["i16", "iq", "f32"])
@pytest.mark.parametrize("p,q,taps", [(160, 147, 12), (250, 147, 16), (500, 441, 12), (147, 160, 8)])
def test_rational_kernel_matches_model(ctx, kind, p, q, taps):
    from wefax_amd import _native as nat
    rng = np.random.default_rng(p + q)
    n_out = 70001
    m0 = 12345678901 % (q * 1000) + 5 * q
    left = taps // 2 - 1
    pos0 = (m0 * p) // q
    n_in = ((m0 + n_out) * p) // q - pos0 + taps
    base0 = pos0 - 3                                       # the first outputs reach 3 - left samples before the buffer: zeros
    table = (rng.standard_normal((q, taps)) / taps).astype(np.float32)
    if kind == "f32":
        raw = (rng.standard_normal(n_in) * 1000).astype(np.float32)
        real, k = raw.astype(np.float64), nat.WFX_IN_F32_MONO
    elif kind == "iq":
        raw = rng.integers(-32768, 32767, size=(n_in, 2)).astype(np.int16)
        real, k = to_real(raw), nat.WFX_IN_I16_STEREO
    else:
        raw = rng.integers(-32768, 32767, size=n_in).astype(np.int16)
        real, k = raw.astype(np.float64), nat.WFX_IN_I16_MONO
    p_in, p_out = _dev(ctx, raw), ctx.dev_malloc(n_out * 4)
    ctx.d_resample_rational(p_in, k, n_in - 7, base0 + left, p, q, table, m0, p_out, n_out)
    got = ctx.dev_download(p_out, (n_out,), np.float32).astype(np.float64)
    want = rational_model(real[:n_in - 7], base0 + left, p, q, table, m0, n_out)
    assert np.max(np.abs(got - want)) < 3e-6 * np.max(np.sum(np.abs(table), axis=1)) * np.max(np.abs(real))
    ctx.dev_free(p_in)
    ctx.dev_free(p_out)


def _oracle(x, fs, lpm):
    import tempfile
    with tempfile.TemporaryDirectory() as td:
        path = os.path.join(td, "x.wav")
        synth.write_wav(path, int(fs), x)
        return wo.process(path, lpm, want_messages=False)


def _image_stats(img, ref):
    d = np.abs(img.astype(np.int16) - ref.astype(np.int16))
    return int(d.max()), float(np.mean(d <= 1)), float(d.mean())


@pytest.mark.gpu
@pytest.mark.parametrize("fs,iq,noise", [(48000, False, 0.05), (192000, True, 0.05), (192000, True, 0.02)])
def test_sharded_decode_with_front_end_gpu(ctx, fs, iq, noise):
    """Whole halo-local decode of an oversampled capture: bit-identical for 1 and 2 ranks; against the oracle's
    exact path (FFT resample + FFT Hilbert) the 11 025 Hz audio agrees to what the filters' transition band
    drops and the image to a few grey levels (measured and printed)."""
    x = _capture(fs, noise, iq=iq)
    fe = pp.FrontEnd(fs)
    mk = lambda: sharded.HipStages(ctx)
    one = sharded.decode_emulated(mk, x, 1, lines_per_minute=240, taps=4095, frontend=fe)
    two = sharded.decode_emulated(mk, x, 2, lines_per_minute=240, taps=4095, frontend=fe)
    assert two["sync"]["start_frame"] == one["sync"]["start_frame"]
    assert np.array_equal(two["audio"], one["audio"])
    assert np.array_equal(two["digitalized"], one["digitalized"])
    assert np.array_equal(two["image"], one["image"])
    n = fe.n_out(x.shape[0])
    ref_audio = wo.resample_fft(to_real(x), n)
    rel = np.max(np.abs(one["audio"] - ref_audio)) / np.max(np.abs(ref_audio))
    ref = _oracle(x, fs, 240)
    smx, sw1, smean = _image_stats(one["digitalized"], ref["digitalized"])
    print(f"fs={fs} iq={iq} noise={noise}: audio rel err {rel:.2e}; uint8 stream max|d|={smx} within1={sw1:.4f} mean|d|={smean:.3f}")
    assert rel < 0.08
    assert smean < (2.5 if fs > 48000 else 4.5) and smx <= 40
    if noise == 0.05:       # (which phasing group closes is a knife-edge decision on cleaner captures)
        assert ref["start_frame"] == one["sync"]["start_frame"]
    if ref.get("start_frame") == one["sync"]["start_frame"]:
        mx, w1, mean = _image_stats(one["image"], ref["image"])
        print(f"    image max|d|={mx} within1={w1:.4f} mean|d|={mean:.3f}")
        assert mean < (2.5 if fs > 48000 else 4.5) and mx <= 40


@pytest.mark.gpu
@pytest.mark.parametrize("two_x", [False, True])
def test_front_end_plus_exact_rest_gpu(ctx, two_x):
    """One GPU: time-domain front end + the exact rest of the path (FrontEndExactDecoder).  With the front end stopping
    at 22 050 Hz the exact FFT resampler applies the reference's own brick wall and only wide, flat filters remain."""
    fs = 1536000
    x = _capture(fs, 0.05, seed=0, lpm=120, seconds=40.0, iq=True)
    fe = pp.FrontEnd(fs, stop_at_2x=two_x)
    dec = sharded.FrontEndExactDecoder(ctx, fe, x, lines_per_minute=120)
    dec.run()
    info = dec.result()
    ref = _oracle(x, fs, 120)
    assert ref["start_frame"] == info.start_frame and dec.n == 441000
    img = dec.fetch("image")
    mx, w1, mean = _image_stats(img, ref["image"])
    smx, sw1, smean = _image_stats(dec.fetch("digitalized"), ref["digitalized"])
    print(f"front end (stop_at_2x={two_x}) + exact rest, 1.536 MS/s IQ 40 s: image max|d|={mx} within1={w1:.4f} mean|d|={mean:.3f}; "
          f"stream max|d|={smx} mean|d|={smean:.3f}")
    assert img.shape == ref["image"].shape and mean < (0.2 if two_x else 1.0) and w1 > (0.99 if two_x else 0.9)
    dec.close()


@pytest.mark.gpu
def test_iq_stream_1536k_against_oracle_gpu(ctx):
    """BASELINE configs[3] at a length the oracle finishes in seconds: 20 s of 1.536 MS/s int16 IQ through the
    time-domain front end + halo-local path on one rank, against the oracle's reference-faithful decode."""
    fs = 1536000
    x = _capture(fs, 0.05, seed=0, lpm=120, seconds=40.0, iq=True)
    fe = pp.FrontEnd(fs)
    got = sharded.decode_emulated(lambda: sharded.HipStages(ctx), x, 1, lines_per_minute=120, taps=4095, frontend=fe)
    ref = _oracle(x, fs, 120)
    assert ref["start_frame"] == got["sync"]["start_frame"]
    n = fe.n_out(x.shape[0])
    rel = np.max(np.abs(got["audio"] - ref["audio_resampled"])) / np.max(np.abs(ref["audio_resampled"])) if "audio_resampled" in ref else None
    mx, w1, mean = _image_stats(got["image"], ref["image"])
    print(f"1.536 MS/s IQ 40 s: image {got['image'].shape} max|d|={mx} within1={w1:.4f} mean|d|={mean:.3f} audio {rel}")
    assert got["image"].shape == ref["image"].shape and n == 441000
    assert mean < 1.5 and w1 > 0.8 and mx <= 40
