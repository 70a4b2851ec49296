"""Time-domain front end (wefax_amd/polyphase.py + csrc/wfx_polyphase.hip): the halo-local
counterpart of the reference's FFT resampler (wefax.py:375-394) for oversampled captures.

CPU: filter design and index bookkeeping, the float64 model of the chain checked against the
oracle's FFT resampler.  GPU (-m gpu): the stencil kernel against its integer / float64 models,
and whole decodes of 44.1 kHz / 48 kHz / 192 kHz / 1.536 MS/s IQ captures -- front end to the hand-over
rate, then the exact path (one GPU fused, and sharded over emulated ranks) -- against the oracle."""
import os

import numpy as np
import pytest

from conftest import GOLDEN
from oracle import wefax_oracle as wo
from wefax_amd import polyphase as pp
from wefax_amd import sharded, synth
from polyphase_models import decimate_model, front_end_model, to_real


def _response(st, freqs_hz):
    """|H(f)| of a stage at its input rate."""
    fs = float(st.fs_in)
    h = st.coef64
    w = np.exp(-2j * np.pi * np.outer(np.asarray(freqs_hz) / fs, np.arange(h.shape[0])))
    return np.abs(w @ h)


@pytest.mark.parametrize("fs", [1536000, 192000, 48000, 44100, 96000, 2400000, 176400])
def test_stage_filters_meet_their_specification(fs):
    """Every chain is decimations only, ends at a rate above 11 025 Hz that divides the capture's rate, and each stage passes
    0..5512.5 Hz flat and rejects whatever would alias into that band at its output rate."""
    fe = pp.FrontEnd(fs)
    assert fe.out_rate == pp.FrontEnd.handover_rate(fs) and 14000 <= fe.out_rate < 24000 and fs % fe.out_rate == 0
    assert float(fe.stages[-1].fs_out) == fe.out_rate and fe.exact_tail and fe.f64
    rate = pp.Fraction(fs)
    prod = np.ones(40)
    passband = np.linspace(0, 5512.5, 40)
    for st in fe.stages:
        assert st.kind == "decimate" and st.fs_in == rate
        rate = st.fs_out
        prod = prod * _response(st, passband)
        # everything that would alias into 0..5512.5 Hz at the stage's output rate is >= 125 dB down
        fo = float(st.fs_out)
        k = np.arange(1, 4)[:, None]
        alias = (k * fo + np.linspace(-5512.5, 5512.5, 25)[None, :]).ravel()
        alias = alias[alias < float(st.fs_in) / 2]
        if alias.size:
            assert np.max(_response(st, alias)) < 10 ** (-125 / 20)
        assert st.ntaps % 2 == 1 and abs(st.coef64.sum() - 1) < (1e-8 if fe.design is not None else 1e-12)    # (a pair: unit gain together)
        assert np.allclose(st.coef64, st.coef64[::-1])                    # linear phase, zero delay
    assert np.max(np.abs(prod - 1.0)) < 4e-7                              # the CHAIN is flat (a least-squares pair: only together)
    with pytest.raises(ValueError):
        pp.FrontEnd(8000)
    with pytest.raises(ValueError):
        pp.FrontEnd(22050)


def test_chain_ranges_are_consistent_and_slice_invariant():
    assert [st.factor for st in pp.FrontEnd(1536000, stop_rate=16000).stages] == [32, 3]
    assert [st.factor for st in pp.FrontEnd(1536000, stop_rate=48000).stages] == [32]
    assert [st.factor for st in pp.FrontEnd(192000).stages] == [4, 3] and [st.factor for st in pp.FrontEnd(44100).stages] == [3]
    assert pp.FrontEnd.handover_rate(1536000) == 16000 and pp.FrontEnd.handover_rate(44100) == 14700 and pp.FrontEnd.handover_rate(96000) == 16000
    assert pp.stage_factors(96) == [32, 3] and pp.stage_factors(128) == [64, 2] and pp.stage_factors(7) is None and pp.stage_factors(1) is None
    fe16 = pp.FrontEnd(1536000, stop_rate=16000)
    # the ingest pair is the least-squares multiband design: 8 taps per polyphase row, stop bands and pair flatness as asked for;
    # its first filter sits on the integer-exact kernel's 2**-30 grid
    assert fe16.stages[0].ntaps == 8 * 32 - 3 and fe16.design is not None and fe16.stages[0].fix_shift == 30 and fe16.stages[1].fix_shift == 0
    assert fe16.design["stage1_stop_db"] < -129 and fe16.design["stage2_stop_db"] < -135 and fe16.design["pair_flatness"] < 1.8e-7
    assert pp.FrontEnd(192000, stop_rate=16000).design is None          # small factors keep the Kaiser designs
    for bad in (13000, 22050, 11025, 14700):                            # not divisors of 1 536 000 (or not above 11 025 Hz)
        with pytest.raises(ValueError):
            pp.FrontEnd(1536000, stop_rate=bad)
    fe = fe16
    ch = fe.chain(-700, 1900)
    for (s0, o0, i0), (s1, o1, i1) in zip(ch[:-1], ch[1:]):
        assert o0 == i1                                    # a stage's output range is the next one's input range
    assert ch[-1][1] == (-700, 1900)
    rng = np.random.default_rng(5)
    n0 = 96 * 6000                                         # 6000 hand-over samples
    x = rng.integers(-20000, 20000, size=(n0, 2)).astype(np.int16)
    n = fe.n_out(n0)
    assert n == n0 // 96
    ia, ib = fe.input_range(0, n)
    full = front_end_model(x[np.arange(ia, ib) % n0], fe.chain(0, n))
    for lo, hi in ((0, 100), (-300, 50), (n - 64, n + 200), (1234, 2345)):
        ja, jb = fe.input_range(lo, hi)
        part = front_end_model(x[np.arange(ja, jb) % n0], fe.chain(lo, hi))
        assert np.allclose(part, full[np.arange(lo, hi) % n], rtol=0, atol=1e-9 * np.abs(full).max())


@pytest.mark.parametrize("fs,stereo", [(48000, False), (1536000, True), (44100, False)])
def test_front_end_model_tracks_the_fft_resampler(fs, stereo):
    """Band-limited content: the stencil chain to the hand-over rate followed by the FFT resampler agrees with FFT resampling of
    the raw capture (the reference, wefax.py:384) to the filters' ripple."""
    seconds = 0.52 if fs > 100000 else 4                  # 11025 * seconds is an integer
    n0 = int(fs * seconds)
    t = np.arange(n0) / fs
    sig = 6000 * np.sin(2 * np.pi * 1500 * t) + 5000 * np.sin(2 * np.pi * 2300 * t + 1) + 3000 * np.sin(2 * np.pi * 3900 * t + 2)
    sig *= np.hanning(n0)                                  # no wrap-around discontinuity
    x = np.rint(sig).astype(np.int16)
    raw = np.stack([x, x], axis=1) if stereo else x
    fe = pp.FrontEnd(fs)
    n_fe, n = fe.n_out(n0), fe.n_target(n0)
    ia, ib = fe.input_range(0, n_fe)
    idx = np.arange(ia, ib) % n0
    got = wo.resample_fft(front_end_model(raw[idx], fe.chain(0, n_fe)), n)
    ref = wo.resample_fft(to_real(raw), n)
    assert np.max(np.abs(got - ref)) < 2e-6 * np.max(np.abs(ref))
    # a tone the hand-over rate cannot carry (above out_rate - 5512.5 Hz: it would alias into the band) is removed by the chain
    f_alias = fe.out_rate - 3000.0
    tone = 8000 * np.sin(2 * np.pi * f_alias * t) * np.hanning(n0)          # (unrounded: int16 rounding noise is broadband)
    out = front_end_model(tone[idx], fe.chain(0, n_fe))
    assert np.max(np.abs(out)) < 8000 * 10 ** (-120 / 20)


def test_hand_over_length_keeps_the_reference_grid_for_captures_that_are_not_whole_seconds():
    """wefax.py:384: num = int(11025 * n0 / fs) and scipy's resample puts output j at input position j * n0 / num.  The chain
    to the hand-over rate tiles the capture's period exactly (n0 * out_rate / fs_in samples) or refuses the length."""
    fe = pp.FrontEnd(1536000, stop_rate=16000)
    assert fe.granule() == 96 and pp.FrontEnd(48000, stop_rate=16000).granule() == 3 and pp.FrontEnd(44100).granule() == 3
    n0 = 1536000 * 7 + 96 * 1234 + 96                      # 7.0772 s
    assert fe.n_out(n0) == n0 // 96
    assert fe.n_target(n0) == int(11025 * (n0 / 1536000)) == 78025
    for bad in (n0 + 1, n0 + 95, n0 - 31):
        with pytest.raises(ValueError, match="granule 96"):
            fe.n_out(bad)


def test_front_end_model_then_fft_resample_tracks_the_reference_off_whole_seconds():
    """A capture of 2.26 s at 48 kHz (a multiple of 3 frames, not of 640): chain to 16 kHz, then the oracle's FFT resampler to
    int(11025 * n0 / fs) samples -- against the oracle's FFT resampler on the raw capture.  The nominal-ratio form (round 2) is off
    by up to one output sample at the end; this one is not."""
    fs = 48000
    n0 = 108480 + 3 * 7                                    # 108501 = 3 * 36167; 11025 * n0 / fs = 24921.32...
    t = np.arange(n0) / fs
    sig = 6000 * np.sin(2 * np.pi * 1500 * t) + 5000 * np.sin(2 * np.pi * 2300 * t + 1) + 3000 * np.sin(2 * np.pi * 3900 * t + 2)
    sig *= np.hanning(n0)
    x = np.rint(sig).astype(np.int16)
    fe = pp.FrontEnd(fs, stop_rate=16000)
    n_fe, n = fe.n_out(n0), fe.n_target(n0)
    assert n_fe == 36167 and n == 24921 and (11025 * n0) % fs != 0
    ia, ib = fe.input_range(0, n_fe)
    mid = front_end_model(x[np.arange(ia, ib) % n0], fe.chain(0, n_fe))
    got = wo.resample_fft(mid, n)
    ref = wo.resample_fft(x.astype(np.float64), n)
    assert np.max(np.abs(got - ref)) < 3e-4 * np.max(np.abs(ref))
    # what the nominal ratio would have given: the same samples on a grid stretched by n0 / (n * fs / 11025)
    stretched = wo.resample_fft(x.astype(np.float64), n + 1)[:n]
    assert np.max(np.abs(stretched - ref)) > 100 * np.max(np.abs(got - ref))


def _capture(fs, noise, seed=1, lpm=240, seconds=61.0, iq=False):
    """Short capture that still holds 100 sync peaks in its first half (240 LPM: a line is 0.25 s).  Whole
    seconds: 11025 * length is then an integer and the reference's n0/num equals the nominal rate ratio (for
    other lengths the two differ by up to one output sample over the whole capture, see polyphase.py)."""
    t_line = 60.0 / lpm
    phasing = 40 if lpm == 240 else 20
    lines = int(round((seconds - 3.0) / t_line)) - phasing
    return synth.synth_capture(float(fs), noise=noise, seed=seed, lpm=lpm, phasing_lines=phasing, image_lines=lines,
                               start_tone_s=1.0, stop_tone_s=1.0, black_tail_s=1.0, iq=iq)


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
@pytest.mark.parametrize("kind", ["i16", "iq", "f64"])
@pytest.mark.parametrize("factor,ntaps,worst,legacy", [(32, 253, False, False), (32, 253, True, False), (32, 253, True, True), (4, 47, False, False),
                                                       (8, 61, False, False), (8, 61, True, False), (3, 95, False, False), (2, 17, False, False),
                                                       (64, 301, False, False), (64, 301, True, False), (16, 125, True, False), (16, 125, True, True)])
def test_decimate_fir64_is_exact_where_it_says_so(ctx, kind, factor, ntaps, worst, legacy):
    """wfx_d_decimate_fir64: int16 input with an aligned power-of-two factor is an integer dot product -- EQUAL to the integer
    model for every input, also the worst case for the accumulators (all samples -32768 / +32767 with the signs of the taps);
    anything else is float64 arithmetic in a fixed order."""
    from wefax_amd import _native as nat
    rng = np.random.default_rng(factor * 1000 + ntaps)
    n_out = 5000 + factor
    first = 37
    n_in = first + (n_out - 1) * factor + ntaps + 11
    h = np.sinc((np.arange(ntaps) - (ntaps - 1) / 2) / factor / 1.3) * np.hanning(ntaps + 2)[1:-1]
    # grid: the finest one the kernel takes for this factor (both halves of the sums flushed every few rows: up to 2**-30; the
    # worst-case inputs below then fill the int32 sums of a flush window to within 2 % of their range), or -- `legacy` -- the one
    # that keeps the high halves of all taps in one int32
    pow2 = factor & (factor - 1) == 0
    sh = pp.fix_shift_for(h / h.sum()) if legacy or not pow2 else pp.fix_shift_for(h / h.sum(), factor)
    assert 20 <= sh <= 30 and (legacy or not pow2 or sh >= pp.fix_shift_for(h / h.sum()))
    coef = pp.quantize_taps(h / h.sum(), sh)
    fix = np.rint(coef * 2.0 ** sh).astype(np.int64)
    assert np.array_equal(fix / 2.0 ** sh, coef) and coef.sum() == 1.0
    if kind == "f64":
        raw = rng.standard_normal(n_in) * 1000
        real, k, esz = raw, nat.WFX_IN_F64_MONO, 8
    elif kind == "iq":
        raw = rng.integers(-32768, 32767, size=(n_in, 2)).astype(np.int16)
        if worst:
            raw[:, 1] = 0
            raw[:, 0] = np.where(rng.integers(0, 2, n_in) > 0, 32767, -32768)
        real, k, esz = to_real(raw), nat.WFX_IN_I16_STEREO, 4
    else:
        raw = rng.integers(-32768, 32767, size=n_in).astype(np.int16)
        if worst:       # every product of output 0's window has the same sign
            raw[:] = -32768
            idx = first + np.arange(ntaps)
            raw[idx] = np.where(fix >= 0, 32767, -32768).astype(np.int16)
        real, k, esz = raw.astype(np.float64), nat.WFX_IN_I16_MONO, 2
    p_in = _dev(ctx, raw)
    per16 = 16 // esz
    expect_exact = kind != "f64" and factor >= per16 and factor & (factor - 1) == 0
    for off in (0, 1, 3):
        p_out = ctx.dev_malloc(n_out * 8)
        ex = ctx.d_decimate_fir64(p_in + off * esz, k, n_in - off - 5, first - off, factor, coef, p_out, n_out, sh)
        got = ctx.dev_download(p_out, (n_out,), np.float64)
        ctx.dev_free(p_out)
        assert ex == expect_exact
        x = real[off:n_in - 5]
        if ex:
            # integer model: sum_j fix[j] * (2 * sample) / 2**28, all in int64 (IQ halves are k/2: doubled they are integers)
            xi = np.rint(2 * x).astype(np.int64)
            lo, hi = first - off, first - off + (n_out - 1) * factor + ntaps
            xp = np.zeros(hi - lo, dtype=np.int64)
            a, b = max(lo, 0), min(hi, xi.shape[0])
            xp[a - lo:b - lo] = xi[a:b]
            acc = np.zeros(n_out, dtype=np.int64)
            for j in range(ntaps):
                acc += fix[j] * xp[j:j + (n_out - 1) * factor + 1:factor]
            want = acc.astype(np.float64) / 2.0 ** (sh + 1)
            assert np.array_equal(got, want)
        else:
            want = decimate_model(x, first - off, factor, coef, n_out)
            assert np.max(np.abs(got - want)) <= 1e-13 * np.sum(np.abs(coef)) * np.max(np.abs(real))
    ctx.dev_free(p_in)


def _image_stats(img, ref):
    d = np.abs(img.astype(np.int16) - ref.astype(np.int16))
    return int(d.max()), int(np.count_nonzero(d > 1)), float(d.mean())


def _oracle(x, fs, lpm):
    import tempfile
    with tempfile.TemporaryDirectory() as td:
        path = os.path.join(td, "x.wav")
        synth.write_wav(path, int(fs), x)
        return wo.process(path, lpm, want_messages=False)


# What separates these decodes from the reference is the front end alone: the exact FFT resampler applies the reference's own
# brick wall and everything after it is the exact path.  Round 3 (integer-exact ingest on the 2**-30 grid, float64 behind it):
# 179 of 182 random clips give the identical uint8 stream, the rest differ by 1 on at most 3 of ~440 000 samples, the image by at
# most 1 -- the north star's bar -- with start_frame equal (profiles/r03_v4/random_fe_parity.jsonl).  The tolerances below are that
# bar and several times the measured count (round 2, fp32 stencils: 19 of 441 000 samples, tolerance 5e-4).
STREAM_MAX, STREAM_NE_FRAC, IMAGE_MAX, IMAGE_GT1_FRAC = 1, 5e-5, 1, 0.0


@pytest.mark.gpu
@pytest.mark.parametrize("fs,iq,lpm,seconds,stop_rate,drop", [(48000, False, 240, 56.0, 16000, 0), (192000, True, 240, 64.0, 16000, 0),
                                                              (44100, False, 240, 56.0, 14700, 0), (1536000, True, 120, 40.0, 48000, 0),
                                                              (1536000, True, 120, 40.0, 16000, 0),
                                                              (48000, False, 240, 56.0, 16000, 3 * 4567), (1536000, True, 120, 40.0, 16000, 96 * 4321)])
def test_front_end_then_exact_path_one_gpu_and_sharded(ctx, fs, iq, lpm, seconds, stop_rate, drop):
    """Front end to the hand-over rate (16 000 Hz; 14 700 Hz from 44.1 kHz; 48 000 Hz straight behind the ingest) + the exact path: the one-GPU fused form, and the sharded form on 1, 2, 3 and 8 emulated
    ranks (bit-identical to each other and to the fused form); against the oracle within the figures above.  ``drop`` frames less
    than whole seconds: int(11025 * n0 / fs) is then not n0 * 11025 / fs and the reference's resampling grid (wefax.py:384) is
    stretched by up to one sample over the capture -- the hand-over keeps it (polyphase.FrontEnd.n_out)."""
    from wefax_amd import _native as nat
    x = _capture(fs, 0.05, seed=0, lpm=lpm, seconds=seconds, iq=iq)
    if drop:
        x = np.ascontiguousarray(x[:x.shape[0] - drop])
        assert (11025 * x.shape[0]) % fs != 0
    ref = _oracle(x, fs, lpm)
    fe = pp.FrontEnd(fs, stop_rate=stop_rate)
    dec = sharded.FrontEndExactDecoder(ctx, fe, x, lines_per_minute=lpm)
    dec.run()
    info = dec.result()
    img1, st1 = dec.fetch("image"), dec.fetch("digitalized")
    dec.close()
    assert info.start_frame == ref["start_frame"]
    peaks1 = [int(info.peak_pos[k]) for k in range(info.npeaks)]
    # (every chain is decimations only: integer-exact ingest where the first stage qualifies, float64 behind it.  The one-stage chain
    # /32 -> 48 kHz is a Kaiser design at 135 dB -- ripple 1.8e-7 -- and keeps a wider count)
    ne_frac = STREAM_NE_FRAC if (fe.design is not None or fs <= 200000) else 2e-4
    for name, got, want, mx, frac in (("stream", st1, ref["digitalized"], STREAM_MAX, ne_frac), ("image", img1, ref["image"], IMAGE_MAX, None)):
        d = np.abs(got.astype(np.int16) - want.astype(np.int16))
        print(f"fs={fs} via {stop_rate}: {name} max|d|={d.max()} differing={np.count_nonzero(d)} of {d.size}, >1: {np.count_nonzero(d > 1)}")
        assert got.shape == want.shape and d.max() <= mx
        if frac is not None:
            assert np.count_nonzero(d) <= frac * d.size
    assert np.count_nonzero(np.abs(img1.astype(np.int16) - ref["image"].astype(np.int16)) > 1) <= IMAGE_GT1_FRAC * img1.size
    first = None
    for world in (1, 2, 3, 8):
        mk = lambda c, m: sharded.FrontEndShardedDecoder(c, m, fe, x, lines_per_minute=lpm, plan="dist")      # noqa: E731
        # (a trimmed clip's hand-over length has no distributed plan -- the distributed transforms take 13-smooth halves only,
        # DESIGN.md 6 -- so it gets the single plan: rank 0 runs the front end over the whole stream and decodes alone; same result)
        assert sharded.layout_supported(fe.n_out(x.shape[0]), fe.out_rate, world, n_out=fe.n_target(x.shape[0]))
        r = sharded.decode_emulated(x, fs, world, lpm, make_decoder=mk)
        # (a peak of the reference may sit elsewhere where the stream differs by one grey level; the group that fixes start_frame does not)
        assert r["sync"]["start_frame"] == ref["start_frame"] and r["sync"]["peaks"] == peaks1
        assert np.array_equal(r["digitalized"], st1) and np.array_equal(r["image"], img1)
        assert np.array_equal(r["digitalized"], r["digitalized_blocks"])
        if first is None:
            first = r
        else:       # the float stages too do not depend on the number of ranks
            assert np.array_equal(r["envelope"], first["envelope"]) and np.array_equal(r["audio"], first["audio"])
            assert r["low"] == first["low"] and r["high"] == first["high"]
    # plan 3 (round 6): the hand-over-rate signal cut into contiguous arcs, resampler AND Hilbert transform by their multipole forms -- nothing
    # but kilobytes of weights, 320 resampled samples per seam and the select's collectives travel before the gather.  Where the lengths have
    # those forms (downsampling to an even count): the same bytes for every world size; against the fused form (transforms over the
    # capture: the same sums in another order, 1e-13 apart) the stream within the parity bar.
    n_fe, n_t = fe.n_out(x.shape[0]), fe.n_target(x.shape[0])
    if n_t % 2 == 0 and n_fe >= 32768 and n_t >= 32768:
        first = None
        for world in (1, 2, 3, 8):
            mk = lambda c, m: sharded.FrontEndShardedDecoder(c, m, fe, x, lines_per_minute=lpm, plan="fmm")      # noqa: E731
            r = sharded.decode_emulated(x, fs, world, lpm, make_decoder=mk)
            assert r["plan"] == 3
            assert r["sync"]["start_frame"] == ref["start_frame"]
            assert np.array_equal(r["digitalized"], r["digitalized_blocks"])
            if first is None:
                first = r
                d = np.abs(r["digitalized"].astype(np.int16) - st1.astype(np.int16))
                assert d.max() <= 1 and np.count_nonzero(d) <= 1e-5 * d.size + 2, (int(d.max()), int(np.count_nonzero(d)))
                assert np.abs(r["image"].astype(np.int16) - img1.astype(np.int16)).max() <= 1
            else:
                for k in ("digitalized", "image", "envelope", "audio"):
                    assert np.array_equal(r[k], first[k]), (world, k)
                assert r["low"] == first["low"] and r["high"] == first["high"]
            sent = [sum(int(e["sent"]) for e in ws if e["name"] != "stream gather") for ws in r["wire"]]
            assert max(sent) <= 1 << 20                  # a rank puts less than a megabyte on the wire in front of the gather
    else:
        with pytest.raises(nat.NativeError, match="no multipole form"):
            sharded.FrontEndShardedDecoder(ctx, nat.Comm.local(1)[0], fe, x, lines_per_minute=lpm, plan="fmm")


@pytest.mark.gpu
@pytest.mark.parametrize("layout", ["dist", "rows"])
def test_each_rank_loads_only_its_slice_of_the_raw_stream(ctx, layout):
    """The oversampled stream is split `world` ways and never moves: a rank asks its loader for its own frames plus the FIR
    chain's halo (a few thousand frames), indices wrapping modulo the capture at its two ends -- one range in the rows layout,
    one range per segment (225 of them) in the columns layout, whose halos add ~0.3 % per segment at the 60-minute size."""
    from wefax_amd import _native as nat
    fs = 192000
    x = _capture(fs, 0.05, seed=2, lpm=240, seconds=40.0, iq=True)
    fe = pp.FrontEnd(fs, stop_rate=pp.FrontEnd.handover_rate(fs))
    n0 = x.shape[0]
    comms = nat.Comm.local(4)
    asked = []
    decs = []
    for r in range(4):
        def loader(lo, hi, r=r):
            asked.append((r, lo, hi))
            return x[np.arange(lo, hi) % n0]
        decs.append(sharded.FrontEndShardedDecoder(ctx, comms[r], fe, None, n_in_total=n0, in_kind=nat.WFX_IN_I16_STEREO, lines_per_minute=240,
                                                   raw_loader=loader, plan=layout))
    halo = fe.halo()
    if layout == "rows":
        assert [a[0] for a in asked] == [0, 1, 2, 3]
        for (r, lo, hi), d in zip(asked, decs):
            share = int(np.ceil((d.layout.in_hi - d.layout.in_lo) * fs / fe.out_rate))       # the rank's rows at the hand-over rate, in raw frames
            assert hi - lo <= share + 2 * halo + 64 and lo < hi
        assert asked[0][1] < 0 and asked[-1][2] > n0                         # the ends wrap (the FFT resampler behind is circular)
    else:
        nseg = decs[0].layout.nseg
        assert nseg > 1 and all(d.layout.plan == 2 for d in decs) and len(asked) == 4 * nseg
        for r, d in enumerate(decs):
            mine = [(lo, hi) for rr, lo, hi in asked if rr == r]
            share = int(d.layout.in_seg_len) * fs // fe.out_rate                              # a segment at the hand-over rate, in raw frames
            assert all(hi - lo <= share + 2 * halo + 64 and lo < hi for lo, hi in mine)
            assert all(b[0] - a[0] == int(d.layout.in_seg_stride) * fs // fe.out_rate for a, b in zip(mine, mine[1:]))   # equally spaced
            assert sum(hi - lo for lo, hi in mine) == d.raw_frames
        assert min(lo for _, lo, _ in asked) < 0 and max(hi for _, _, hi in asked) > n0
    for d in decs:
        d.close()
    for c in comms:
        c.close()


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["iq1536k_2s_240", "stereo192k_6s_240", "mono48k_image_240", "stereo48k_image_240", "mono48k_noisy_120", "mono_noisy_240",
                                  "mono48k_f32_240"])
def test_demodulator_reaches_the_time_domain_front_end_on_request(name):
    """`Demodulator(front_end=...)`: the drop-in's two routes for a capture that is not at 11 025 Hz (wefax.py:351-394), on the
    REFERENCE's goldens.  "exact" (default) = the reference's own operators, bit-identical; "time-domain" = the decimator chain of
    BASELINE configs[3] (for 1.536 MS/s: the streaming / 32 -> / 3 kernel) + exact FFT resample from the hand-over rate: same
    exception, same start frame, stream and image within one grey level.  Captures the chain does not take (11 025 Hz, float32)
    run the exact route whatever was asked."""
    from conftest import golden_cases, input_by_name, load_golden
    from wefax_amd import Demodulator
    case = next(c for c in golden_cases() if c["name"] == name)
    g = load_golden(name)
    for mode in ("exact", "time-domain"):
        d = Demodulator(input_by_name(name), lines_per_minute=case["lpm"], quiet=True, tcp_stream=True, front_end=mode)
        try:
            d.process()
            exc = None
        except (ValueError, IndexError) as e:
            exc = [type(e).__name__, str(e)]
        takes = mode == "time-domain" and case["file_info"]["sample_rate"] >= 28000 and name != "mono48k_f32_240"
        assert d.front_end_used == ("time-domain" if takes else "exact")
        assert exc == case["exception"]
        assert d.sample_rate == case["sample_rate"] and d.length == case["length"]
        dig = d.digitalized_data
        assert dig.shape == g["digitalized"].shape
        delta = np.abs(dig.astype(np.int16) - g["digitalized"].astype(np.int16))
        exact = case.get("oracle_exact", True) and not takes
        print(f"{name} via {mode}: stream differing {np.count_nonzero(delta)} of {delta.size}, max {delta.max()}")
        assert delta.max() <= STREAM_MAX
        assert np.count_nonzero(delta) <= (0 if exact else max(3, 1e-3 * delta.size if not case.get("oracle_exact", True) else 2 * STREAM_NE_FRAC * delta.size))
        if exc is None:
            assert d.start_frame == case["start_frame"]
            img = d.output_array
            di = np.abs(img.astype(np.int16) - g["image"].astype(np.int16))
            assert img.shape == g["image"].shape and di.max() <= IMAGE_MAX
            assert not exact or not di.any()
        msgs = [[m.get("data_type"), m.get("progress_title", m.get("message_content")),
                 None if "percentage" not in m else float(m["percentage"])] for m in d.websocket_stack]
        assert msgs == [list(m) for m in case["websocket_stack"]] or (takes and exc is None and len(msgs) == len(case["websocket_stack"]))
        d.close()


def test_demodulator_front_end_argument_is_checked(tmp_path, monkeypatch):
    from wefax_amd import Demodulator
    p = tmp_path / "x.wav"
    synth.write_wav(str(p), 11025, np.zeros(100, dtype=np.int16))
    with pytest.raises(ValueError):
        Demodulator(str(p), quiet=True, front_end="fir")
    assert Demodulator(str(p), quiet=True).front_end == "exact"
    monkeypatch.setenv("WEFAX_FRONT_END", "time-domain")
    assert Demodulator(str(p), quiet=True).front_end == "time-domain"
    assert Demodulator(str(p), quiet=True, front_end="exact").front_end == "exact"
