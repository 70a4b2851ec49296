"""a7 without a transform over the capture (csrc/wfx_fmm.hip, hilbert mode WFX_HILBERT_FMM; round 5): the imaginary part of
scipy.signal.hilbert as a directly summed near field plus a fast multipole far field on 16 Chebyshev nodes per box.  Held against
the oracle's FFT form (= scipy's arithmetic) to 1e-12 relative -- the NumPy model of the same arithmetic (tools/farfield_model.py,
tests/test_farfield_model.py) reaches 1e-14 -- and, through the whole decode, against the reference's goldens: identical uint8
stream, peaks, start frame and image."""
import numpy as np
import pytest

from conftest import golden_cases, input_path, load_golden
from oracle import wefax_oracle as wo

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    from wefax_amd import _native as nat
    c = nat.Context(0)
    yield c
    c.close()


@pytest.mark.parametrize("n", [32768, 65536, 40000, 100002, 250008, 1433250, 7166250, 4194304 + 2])
def test_device_hilbert_by_fast_multipole_equals_the_transform_form(ctx, n):
    rng = np.random.default_rng(n)
    x = rng.standard_normal(n) * 1000 + 3000 * np.sin(np.arange(n) * 0.7)
    ref = wo.hilbert_fft(x)
    px, po = ctx.dev_malloc(n * 8 + 64), ctx.dev_malloc(n * 8 + 64)
    ctx.dev_upload(px, x)
    assert ctx.d_hilbert_fmm(px, n, po)
    got = ctx.dev_download(po, (n,), np.float64)
    scale = np.max(np.abs(ref.imag))
    assert np.max(np.abs(got - ref.imag)) <= 1e-12 * scale
    assert ctx.d_hilbert_fmm(px, n, po, True)
    env = ctx.dev_download(po, (n,), np.float64)
    assert np.max(np.abs(env - np.abs(ref))) <= 1e-12 * scale
    ctx.dev_free(px)
    ctx.dev_free(po)


def test_lengths_the_fast_multipole_form_does_not_take(ctx):
    p = ctx.dev_malloc(1 << 20)
    for n in (1000, 32766, 100001):          # short; below the smallest tree; odd
        assert not ctx.d_hilbert_fmm(p, n, p + (1 << 19))
    ctx.dev_free(p)


@pytest.mark.parametrize("name", ["mono_noisy_240", "mono_clean_120", "stereo_overflow_120", "mono48k_image_240", "mono_noise20_lead", "mono_noisy_120",
                                  "stereo48k_image_240", "mono_u8_240", "stereo_i32_240", "stereo48k_120", "stereo192k_6s_240", "mono48k_noisy_120",
                                  "mono_i24_240", "mono_f64_240", "three_ch_240", "iq1536k_2s_240"])
def test_whole_decode_with_the_fast_multipole_hilbert_matches_the_reference_goldens(name):
    """The reference's own streams: nothing moves when the decode takes the multipole forms (an odd-length capture -- mono_noisy_120 --
    runs the transform form under the same mode).  The 48 kHz and 192 kHz cases go through the RESAMPLER's multipole form as well
    (wfx_fmm.hip rs_*): scipy.signal.resample's output reproduced to 1e-13 without a transform, the reference's uint8 stream bit for bit."""
    from wefax_amd import Demodulator, _native as nat
    case = next(c for c in golden_cases() if c["name"] == name)
    g = load_golden(name)
    d = Demodulator(input_path(case), lines_per_minute=case["lpm"], quiet=True, tcp_stream=True, hilbert_mode=nat.WFX_HILBERT_FMM)
    try:
        d.process()
        exc = None
    except (ValueError, IndexError) as e:
        exc = [type(e).__name__, str(e)]
    assert exc == case["exception"]
    assert np.array_equal(d.digitalized_data, g["digitalized"])
    assert d.peaks == g["peaks"].tolist()
    st = case.get("float_stride", 5)
    scale = np.max(np.abs(g["demod_sub"]))
    assert np.max(np.abs(d.demodulated_data[::st] - g["demod_sub"])) <= 1e-11 * scale
    if exc is None:
        assert d.start_frame == case["start_frame"] and np.array_equal(d.output_array, g["image"])
    d.close()


def test_ten_minute_capture_both_hilbert_forms_give_one_stream():
    from wefax_amd import _native as nat, synth
    from wefax_amd.wefax import DecodeJob
    x = synth.config_c2(noise=0.05, seed=3)
    c = nat.Context(0)
    res = []
    for mode in (nat.WFX_HILBERT_FFT, nat.WFX_HILBERT_FMM):
        job = DecodeJob(c, x, 11025, 120, hilbert_mode=mode)
        job.run()
        info = job.result()
        res.append((job.fetch("digitalized"), job.fetch("envelope"), info.start_frame, job.fetch("image")))
    a, b = res
    assert np.array_equal(a[0], b[0]) and a[2] == b[2] and np.array_equal(a[3], b[3])
    assert np.max(np.abs(a[1] - b[1])) <= 1e-12 * np.max(a[1])
    c.close()


@pytest.mark.parametrize("n0,num", [(960000, 220500), (960000, 661500), (596801, 411220), (32768, 22580), (1000000, 999998), (3000000, 689062), (1048576, 524288),
                                    (65536 + 13, 45164)])
def test_device_resampler_by_fast_multipole_equals_the_transform_form(ctx, n0, num):
    """a5's scipy.signal.resample(x, num) (wefax.py:160-161) without a transform over the capture (round 6, csrc/wfx_fmm.hip rs_*): periodic sinc sum =
    one global constant minus a cotangent sum between the input grid and the output grid, near field on the vector pipe, far field on the Hilbert
    transform's tree.  Against the oracle's FFT form (= scipy's arithmetic) to 1e-12 of the largest sample (measured: 5e-14; the NumPy model of the
    same arithmetic, tests/test_resample_farfield_model.py, is gated at 1e-11).  Covers odd input counts, coincident grids (2 : 1), leaves of 32
    and of 64 samples, and the short-capture variant that keeps cot's x^3 term (levels < 13)."""
    rng = np.random.default_rng(n0 + num)
    x = rng.standard_normal(n0) * 1000 + 3000 * np.sin(np.arange(n0) * 0.7)
    ref = wo.resample_fft(x, num)
    px, py = ctx.dev_malloc(n0 * 8 + 64), ctx.dev_malloc(num * 8 + 64)
    ctx.dev_upload(px, x)
    assert ctx.d_resample_fmm(px, n0, num, py)
    got = ctx.dev_download(py, (num,), np.float64)
    assert np.max(np.abs(got - ref)) <= 1e-12 * np.max(np.abs(ref))
    assert np.array_equal(ctx.dev_download(px, (n0,), np.float64), x)          # the input is left alone
    ctx.dev_free(px)
    ctx.dev_free(py)


def test_device_resampler_declines_what_it_has_no_form_for(ctx):
    p = ctx.dev_malloc(1 << 21)
    for n0, num in ((100000, 100001), (100000, 200000), (100000, 68907), (20000, 13780)):       # upsampling, an odd count, a short capture
        assert not ctx.d_resample_fmm(p, n0, num, p + (1 << 20))
    ctx.dev_free(p)


@pytest.mark.parametrize("n", [2 * 1000003, 4000002])
def test_both_routes_stay_within_1e_12_of_scipy_at_a_general_length(n):
    """A GENERAL even length (half-length with a large prime factor) costs the transform route a padded convolution.  Its kernel's largest taps are
    the small NEGATIVE lags, laid out just below N: formed from r / N next to 1 they had lost N x 1e-16 of their relative accuracy (2e-10 .. 8e-10
    in the envelope at 14 .. 40 M samples, found in round 6 beside the multipole route's 8e-14) -- the lag is reduced to (-N/2, N/2] first now.
    Envelope + median of either route against scipy on the same filtered audio."""
    from scipy.signal import hilbert, medfilt
    from wefax_amd import _native as nat, synth
    from wefax_amd.wefax import DecodeJob
    x = synth.config_c2(noise=0.05, seed=5)[:n]
    assert x.shape[0] == n
    ref = None
    for mode in (nat.WFX_HILBERT_FFT, nat.WFX_HILBERT_FMM):
        c = nat.Context(0)
        job = DecodeJob(c, x, 11025, 120, hilbert_mode=mode)
        job.run()
        job.result()
        audio, env = job.fetch("audio"), job.fetch("envelope")
        if ref is None:
            ref = medfilt(np.abs(hilbert(audio)), 5)
        assert np.max(np.abs(env - ref)) <= 1e-12 * np.max(ref), mode
        del job
        c.close()


def test_auto_takes_the_multipole_route_for_a_resampled_capture_of_general_length():
    """A 48 kHz recording of arbitrary length: the transform-based resampler needs its chirp-z form (3.5x), the multipole forms do not care about the
    lengths' factors -- `auto` takes them (resampler AND Hilbert transform) where they exist; whole-second lengths keep the mixed-radix passes.
    The stream is the oracle's either way."""
    from wefax_amd import _native as nat, synth
    from wefax_amd.wefax import DecodeJob
    kw = dict(start_tone_s=2.0, phasing_lines=20, image_lines=30, stop_tone_s=1.0, black_tail_s=2.0)
    x = synth.synth_capture(48000.0, noise=0.05, seed=7, **kw)
    assert x.shape[0] == 1440000
    import os, tempfile
    for drop, want in ((14, nat.WFX_HILBERT_FMM), (0, nat.WFX_HILBERT_FFT), (10, nat.WFX_HILBERT_FFT)):
        xs = np.ascontiguousarray(x[:x.shape[0] - drop])
        c = nat.Context(0)
        job = DecodeJob(c, xs, 48000, 120)
        assert job.hilbert_mode == want, (drop, job.n, job.hilbert_mode)
        job.run()
        info = job.result()
        with tempfile.TemporaryDirectory() as td:
            path = os.path.join(td, "x.wav")
            synth.write_wav(path, 48000, xs)
            ref = wo.process(path, 120, want_messages=False)
        assert info.start_frame == ref["start_frame"] and np.array_equal(job.fetch("digitalized"), ref["digitalized"])
        del job
        c.close()

