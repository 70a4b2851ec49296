"""The streaming ingest kernel (csrc/wfx_ingest.hip; round 5): the / 32 integer-exact stage of the oversampled front end, alone or
with the float64 stage behind it in the same kernel.

What is asserted: stage 1 EQUALS the int64 model of the fixed-point FIR for every input, the accumulators' worst case included;
the fused chain is BIT-IDENTICAL to the two tile-kernel launches it replaces (wfx_polyphase.hip), whatever the length, the run
length, the batch shape or the end of the buffer (the range-checked form of the kernel takes the last runs)."""
import os

import numpy as np
import pytest

from wefax_amd import polyphase as pp


@pytest.fixture(scope="module")
def ctx():
    from wefax_amd import _native as nat
    c = nat.Context(0)
    yield c
    c.close()


def _dev(ctx, a):
    p = ctx.dev_malloc(max(a.nbytes, 16) + 64)
    ctx.dev_upload(p, np.ascontiguousarray(a))
    return p


def _filters():
    fe = pp.FrontEnd(1536000)
    s1, s2 = fe.stages
    assert s1.factor == 32 and s1.fix_shift == 30 and s2.factor == 3
    return s1, s2


def _merged(raw):
    """int16 samples the kernel works on: IQ pairs merged with the int16 wrap of wefax.py:367 (the / 2 is in the scale)."""
    if raw.ndim == 2:
        return (raw[:, 0].astype(np.int32) + raw[:, 1].astype(np.int32)).astype(np.int16).astype(np.int64)
    return raw.astype(np.int64)


def _model_stage1(raw, coef, sh, n_out):
    fix = np.rint(coef * 2.0 ** sh).astype(np.int64)
    x = _merged(raw)
    need = (n_out - 1) * 32 + coef.shape[0]
    xp = np.zeros(need, dtype=np.int64)
    xp[:min(need, x.shape[0])] = x[:need]
    acc = np.zeros(n_out, dtype=np.int64)
    for j in range(coef.shape[0]):
        acc += fix[j] * xp[j:j + (n_out - 1) * 32 + 1:32]
    return acc.astype(np.float64) / 2.0 ** (sh + (1 if raw.ndim == 2 else 0))


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["iq", "i16"])
@pytest.mark.parametrize("worst", [False, True])
@pytest.mark.parametrize("n_out,slack", [(700, 0), (5000, 0), (5000, 40000), (40000, 3), (40000, 10 ** 6)])
def test_stage1_equals_the_integer_model(ctx, kind, worst, n_out, slack):
    from wefax_amd import _native as nat
    s1, _ = _filters()
    coef, sh = s1.coef64, s1.fix_shift
    rng = np.random.default_rng(n_out + slack)
    n_in = (n_out - 1) * 32 + coef.shape[0] + slack
    shape = (n_in, 2) if kind == "iq" else (n_in,)
    raw = rng.integers(-32768, 32768, size=shape).astype(np.int16)
    if worst:
        # full-scale samples (both byte planes at their extremes: -32768 = (-128, -128), 32767 = (127, 127)) whose signs follow the
        # taps -- and each of the four balanced byte digits a tap is split into (q0 + 2^8 q1 + 2^16 q2 + 2^24 q3, csrc/wfx_ingest.hip):
        # the windows of some outputs drive the int32 digit sums of every tap chunk to their largest magnitude, both signs
        fix = np.rint(coef * 2.0 ** sh).astype(np.int64)
        digits, v = [], fix.copy()
        for _ in range(4):
            d = ((v + 128) & 255) - 128
            v = (v - d) >> 8
            digits.append(d)
        assert not v.any() and np.array_equal(sum(d << (8 * q) for q, d in enumerate(digits)), fix)
        col = raw[:, 0] if kind == "iq" else raw
        if kind == "iq":
            raw[:, 1] = 0
        col[:] = -32768
        outs = [0, 9, 18, 27, 255 - 8, 256 + 1, 511 - 8, 512 + 1, 521, 530, 539, 548, 557, n_out - 10, n_out - 1]      # (around the ends of an iteration's block)
        q0, q1, q2, q3 = digits
        pats = [fix, -fix, q0, -q0, q1, -q1, q2, -q2, fix, q3, -q3, q0 + q1, -q2 - q3, -q0, -q2]
        for o, q in zip(outs, pats):       # (at least 9 outputs apart: the windows do not overlap)
            col[32 * o:32 * o + coef.shape[0]] = np.where(q >= 0, -32768, 32767).astype(np.int16)
    k = nat.WFX_IN_I16_STEREO if kind == "iq" else nat.WFX_IN_I16_MONO
    p_in, p_out = _dev(ctx, raw), ctx.dev_malloc(n_out * 8)
    assert ctx.d_ingest_chain(p_in, k, n_in, 32, coef, sh, 0, None, p_out, n_out)
    got = ctx.dev_download(p_out, (n_out,), np.float64)
    assert np.array_equal(got, _model_stage1(raw, coef, sh, n_out))
    # the generic entry point takes the same kernel when the first window sits on the 16-byte grid, and agrees with the tile kernel
    assert ctx.d_decimate_fir64(p_in, k, n_in, 0, 32, coef, p_out, n_out, sh)
    assert np.array_equal(ctx.dev_download(p_out, (n_out,), np.float64), got)
    os.environ["WFX_INGEST_TILE"] = "1"
    try:
        assert ctx.d_decimate_fir64(p_in, k, n_in, 0, 32, coef, p_out, n_out, sh)
    finally:
        del os.environ["WFX_INGEST_TILE"]
    assert np.array_equal(ctx.dev_download(p_out, (n_out,), np.float64), got)
    ctx.dev_free(p_in)
    ctx.dev_free(p_out)


def _two_launches(ctx, nat, p_in, k, n_in, s1, s2, n2, nbatch=1, in_stride=0, mid_stride=0, out_stride=0):
    n1 = (n2 - 1) * s2.factor + s2.ntaps
    p_mid, p_out = ctx.dev_malloc(8 * max(n1, mid_stride) * nbatch), ctx.dev_malloc(8 * max(n2, out_stride) * nbatch)
    os.environ["WFX_INGEST_TILE"] = "1"
    try:
        assert ctx.d_decimate_fir64(p_in, k, n_in, 0, s1.factor, s1.coef64, p_mid, n1, s1.fix_shift, nbatch=nbatch, in_stride=in_stride, out_stride=mid_stride)
    finally:
        del os.environ["WFX_INGEST_TILE"]
    assert not ctx.d_decimate_fir64(p_mid, nat.WFX_IN_F64_MONO, n1, 0, s2.factor, s2.coef64, p_out, n2, 0, nbatch=nbatch, in_stride=mid_stride, out_stride=out_stride)
    out = ctx.dev_download(p_out, (max(n2, out_stride) * nbatch,), np.float64)
    ctx.dev_free(p_mid)
    ctx.dev_free(p_out)
    return out


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["iq", "i16"])
@pytest.mark.parametrize("n2,slack,ni", [(1, 0, None), (131, 0, None), (133, 5, None), (3000, 0, None), (3000, 10 ** 6, 1), (60000, 0, 4), (60000, 10 ** 6, None),
                                         (400000, 77, 16), (400000, 3 * 10 ** 6, 16)])
def test_fused_chain_is_bit_identical_to_the_two_launches(ctx, kind, n2, slack, ni):
    from wefax_amd import _native as nat
    s1, s2 = _filters()
    rng = np.random.default_rng(n2)
    n1 = (n2 - 1) * s2.factor + s2.ntaps
    n_in = (n1 - 1) * 32 + s1.ntaps + slack
    raw = rng.integers(-32768, 32768, size=(n_in, 2) if kind == "iq" else (n_in,)).astype(np.int16)
    k = nat.WFX_IN_I16_STEREO if kind == "iq" else nat.WFX_IN_I16_MONO
    p_in, p_out = _dev(ctx, raw), ctx.dev_malloc(n2 * 8)
    if ni:
        os.environ["WFX_INGEST_NI"] = str(ni)
    try:
        assert ctx.d_ingest_chain(p_in, k, n_in, 32, s1.coef64, s1.fix_shift, s2.factor, s2.coef64, p_out, n2)
    finally:
        os.environ.pop("WFX_INGEST_NI", None)
    got = ctx.dev_download(p_out, (n2,), np.float64)
    want = _two_launches(ctx, nat, p_in, k, n_in, s1, s2, n2)
    assert np.array_equal(got, want)
    # and against float64 NumPy (not bit for bit: another order of additions)
    y1 = _model_stage1(raw, s1.coef64, s1.fix_shift, n1)
    idx = np.arange(min(n2, 2000)) * s2.factor
    ref = np.array([np.dot(s2.coef64, y1[i:i + s2.ntaps]) for i in idx])
    assert np.max(np.abs(got[:idx.shape[0]] - ref)) <= 1e-12 * np.max(np.abs(y1))
    ctx.dev_free(p_in)
    ctx.dev_free(p_out)


@pytest.mark.gpu
def test_fused_chain_with_a_factor_of_two_behind_the_ingest(ctx):
    from wefax_amd import _native as nat
    s1, _ = _filters()
    s2 = pp.Decimate(s1.fs_out, 2, pp.NYQ, float(s1.fs_out) / 2 - pp.NYQ, 120.0)
    n2 = 20000
    n1 = (n2 - 1) * 2 + s2.ntaps
    n_in = (n1 - 1) * 32 + s1.ntaps
    raw = np.random.default_rng(5).integers(-32768, 32768, size=(n_in, 2)).astype(np.int16)
    p_in, p_out = _dev(ctx, raw), ctx.dev_malloc(n2 * 8)
    assert ctx.d_ingest_chain(p_in, nat.WFX_IN_I16_STEREO, n_in, 32, s1.coef64, s1.fix_shift, 2, s2.coef64, p_out, n2)
    got = ctx.dev_download(p_out, (n2,), np.float64)
    assert np.array_equal(got, _two_launches(ctx, nat, p_in, nat.WFX_IN_I16_STEREO, n_in, s1, s2, n2))
    ctx.dev_free(p_in)
    ctx.dev_free(p_out)


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["iq", "i16"])
def test_fused_chain_in_batches(ctx, kind):
    """The segments of a rank's columns layout: equally shaped jobs a fixed stride apart in one launch."""
    from wefax_amd import _native as nat
    s1, s2 = _filters()
    nb, n2 = 5, 7001
    n1 = (n2 - 1) * 3 + s2.ntaps
    n_seg = (n1 - 1) * 32 + s1.ntaps
    in_stride = n_seg + 1000 - (n_seg + 1000) % 8
    out_stride = n2 + 3
    tot = in_stride * (nb - 1) + n_seg
    raw = np.random.default_rng(9).integers(-32768, 32768, size=(tot, 2) if kind == "iq" else (tot,)).astype(np.int16)
    k = nat.WFX_IN_I16_STEREO if kind == "iq" else nat.WFX_IN_I16_MONO
    p_in, p_out = _dev(ctx, raw), ctx.dev_malloc(out_stride * nb * 8)
    ctx.dev_upload(p_out, np.zeros(out_stride * nb))
    assert ctx.d_ingest_chain(p_in, k, n_seg, 32, s1.coef64, s1.fix_shift, 3, s2.coef64, p_out, n2, nbatch=nb, in_stride=in_stride, out_stride=out_stride)
    got = ctx.dev_download(p_out, (nb, out_stride), np.float64)
    want = _two_launches(ctx, nat, p_in, k, n_seg, s1, s2, n2, nbatch=nb, in_stride=in_stride, mid_stride=n1 + (n1 & 1), out_stride=out_stride).reshape(nb, out_stride)
    assert np.array_equal(got[:, :n2], want[:, :n2]) and not got[:, n2:].any()
    # each member equals a job of its own
    p1 = ctx.dev_malloc(n2 * 8)
    esz = 4 if kind == "iq" else 2
    for b in (0, nb - 1):
        assert ctx.d_ingest_chain(p_in + b * in_stride * esz, k, n_seg, 32, s1.coef64, s1.fix_shift, 3, s2.coef64, p1, n2)
        assert np.array_equal(ctx.dev_download(p1, (n2,), np.float64), got[b, :n2])
    ctx.dev_free(p1)
    ctx.dev_free(p_in)
    ctx.dev_free(p_out)


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["iq", "i16"])
@pytest.mark.parametrize("ntaps,bits,sh", [(256, 26, 30), (255, 26, 28), (253, 12, 12), (64, 27, 30), (17, 29, 33), (1, 29, 30), (200, 8, 8)])
def test_stage1_with_any_taps_on_the_grid(ctx, kind, ntaps, bits, sh):
    """Not only the front end's own low-pass: any <= 256 taps on the grid 2^-sh below 2^30 (sum |tap| < 2^35) go through the digit planes
    exactly -- random ones, full-range samples."""
    from wefax_amd import _native as nat
    rng = np.random.default_rng(ntaps * 100 + bits)
    fix = rng.integers(-(1 << bits) + 1, 1 << bits, size=ntaps).astype(np.int64)
    fix[rng.integers(0, ntaps, size=max(1, ntaps // 8))] = 0
    fix[0] = (1 << bits) - 1
    coef = fix.astype(np.float64) / 2.0 ** sh
    n_out = 2100
    n_in = (n_out - 1) * 32 + ntaps + 77
    raw = rng.integers(-32768, 32768, size=(n_in, 2) if kind == "iq" else (n_in,)).astype(np.int16)
    k = nat.WFX_IN_I16_STEREO if kind == "iq" else nat.WFX_IN_I16_MONO
    p_in, p_out = _dev(ctx, raw), ctx.dev_malloc(n_out * 8)
    assert ctx.d_ingest_chain(p_in, k, n_in, 32, coef, sh, 0, None, p_out, n_out)
    assert np.array_equal(ctx.dev_download(p_out, (n_out,), np.float64), _model_stage1(raw, coef, sh, n_out))
    ctx.dev_free(p_in)
    ctx.dev_free(p_out)


@pytest.mark.gpu
def test_shapes_the_streaming_kernel_declines(ctx):
    """Nothing is enqueued and False comes back: the caller runs the tile kernels (same results, see above)."""
    from wefax_amd import _native as nat
    s1, s2 = _filters()
    raw = np.zeros((70000, 2), dtype=np.int16)
    p_in, p_out = _dev(ctx, raw), ctx.dev_malloc(8 * 4096)
    k = nat.WFX_IN_I16_STEREO
    assert not ctx.d_ingest_chain(p_in, k, 70000, 16, s1.coef64[:125], 30, 3, s2.coef64, p_out, 10)            # another factor
    assert not ctx.d_ingest_chain(p_in + 4, k, 69999, 32, s1.coef64, 30, 3, s2.coef64, p_out, 10)             # not on the 16-byte grid
    assert not ctx.d_ingest_chain(p_in, k, 70000, 32, s1.coef64, 30, 5, s2.coef64, p_out, 10)                 # a factor behind it the kernel has no form for
    assert not ctx.d_ingest_chain(p_in, k, 70000, 32, np.full(253, 0.9), 30, 3, s2.coef64, p_out, 10)         # taps whose sums a float64 cannot hold exactly
    ctx.dev_free(p_in)
    ctx.dev_free(p_out)


@pytest.mark.gpu
def test_stream_rate_probe_and_placed_allocation(ctx):
    """The ingest's own access pattern as a read-rate probe (nothing is written), and the allocation helper built on it."""
    nbytes = 1 << 30
    p = ctx.dev_malloc(nbytes)
    g = ctx.d_stream_rate(p, nbytes)
    assert 500.0 < g < 8000.0
    with pytest.raises(Exception):
        ctx.d_stream_rate(p, 1 << 20)                  # too small to say anything
    ctx.dev_free(p)
    q, rates = ctx.dev_malloc_placed(nbytes, tries=3, good_gbs=1e9)       # (an unreachable bar: all three candidates are timed)
    assert len(rates) == 3 and all(500.0 < r < 8000.0 for r in rates)
    ctx.dev_upload(q, np.arange(16, dtype=np.int16))
    assert np.array_equal(ctx.dev_download(q, (16,), np.int16), np.arange(16, dtype=np.int16))
    ctx.dev_free(q)
    r, none = ctx.dev_malloc_placed(1 << 20, tries=3)                     # small buffers: the first allocation
    assert none == []
    ctx.dev_free(r)
