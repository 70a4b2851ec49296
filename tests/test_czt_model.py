"""The arithmetic of the any-length resampler (csrc/wfx_mrfft.hip: wfx_dev_resample_czt) restated in NumPy, step by
step as the kernels do it -- packed samples, a chirp-z transform for the kept bins only, scipy's bin rules, the inverse chirp-z
transform to pairs of output samples -- and held against the oracle's `resample_fft` (= scipy.signal.resample, wefax.py:384) for
every parity of input and output length, down- and up-sampling.  No GPU: this pins the formulas; the kernels are held against the
oracle in tests/test_gpu_parity.py.
"""
import numpy as np
import pytest

from oracle import wefax_oracle as wo


def _chirp(m, n, sign):
    """e^{sign 2 pi i m^2 / n} with m^2 reduced modulo n in integers first (czt_chirp)."""
    m = np.asarray(m, dtype=np.int64)
    r = (m * m) % n
    return np.exp(sign * 2j * np.pi * r.astype(np.float64) / n)


def _cyclic_conv(a, b, m):
    return np.fft.ifft(np.fft.fft(a, m) * np.fft.fft(b, m))


def czt_resample_model(x, num, slack1=0, slack2=0):
    x = np.asarray(x, dtype=np.float64)
    n0 = x.shape[0]
    l1, p = (n0 + 1) // 2, (num + 1) // 2
    h = min(n0, num) // 2
    # ---- forward: Z[k] = sum_q z[q] e^{-4 pi i q k / n0}, k in [-h, h], as one cyclic convolution of M1 >= L1 + 2h points
    m1 = l1 + 2 * h + slack1
    xe = np.zeros(2 * l1)
    xe[:n0] = x
    q = np.arange(l1)
    a = (xe[0::2] + 1j * xe[1::2]) * _chirp(q, n0, -1)                       # czt_prologue
    b = np.zeros(m1, dtype=complex)                                           # czt_fill_b1: conj c1 at lags [-(h + L1 - 1), h]
    lag = np.arange(0, h + 1)
    b[lag] = _chirp(lag, n0, +1)
    lag = np.arange(1, h + l1)
    b[m1 - lag] = _chirp(-lag, n0, +1)
    c = _cyclic_conv(np.concatenate([a, np.zeros(m1 - l1)]), b, m1)
    # ---- bins (czt_glue)
    k = np.arange(h + 1)
    ck = _chirp(k, n0, -1)
    zk = ck * c[k]
    zm = ck * c[(m1 - k) % m1]
    e = 0.5 * (zk + np.conj(zm))
    o = (zk - np.conj(zm)) / 2j
    xk = e + np.exp(-2j * np.pi * k / n0) * o                                # rfft(x)[k]
    nmin = min(n0, num)
    if nmin % 2 == 0:
        xk[h] *= 2.0 if num < n0 else (0.5 if num > n0 else 1.0)
    yp, ym = xk.copy(), np.conj(xk)
    yp[0] = ym[0] = xk[0].real
    if num % 2 == 0 and 2 * h == num:
        yp[h] = ym[h] = 0.5 * xk[h].real
    gp = yp * (1 + 1j * np.exp(2j * np.pi * k / num))
    gm = ym * (1 + 1j * np.exp(-2j * np.pi * k / num))
    c2k = _chirp(k, num, +1)
    m2 = p + 2 * h + slack2
    a2 = np.zeros(m2, dtype=complex)
    a2[h + k] = gp * c2k / n0
    a2[h - k[1:]] = (gm * c2k)[1:] / n0
    # ---- inverse: u[p] = sum_k G[k] e^{4 pi i k p / num} as one cyclic convolution of M2 >= P + 2h points
    b2 = np.zeros(m2, dtype=complex)                                          # czt_fill_b2: conj c2[m + h] at lags [-2h, P - 1]
    lag = np.arange(0, p)
    b2[lag] = _chirp(lag + h, num, -1)
    lag = np.arange(1, 2 * h + 1)
    b2[m2 - lag] = _chirp(-lag + h, num, -1)
    c2 = _cyclic_conv(a2, b2, m2)
    u = _chirp(np.arange(p), num, +1) * c2[:p]                                # czt_epilogue
    out = np.empty(2 * p)
    out[0::2] = u.real
    out[1::2] = u.imag
    return out[:num]


@pytest.mark.parametrize("n0,num", [(400, 92), (401, 92), (400, 91), (401, 91), (92, 400), (91, 401), (92, 401), (91, 400),
                                    (300, 300), (301, 301), (300, 301), (301, 300), (4801, 1102), (1102, 4801), (2, 2), (3, 5), (17, 4)])
def test_chirp_z_form_equals_scipy_resample(n0, num):
    rng = np.random.default_rng(n0 * 7 + num)
    x = rng.standard_normal(n0) * 1000.0
    want = wo.resample_fft(x, num)
    got = czt_resample_model(x, num)
    assert got.shape == want.shape
    assert np.max(np.abs(got - want)) <= 1e-9 * max(1.0, np.max(np.abs(want)))
    # convolution lengths above the bound (the 13-smooth lengths the library picks) change nothing
    got2 = czt_resample_model(x, num, slack1=13, slack2=7)
    assert np.max(np.abs(got2 - want)) <= 1e-9 * max(1.0, np.max(np.abs(want)))


def test_convolution_lengths_of_the_60_minute_capture():
    """What the form saves: 2 N0 - 1 = 345.6 M points (536.9 M as a power of two) for the forward transform of rounds 1-3's Bluestein
    form; L1 + 2h = 126.1 M here."""
    n0, num = 172799998, 39689999
    l1, p, h = (n0 + 1) // 2, (num + 1) // 2, min(n0, num) // 2
    assert l1 + 2 * h == 126089997 and p + 2 * h == 59534998
    assert (l1 + 2 * h) < 0.37 * (2 * n0 - 1)
