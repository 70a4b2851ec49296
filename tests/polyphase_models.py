"""Float64 NumPy models of the time-domain front end's stencil kernel (csrc/wfx_polyphase.hip): what
the GPU tests compare the kernels with, and what the CPU tests run the filter design through."""
import numpy as np

from oracle import wefax_oracle as wo


def _keys(v):
    u = np.ascontiguousarray(v, dtype=np.float64).view(np.uint64)
    neg = (u >> np.uint64(63)).astype(bool)
    return np.where(neg, ~u, u | np.uint64(1 << 63))


def to_real(raw):
    """int16 [n] -> float64; int16 [n, 2] -> the reference's stereo merge (wefax.py:360-373): int16 wrap, /2."""
    raw = np.asarray(raw)
    if raw.ndim == 2:
        return (raw[:, 0].astype(np.int16) + raw[:, 1].astype(np.int16)).astype(np.int16).astype(np.float64) / 2
    return raw.astype(np.float64)


def decimate_model(x, first, factor, coef, n_out):
    """float64 model of wfx_d_decimate_fir: y[i] = sum_j c[j] x[first + i*factor + j], zeros outside x."""
    x = np.asarray(x, dtype=np.float64)
    c = np.asarray(coef, dtype=np.float64)
    lo, hi = first, first + (n_out - 1) * factor + c.shape[0]
    xp = np.zeros(hi - lo)
    a, b = max(lo, 0), min(hi, x.shape[0])
    if b > a:
        xp[a - lo:b - lo] = x[a:b]
    y = np.zeros(n_out)
    for j in range(c.shape[0]):
        y += c[j] * xp[j:j + (n_out - 1) * factor + 1:factor]
    return y


def front_end_model(raw, chain):
    """The stage chain of polyphase.FrontEnd.chain on a raw slice, in float64 (decimations with their float64 taps)."""
    cur = to_real(raw)
    for st, (a, b), (ia, ib) in chain:
        assert cur.shape[0] == ib - ia and st.kind == "decimate"
        cur = decimate_model(cur, 0, st.factor, st.coef64, b - a)
    return cur
