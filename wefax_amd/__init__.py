"""wefax_amd -- MI355X-native WEFAX (HF radiofax) file decoding hot path.

Host side: Python mirroring the reference's ``wefax.Demodulator``; device side:
hand-written HIP kernels for gfx950 behind the C ABI of include/wefax_hip.h.
"""
from .wefax import Demodulator, DecodeJob  # noqa: F401

__all__ = ["Demodulator", "DecodeJob"]
