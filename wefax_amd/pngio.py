"""Fast PNG writer for the decoder's output (8-bit grayscale, w x 4h) -- SURVEY.md 8f-1.

After the kernels the slowest step of ``python wefax.py in.wav 120 out.png`` is
``PIL.Image.save`` (single-threaded zlib level 6 over tens of megabytes,
/root/reference/wefax.py:407-408).  The file written here is a standard PNG with the same
pixel layout (colour type 0, bit depth 8, no interlace); only the encoder differs: rows are
"Up"-filtered with numpy (the 4x vertical interpolation makes consecutive rows nearly
equal), the filtered bytes are cut into row bands, and every band is deflated on its own
thread (zlib releases the GIL) as a raw stream ending in a sync flush, pigz style.
"""
from __future__ import annotations

import os
import struct
import zlib
from concurrent.futures import ThreadPoolExecutor

import numpy as np

_SIG = b"\x89PNG\r\n\x1a\n"


def _chunk(tag: bytes, data: bytes) -> bytes:
    return struct.pack(">I", len(data)) + tag + data + struct.pack(">I", zlib.crc32(tag + data) & 0xFFFFFFFF)


def encode_png_gray8(img: np.ndarray, level: int = 1, threads: int | None = None, band_rows: int | None = None) -> bytes:
    img = np.ascontiguousarray(img, dtype=np.uint8)
    if img.ndim != 2:
        raise ValueError("expected a 2-D uint8 image")
    h, w = img.shape
    if h == 0 or w == 0:
        raise ValueError("height and width must be > 0")
    raw = np.empty((h, w + 1), dtype=np.uint8)
    raw[:, 0] = 2                       # filter type 2 (Up) on every row
    raw[0, 1:] = img[0]                 # the row above the first one is all zeros
    np.subtract(img[1:], img[:-1], out=raw[1:, 1:])      # uint8 arithmetic wraps modulo 256, as the filter specifies
    threads = threads or min(64, os.cpu_count() or 1)
    if band_rows is None:
        band_rows = max(16, -(-h // (threads * 2)))
    bands = [raw[r:r + band_rows] for r in range(0, h, band_rows)]

    def deflate(i):
        c = zlib.compressobj(level, zlib.DEFLATED, -15)
        out = c.compress(bands[i].tobytes())
        return out + c.flush(zlib.Z_FINISH if i == len(bands) - 1 else zlib.Z_SYNC_FLUSH)

    if threads > 1 and len(bands) > 1:
        with ThreadPoolExecutor(max_workers=threads) as ex:
            parts = list(ex.map(deflate, range(len(bands))))
    else:
        parts = [deflate(i) for i in range(len(bands))]
    adler = zlib.adler32(raw.tobytes()) & 0xFFFFFFFF
    stream = b"\x78\x01" + b"".join(parts) + struct.pack(">I", adler)
    ihdr = struct.pack(">IIBBBBB", w, h, 8, 0, 0, 0, 0)
    return _SIG + _chunk(b"IHDR", ihdr) + _chunk(b"IDAT", stream) + _chunk(b"IEND", b"")


def write_png_gray8(path: str, img: np.ndarray, level: int = 1, threads: int | None = None) -> int:
    data = encode_png_gray8(img, level, threads)
    with open(path, "wb") as fh:
        fh.write(data)
    return len(data)
