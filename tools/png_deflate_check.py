#!/usr/bin/env python3
"""Device deflate PNG (wfx_decode_png_ex, deflate = 1) against zlib: the file decodes to the decoder's image, its size beside the
stored form and beside zlib level 6 on the same Up-filtered bytes; timing of both forms on the 10-minute capture.

    python tools/png_deflate_check.py [--full]
"""
import argparse
import glob
import io
import json
import os
import struct
import sys
import time
import zlib

import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from wefax_amd import Demodulator, synth      # noqa: E402


def read_png_gray8(blob: bytes) -> np.ndarray:
    assert blob[:8] == b"\x89PNG\r\n\x1a\n"
    pos, idat, w, h = 8, b"", 0, 0
    while pos < len(blob):
        n, tag = struct.unpack(">I4s", blob[pos:pos + 8])
        data = blob[pos + 8:pos + 8 + n]
        crc, = struct.unpack(">I", blob[pos + 8 + n:pos + 12 + n])
        assert zlib.crc32(tag + data) & 0xFFFFFFFF == crc, tag
        if tag == b"IHDR":
            w, h, depth, ctype, _, _, inter = struct.unpack(">IIBBBBB", data)
            assert (depth, ctype, inter) == (8, 0, 0)
        elif tag == b"IDAT":
            idat += data
        pos += 12 + n
    raw = np.frombuffer(zlib.decompress(idat), np.uint8).reshape(h, w + 1)
    out = np.zeros((h, w), np.uint8)
    for y in range(h):
        f = raw[y, 0]
        assert f in (0, 2)
        out[y] = raw[y, 1:] if f == 0 or y == 0 else raw[y, 1:] + out[y - 1]
    return out


def up_filtered(img: np.ndarray) -> bytes:
    raw = np.empty((img.shape[0], img.shape[1] + 1), np.uint8)
    raw[:, 0] = 2
    raw[0, 1:] = img[0]
    np.subtract(img[1:], img[:-1], out=raw[1:, 1:])
    return raw.tobytes()


def check(d, name):
    img = d.output_array
    t0 = time.perf_counter()
    stored = d._ctx.decode_png()
    t1 = time.perf_counter()
    comp = d._ctx.decode_png(deflate=True)
    t2 = time.perf_counter()
    comp2 = d._ctx.decode_png(deflate=True)
    t3 = time.perf_counter()
    ok = bool(np.array_equal(read_png_gray8(comp), img)) and comp == comp2
    z6 = len(zlib.compress(up_filtered(img), 6))
    rec = dict(case=name, shape=list(img.shape), ok=ok, stored=len(stored), deflate=len(comp), zlib6_up=z6, ratio_vs_stored=round(len(comp) / len(stored), 4),
               ratio_vs_zlib6=round(len(comp) / z6, 4), ms_stored=round((t1 - t0) * 1e3, 2), ms_deflate_first=round((t2 - t1) * 1e3, 2), ms_deflate=round((t3 - t2) * 1e3, 2))
    print(json.dumps(rec), flush=True)
    return ok


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--full", action="store_true", help="also the 10-minute capture of the benchmark")
    a = ap.parse_args()
    bad = 0
    gold = os.path.join(REPO, "tests", "golden")
    sys.path.insert(0, gold)
    import recipes
    recipes.ensure_all(gold)
    man = json.load(open(os.path.join(gold, "manifest.json")))
    for c in man["cases"] if isinstance(man, dict) else man:
        name, lpm = c["name"], c.get("lpm", 120)
        g = recipes.load_golden(gold, name)
        if "image" not in g.files or g["image"].ndim != 2 or g["image"].size == 0:
            continue
        d = Demodulator(os.path.join(gold, "inputs", name + ".wav"), lines_per_minute=lpm, quiet=True)
        d.process()
        bad += 0 if check(d, name) else 1
        d.close()
    if a.full:
        import tempfile
        with tempfile.TemporaryDirectory() as td:
            for noise in (0.05, 0.0):
                x = synth.synth_capture(11025.0, noise=noise, seed=1)
                path = os.path.join(td, "c.wav")
                synth.write_wav(path, 11025, x)
                d = Demodulator(path, lines_per_minute=120, quiet=True)
                d.process()
                bad += 0 if check(d, f"ten_minutes_noise_{noise}") else 1
                for defl in (False, True):
                    out = os.path.join(td, "o.png")
                    ts = []
                    for _ in range(5):
                        t0 = time.perf_counter()
                        d._ctx.decode_save_png(out, deflate=defl)
                        ts.append((time.perf_counter() - t0) * 1e3)
                    print(json.dumps(dict(case=f"save noise {noise}", deflate=defl, bytes=os.path.getsize(out), ms=[round(t, 2) for t in ts])), flush=True)
                d.close()
    if bad:
        raise SystemExit(f"{bad} case(s) failed")


if __name__ == "__main__":
    main()
