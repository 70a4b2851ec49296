"""BASELINE.json configs that the golden fixtures do not reach, and the drop-in entry points end to end.  Needs an MI355X.

  * configs[4]: a batch of independent captures with mixed 120 / 240 LPM, IOC576 / 288, decoded CONCURRENTLY on one native
    context (= HIP stream) each; every image bit-equal to the oracle's.
  * configs[2]: the resampler's first pass reading the int16 capture IN PLACE (csrc/wfx_api.hip, `in_place16`), through the
    fused decode, at a 30-second size and at the full 60-minute size (172.8 M samples).
  * the CLI `python wefax.py in.wav LPM out.png` (/root/reference/wefax.py:411-424): the PNG on disk decodes to the golden image.
  * filtfilt's odd extension in the capture's own dtype (wefax.py:72) as a stage call.
"""
import json
import os
import subprocess
import sys
import tempfile
import threading
import zlib

import numpy as np
import pytest

from conftest import GOLDEN, REPO, load_golden, input_by_name

pytestmark = pytest.mark.gpu


def _oracle(x, sr, lpm):
    from oracle import wefax_oracle as wo
    from wefax_amd import synth
    with tempfile.TemporaryDirectory() as td:
        path = os.path.join(td, "x.wav")
        synth.write_wav(path, sr, x)
        return wo.process(path, lpm, want_messages=False)


def test_batch_of_mixed_captures_on_concurrent_contexts():
    """BASELINE configs[4] members 0..7 (all four LPM / IOC combinations, twice), one context and one host thread each, three
    decodes per capture back to back so that the streams really overlap."""
    from wefax_amd import _native as nat
    from wefax_amd import synth
    from wefax_amd.wefax import DecodeJob
    members = [synth.config_c5_member(i, noise=0.05) for i in range(8)]
    assert sorted({(lpm, x.shape[0]) for x, lpm in members}) == [(120, 3858750), (120, 7166250), (240, 3858750), (240, 7166250)]
    refs = [_oracle(x, 11025, lpm) for x, lpm in members]
    ctxs = [nat.Context(0) for _ in members]
    jobs = [DecodeJob(c, x, 11025, lpm) for c, (x, lpm) in zip(ctxs, members)]
    errors = []

    def work(j):
        try:
            for _ in range(3):
                j.run()
            j.result()
        except Exception as e:           # noqa: BLE001
            errors.append(e)

    threads = [threading.Thread(target=work, args=(j,)) for j in jobs]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors
    for k, (job, ref, (x, lpm)) in enumerate(zip(jobs, refs, members)):
        info = job.result()
        assert info.start_frame == ref["start_frame"], f"member {k}"
        assert [int(info.peak_pos[i]) for i in range(info.npeaks)] == [int(v) for v in ref["peaks"]], f"member {k}"
        assert np.array_equal(job.fetch("digitalized"), ref["digitalized"]), f"member {k}: uint8 stream"
        img = job.fetch("image")
        assert img.shape == ref["image"].shape == (4 * info.height, 5512 if lpm == 120 else 2756)
        assert np.array_equal(img, ref["image"]), f"member {k}: image"
    for c in ctxs:
        c.close()


def test_all_64_members_of_the_batch_eight_contexts_at_a_time():
    """BASELINE configs[4] in full: 64 captures (mixed 120 / 240 LPM, IOC576 / 288, seeds 0..63) decoded 8 at a time on 8 contexts
    that are REUSED round after round (what one GPU of the 8-GPU configuration does: 8 captures per GPU stream set; captures are
    independent objects, nothing is exchanged).  A context keeps its plans and buffers while captures of different lengths and
    line rates pass through it; every sixth member is checked against the oracle in full, every member decodes without error,
    and a member decoded again on another context gives the same bytes."""
    from wefax_amd import _native as nat
    from wefax_amd import synth
    from wefax_amd.wefax import DecodeJob
    ctxs = [nat.Context(0) for _ in range(8)]
    digests = {}
    for rnd in range(8):
        members = [synth.config_c5_member(8 * rnd + k, noise=0.05) for k in range(8)]
        jobs = [DecodeJob(ctxs[(k + rnd) % 8], x, 11025, lpm) for k, (x, lpm) in enumerate(members)]      # contexts see other shapes each round
        errors = []

        def work(j):
            try:
                j.run()
                j.result()
            except Exception as e:           # noqa: BLE001
                errors.append(e)

        threads = [threading.Thread(target=work, args=(j,)) for j in jobs]
        for t in threads:
            t.start()
        for t in threads:
            t.join()
        assert not errors, errors
        for k, (job, (x, lpm)) in enumerate(zip(jobs, members)):
            i = 8 * rnd + k
            info = job.result()
            stream = job.fetch("digitalized")
            digests[i] = (int(info.start_frame), int(info.height), int(stream.astype(np.uint64).sum()), int(stream[::997].astype(np.uint64).sum()))
            assert info.width == (5512 if lpm == 120 else 2756) and info.n == x.shape[0]
            if i % 6 == 0:
                ref = _oracle(x, 11025, lpm)
                assert info.start_frame == ref["start_frame"], f"member {i}"
                assert np.array_equal(stream, ref["digitalized"]), f"member {i}: uint8 stream"
                assert np.array_equal(job.fetch("image"), ref["image"]), f"member {i}: image"
    # a few members again, on other contexts, after everything else has been through them
    for i in (3, 29, 62):
        x, lpm = synth.config_c5_member(i, noise=0.05)
        job = DecodeJob(ctxs[(i * 5) % 8], x, 11025, lpm)
        job.run()
        info = job.result()
        stream = job.fetch("digitalized")
        assert digests[i] == (int(info.start_frame), int(info.height), int(stream.astype(np.uint64).sum()), int(stream[::997].astype(np.uint64).sum()))
    assert len(digests) == 64
    for c in ctxs:
        c.close()


def test_int16_capture_is_resampled_in_place_thirty_seconds():
    """n0 = 1 440 000 int16 samples at 48 kHz: even, 13-smooth halves -> the mixed-radix resampler whose first pass reads the
    int16 pairs directly (no float64 copy of the capture).  Against the oracle: identical stream / peaks / image; and the
    float64 route (WFX_NO_I16_RESAMPLE=1) gives the same bytes."""
    from wefax_amd import _native as nat
    from wefax_amd import synth
    from wefax_amd.wefax import DecodeJob
    x = synth.synth_capture(48000.0, noise=0.05, seed=4, start_tone_s=2.0, phasing_lines=20, image_lines=30, stop_tone_s=1.0, black_tail_s=2.0)
    assert x.shape[0] == 1440000 and x.dtype == np.int16
    ref = _oracle(x, 48000, 120)
    ctx = nat.Context(0)
    job = DecodeJob(ctx, x, 48000, 120)
    job.run()
    info = job.result()
    stream, audio = job.fetch("digitalized"), job.fetch("audio")
    assert np.array_equal(stream, ref["digitalized"])
    assert np.max(np.abs(audio - ref["audio"])) <= 1e-9 * np.max(np.abs(ref["audio"]))
    assert info.start_frame == ref["start_frame"] and np.array_equal(job.fetch("image"), ref["image"])
    os.environ["WFX_NO_I16_RESAMPLE"] = "1"
    try:
        job2 = DecodeJob(ctx, x, 48000, 120)
        job2.run()
        assert np.array_equal(job2.fetch("digitalized"), stream)
        assert np.max(np.abs(job2.fetch("audio") - audio)) <= 1e-12 * np.max(np.abs(audio))
    finally:
        del os.environ["WFX_NO_I16_RESAMPLE"]
    ctx.close()


@pytest.mark.parametrize("fs", [48000.0, 11025.0])
def test_non_temporal_loads_in_the_transform_passes_change_nothing(fs, monkeypatch):
    """The passes read arrays beyond the Infinity Cache with non-temporal loads (mr_pass_desc::nt_in, compile-time variants of the
    kernels): forced on for a small capture (WFX_MR_NT=1) and forced off (0), the same bytes and the same float64 audio come out."""
    from wefax_amd import _native as nat
    from wefax_amd import synth
    from wefax_amd.wefax import DecodeJob
    x = synth.synth_capture(fs, noise=0.05, seed=11, start_tone_s=2.0, phasing_lines=20, image_lines=40, stop_tone_s=1.0, black_tail_s=2.0)
    ctx = nat.Context(0)
    got = {}
    for mode in ("0", "1"):
        monkeypatch.setenv("WFX_MR_NT", mode)
        job = DecodeJob(ctx, x, int(fs), 120)
        job.run()
        info = job.result()
        got[mode] = (int(info.start_frame), job.fetch("digitalized"), job.fetch("audio"), job.fetch("image"))
    monkeypatch.delenv("WFX_MR_NT")
    assert got["0"][0] == got["1"][0]
    for k in (1, 2, 3):
        assert np.array_equal(got["0"][k], got["1"][k])
    ref = _oracle(x, int(fs), 120)
    assert np.array_equal(got["1"][1], ref["digitalized"]) and np.array_equal(got["1"][3], ref["image"])
    ctx.close()


def test_sixty_minute_48k_capture_full_size():
    """BASELINE configs[2] at full size: 172 800 000 int16 samples (synthesised on the device, checked against the NumPy
    generator elsewhere) -> 39 690 000 at 11 025 Hz -> 5512 x 28 784 image.  The oracle needs ~10 s for it."""
    from wefax_amd import _native as nat
    from wefax_amd import synth_device
    from wefax_amd.wefax import DecodeJob
    ctx = nat.Context(0)
    sp = synth_device.synth_params(48000.0, noise=0.05, seed=0, iq=False, image_lines=7110, black_tail_s=5.0)
    n0 = int(ctx.lib.wfx_synth_frames(sp))
    assert n0 == 172800000
    ptr = synth_device.synth_slice(ctx, sp, 0, n0)
    x = ctx.dev_download(ptr, (n0,), np.int16)
    ctx.dev_free(ptr)
    job = DecodeJob(ctx, x, 48000, 120)
    assert job.n == 39690000
    job.run()
    info = job.result()
    ref = _oracle(x, 48000, 120)
    assert info.start_frame == ref["start_frame"]
    assert np.array_equal(job.fetch("digitalized"), ref["digitalized"])
    img = job.fetch("image")
    assert img.shape == ref["image"].shape and np.array_equal(img, ref["image"])
    ctx.close()


@pytest.mark.parametrize("trim", [1, 2])
def test_ten_minute_48k_capture_of_arbitrary_length(trim):
    """A 48 kHz recording stopped by hand: a 570-second capture less one sample (N0 odd) and less two (N0 / 2 = 13 679 999 =
    1229 x 11 131; the reference's output length 6 284 249 odd).  The resampler's chirp-z form on
    21 M-point convolutions, the int16 samples read in place, the Hilbert transform in its odd-length form behind it -- stream,
    start frame and image equal to the oracle's."""
    from wefax_amd import _native as nat
    from wefax_amd import synth
    from wefax_amd.wefax import DecodeJob
    x = synth.synth_capture(48000.0, noise=0.05, seed=7, start_tone_s=5.0, phasing_lines=60, image_lines=1060, stop_tone_s=2.0, black_tail_s=3.0)
    assert x.shape[0] == 27360000                                  # 570 s: 13-smooth halves as generated
    x = np.ascontiguousarray(x[:x.shape[0] - trim])
    ctx = nat.Context(0)
    job = DecodeJob(ctx, x, 48000, 120)
    assert job.n == 6284249                                        # int(11025 * (n0 / 48000)): odd
    job.run()
    info = job.result()
    ref = _oracle(x, 48000, 120)
    assert info.start_frame == ref["start_frame"]
    assert np.array_equal(job.fetch("digitalized"), ref["digitalized"])
    img = job.fetch("image")
    assert img.shape == ref["image"].shape and np.array_equal(img, ref["image"])
    ctx.close()


def test_sixty_minute_iq_stream_full_size_properties():
    """BASELINE configs[3] at FULL size: one 60-minute 1.536 MS/s int16 IQ stream, 5 529 600 000 frames = 22 GB synthesised in
    HBM.  Neither the reference (wefax.py:360-373 merges per sample in Python) nor the oracle can process it, so the checks are
    properties: (1) two different float64 front ends -- hand-over at 16 000 Hz (integer-exact /32 on the least-squares pair + float64 /3)
    and at 48 000 Hz (one integer-exact Kaiser /32 and nothing else: the exact FFT resampler then works on 172.8 M samples, the
    size of configs[2]) -- in front of the same exact path give the same start_frame, the same picture size and pictures that
    agree within one grey level on every one of 158 M pixels; (2) the sharded form on 8 emulated ranks, each synthesising
    only its own 1/8 of the stream, reproduces the one-GPU decode bit for bit (uint8 stream and image)."""
    from wefax_amd import _native as nat
    from wefax_amd import polyphase as pp
    from wefax_amd import sharded, synth_device
    fs = 1536000
    sp = synth_device.synth_params(float(fs), noise=0.05, seed=0, iq=True, start_tone_s=5.0, phasing_lines=60, image_lines=7110,
                                   stop_tone_s=5.0, black_tail_s=5.0)
    ctx = nat.Context(0)
    n0 = int(ctx.lib.wfx_synth_frames(sp))
    assert n0 == 3600 * fs

    def loader_on(c, keep):
        def load(lo, hi):
            ptr = synth_device.synth_slice(c, sp, lo, hi)
            keep.append((c, ptr))
            return ptr, hi - lo
        return load

    results = {}
    for rate in (16000, 48000):
        keep = []
        fe = pp.FrontEnd(fs, stop_rate=rate)
        dec = sharded.FrontEndExactDecoder(ctx, fe, None, n_in_total=n0, in_kind=nat.WFX_IN_I16_STEREO, lines_per_minute=120,
                                           raw_loader=loader_on(ctx, keep))
        assert dec.n == 39690000
        dec.run()
        info = dec.result()
        results[rate] = (int(info.start_frame), int(info.height), int(info.width), dec.fetch("digitalized"), dec.fetch("image"),
                         [int(info.peak_pos[k]) for k in range(info.npeaks)])
        assert dec.fe.exact_ingest is True
        dec.close()
        for c, p in keep:
            c.dev_free(p)
    s16, h16, w16, st16, img16, pk16 = results[16000]
    s22, h22, w22, st22, img22, _ = results[48000]
    assert (w16, h16) == (5512, (39690000 - s16) // 5512) and img16.shape == (4 * h16, w16)
    assert s16 == s22 and (h16, w16) == (h22, w22)
    d = np.abs(img16.astype(np.int16) - img22.astype(np.int16))
    ds = np.abs(st16.astype(np.int16) - st22.astype(np.int16))
    print(f"16 kHz vs 48 kHz hand-over, full size: stream differs on {np.count_nonzero(ds)} of {ds.size} (max {ds.max()}), "
          f"image on {np.count_nonzero(d)} of {d.size} (max {d.max()}, > 1: {np.count_nonzero(d > 1)})")
    assert ds.max() <= 1 and d.max() <= 1
    del d, ds, st22, img22
    # (2) eight ranks, each with its own context, its own slice of the stream and its own buffers
    keep = []
    fe = pp.FrontEnd(fs, stop_rate=16000)
    mk = lambda c, m: sharded.FrontEndShardedDecoder(c, m, fe, None, n_in_total=n0, in_kind=nat.WFX_IN_I16_STEREO, lines_per_minute=120,   # noqa: E731
                                                     raw_loader=synth_device.SliceLoader(c, sp), plan="dist")
    try:
        r = sharded.decode_emulated(np.zeros(1, dtype=np.int16), fs, 8, 120, want=("image", "stream"), make_decoder=mk, free_after=keep)
    finally:
        pass
    assert r["plan"] == 2                                                             # the columns layout: 4 array transposes, not 8
    assert r["sync"]["start_frame"] == s16 and r["sync"]["peaks"] == pk16
    assert np.array_equal(r["digitalized"], st16) and np.array_equal(r["digitalized_blocks"], st16)
    assert np.array_equal(r["image"], img16)
    own = r["own"]
    assert sum(own) == 39690000 and max(own) - min(own) <= 39690000 // 200           # even shares up to a few columns of every row
    sent = sum(e["sent"] for rank_stats in r["wire"] for e in rank_stats)
    assert 1.2e9 < sent < 1.32e9                                                      # (rows layout: 2.54 GB)
    del r
    # (3) round 6, the north star's shape: the same eight ranks on plan 3 -- every rank ingests its eighth of the stream, resamples and
    # Hilbert-transforms its arc by the multipole forms; in front of the ONE gather of the uint8 stream a rank sends well under a megabyte
    keep = []
    mk = lambda c, m: sharded.FrontEndShardedDecoder(c, m, fe, None, n_in_total=n0, in_kind=nat.WFX_IN_I16_STEREO, lines_per_minute=120,   # noqa: E731
                                                     raw_loader=synth_device.SliceLoader(c, sp), plan="fmm")
    r = sharded.decode_emulated(np.zeros(1, dtype=np.int16), fs, 8, 120, want=("image", "stream"), make_decoder=mk, free_after=keep)
    assert r["plan"] == 3
    assert r["sync"]["start_frame"] == s16
    assert np.array_equal(r["digitalized"], r["digitalized_blocks"])
    ds = np.abs(r["digitalized"].astype(np.int16) - st16.astype(np.int16))
    d = np.abs(r["image"].astype(np.int16) - img16.astype(np.int16))
    print(f"plan 3 against the transposing plan, full size: stream differs on {np.count_nonzero(ds)} of {ds.size} (max {ds.max()}), image on {np.count_nonzero(d)} (max {d.max()})")
    assert ds.max() <= 1 and np.count_nonzero(ds) <= 1e-6 * ds.size + 2 and d.max() <= 1
    own = r["own"]
    assert sum(own) == 39690000 and max(own) - min(own) <= 64
    small = [sum(int(e["sent"]) for e in ws if e["name"] != "stream gather") for ws in r["wire"]]
    gather = sum(int(e["sent"]) for ws in r["wire"] for e in ws if e["name"] == "stream gather")
    trees = [sum(int(e["sent"]) for e in ws if e["name"] in ("resampler weights", "resampled halos", "fmm weights", "fmm seams")) for ws in r["wire"]]
    print("plan 3, bytes a rank sends in front of the gather:", small, " of which for the two trees:", trees, " gather:", gather)
    # (the rest is the exact percentile select's: two histogram all-reduces and the candidates of the two bins, every rank's to every rank)
    assert max(trees) <= 256 * 1024 and max(small) <= 4 << 20 and gather <= 39690000
    ctx.close()


def _read_png_gray8(path):
    """Minimal PNG reader (8-bit gray, no interlace): the file must be readable without the library that wrote it."""
    blob = open(path, "rb").read()
    assert blob[:8] == b"\x89PNG\r\n\x1a\n"
    pos, idat, w, h = 8, b"", 0, 0
    while pos < len(blob):
        ln = int.from_bytes(blob[pos:pos + 4], "big")
        kind = blob[pos + 4:pos + 8]
        body = blob[pos + 8:pos + 8 + ln]
        assert zlib.crc32(kind + body) == int.from_bytes(blob[pos + 8 + ln:pos + 12 + ln], "big")
        if kind == b"IHDR":
            w, h = int.from_bytes(body[:4], "big"), int.from_bytes(body[4:8], "big")
            assert body[8:13] == bytes([8, 0, 0, 0, 0])          # 8 bits, gray, deflate, adaptive filters, no interlace
        elif kind == b"IDAT":
            idat += body
        pos += 12 + ln
    raw = np.frombuffer(zlib.decompress(idat), dtype=np.uint8).reshape(h, w + 1)
    out = np.zeros((h, w), dtype=np.uint8)
    for y in range(h):
        f, line = int(raw[y, 0]), raw[y, 1:].astype(np.int32)
        up = out[y - 1].astype(np.int32) if y else np.zeros(w, dtype=np.int32)
        if f == 0:
            out[y] = line
        elif f == 2:
            out[y] = (line + up) & 255
        elif f == 1:
            out[y] = np.cumsum(line) & 255
        else:           # Average / Paeth: serial in x
            cur = np.zeros(w, dtype=np.int32)
            for xk in range(w):
                a = cur[xk - 1] if xk else 0
                b, c = up[xk], (up[xk - 1] if xk else 0)
                if f == 3:
                    pred = (a + b) // 2
                else:
                    pa, pb, pc = abs(b - c), abs(a - c), abs(a + b - 2 * c)
                    pred = a if (pa <= pb and pa <= pc) else (b if pb <= pc else c)
                cur[xk] = (line[xk] + pred) & 255
            out[y] = cur
    return out


@pytest.mark.parametrize("name,lpm", [("mono_noisy_120", 120), ("mono_noisy_240", 240), ("stereo48k_image_240", 240), ("mono_u8_240", 240)])
def test_command_line_wav_to_png(name, lpm, tmp_path):
    """`python wefax.py <wav> <lpm> <out.png>` (wefax.py:411-424), run as a child process from the reference's working
    directory layout: prints file_info, writes an 8-bit gray PNG whose pixels are the reference's image."""
    out = tmp_path / "out.png"
    wav = input_by_name(name)
    r = subprocess.run([sys.executable, os.path.join(REPO, "wefax.py"), wav, str(lpm), str(out)], capture_output=True, text=True, timeout=300, cwd=str(tmp_path))
    assert r.returncode == 0, r.stderr[-2000:]
    assert "filename : " + name + ".wav" in r.stdout and "sample_rate :" in r.stdout        # wefax.py:418-419
    g = load_golden(name)
    img = _read_png_gray8(str(out))
    assert img.shape == g["image"].shape and np.array_equal(img, g["image"])
    try:
        from PIL import Image
        pil = np.asarray(Image.open(str(out)))
        assert pil.dtype == np.uint8 and np.array_equal(pil, g["image"])
    except ImportError:
        pass


def test_png_assembled_on_the_device_and_the_compressed_alternative(tmp_path, monkeypatch):
    """save_output_image: by default the file comes from wfx_decode_save_png_ex (device deflate; WEFAX_PNG_STORED=1: stored deflate
    blocks; Adler-32 and CRC-32 from kernels either way: zlib and the chunk reader below verify both); compress=6 takes the threaded
    zlib encoder on the host.  Same pixels every way, and a context that has moved on to another decode falls back to encoding the
    host copy."""
    from wefax_amd import Demodulator
    g = load_golden("mono_noisy_120")
    d = Demodulator(input_by_name("mono_noisy_120"), lines_per_minute=120, quiet=True)
    d.process()
    a, b, c, e = str(tmp_path / "a.png"), str(tmp_path / "b.png"), str(tmp_path / "c.png"), str(tmp_path / "e.png")
    monkeypatch.setenv("WEFAX_PNG_STORED", "1")
    d.save_output_image(a)
    monkeypatch.delenv("WEFAX_PNG_STORED")
    d.save_output_image(e)
    d.save_output_image(b, compress=6)
    assert os.path.getsize(a) > d.output_array.size and os.path.getsize(b) < os.path.getsize(a)
    assert os.path.getsize(e) < 0.75 * os.path.getsize(a) and os.path.getsize(e) < 1.1 * os.path.getsize(b)
    assert np.array_equal(_read_png_gray8(e), g["image"])
    assert d._ctx.decode_png(deflate=True) == open(e, "rb").read()
    blob = d._ctx.decode_png()
    assert blob == open(a, "rb").read()
    d._ctx.notch_filtfilt(np.zeros(100), [1.0, 0.0, 0.0], [1.0, 0.0, 0.0])      # the context forgets the decode ...
    d.save_output_image(c)                                                       # ... and the host copy is encoded instead
    for path in (a, b, c):
        assert np.array_equal(_read_png_gray8(path), g["image"]), path


@pytest.mark.parametrize("name,lpm", [("mono_clean_120", 120), ("mono_noisy_240", 240), ("mono48k_image_240", 240), ("synthetic", 100)])
def test_device_deflate_png(name, lpm, tmp_path):
    """wfx_decode_png_ex(deflate = 1): the compressed PNG encoded by kernels (Up filter, dynamic-Huffman blocks with distance-1
    runs, one code per image, chunks joined by empty stored blocks) inflates -- with zlib, which checks the Adler-32, through a
    reader that checks the CRC-32s, and with Pillow -- to the decoder's image; the same bytes every time; far smaller than the
    stored form on a clean picture, within 15 % of zlib level 6 (which also has LZ77 matches) on the same filtered bytes otherwise.
    LPM 100 gives a width that is not a multiple of four (the byte-wise loader)."""
    from wefax_amd import Demodulator, synth
    if name == "synthetic":
        wav = str(tmp_path / "s.wav")
        synth.write_wav(wav, 11025, synth.synth_capture(11025.0, noise=0.02, seed=5, lpm=lpm, image_lines=90, phasing_lines=20))
    else:
        wav = input_by_name(name)
    d = Demodulator(wav, lines_per_minute=lpm, quiet=True)
    d.process()
    img = d.output_array
    if name != "synthetic":
        assert np.array_equal(img, load_golden(name)["image"])
    blob, stored = d._ctx.decode_png(deflate=True), d._ctx.decode_png()
    assert blob == d._ctx.decode_png(deflate=True)
    out = str(tmp_path / "d.png")
    open(out, "wb").write(blob)
    assert np.array_equal(_read_png_gray8(out), img)
    raw = np.empty((img.shape[0], img.shape[1] + 1), np.uint8)
    raw[:, 0] = 2
    raw[0, 1:] = img[0]
    np.subtract(img[1:], img[:-1], out=raw[1:, 1:])
    z6 = len(zlib.compress(raw.tobytes(), 6))
    assert len(blob) < (0.2 if name == "mono_clean_120" else 0.8) * len(stored)
    assert len(blob) < (1.6 if name == "mono_clean_120" else 1.15) * z6
    try:
        from PIL import Image
        assert np.array_equal(np.asarray(Image.open(out)), img)
    except ImportError:
        pass
    d.close()


def test_random_captures_against_the_oracle():
    """tools/random_parity.py: random rate / length / LPM / noise / sample format / channel count, the drop-in Demodulator
    against the oracle -- same exception type or same start_frame, identical uint8 stream, progress messages and image
    (profiles/r02_v3/random_parity_summary.json holds a 400-case run)."""
    r = subprocess.run([sys.executable, os.path.join(REPO, "tools", "random_parity.py"), "--cases", "16", "--seed", "3"], capture_output=True, text=True,
                       timeout=600)
    last = json.loads(r.stdout.strip().splitlines()[-1])
    assert r.returncode == 0 and last == {"cases": 16, "failed": 0}, r.stdout[-2000:] + r.stderr[-2000:]


def test_demodulators_share_idle_contexts():
    """One Demodulator per file like the reference, but the context (stream, buffers, plans) outlives it: a closed or collected
    Demodulator hands it to the idle pool and the next one takes it from there; two that are alive at once get two contexts;
    a Demodulator keeps its image until it is closed."""
    from wefax_amd import Demodulator, wefax as wx
    g = load_golden("mono_noisy_120")
    wav = input_by_name("mono_noisy_120")
    wx.release_contexts()
    d1 = Demodulator(wav, lines_per_minute=120, quiet=True)
    d1.process()
    h1 = d1._ctx.h
    d2 = Demodulator(wav, lines_per_minute=120, quiet=True)
    d2.process()
    assert d2._ctx.h != h1                                      # both alive: separate contexts
    assert np.array_equal(d1.output_array, g["image"]) and np.array_equal(d2.output_array, g["image"])
    d1.close()
    d3 = Demodulator(wav, lines_per_minute=120, quiet=True)
    d3.process()
    assert d3._ctx.h == h1                                      # taken from the pool
    assert np.array_equal(d3.output_array, g["image"])
    del d2, d3
    assert sum(len(v) for v in wx._POOL.values()) == 2
    wx.release_contexts()
    assert not wx._POOL


def test_command_line_rejects_what_the_reference_rejects(tmp_path):
    r = subprocess.run([sys.executable, os.path.join(REPO, "wefax.py"), str(tmp_path / "missing.wav"), "120", str(tmp_path / "o.png")],
                       capture_output=True, text=True, timeout=120)
    assert r.returncode != 0 and "INVALID FILE: file at path" in r.stderr                     # wefax.py:25


@pytest.mark.parametrize("dtype", ["uint8", "int32", "float32"])
def test_odd_extension_in_the_captures_own_dtype(dtype):
    """scipy evaluates 2 x[0] - x[k] in the array's dtype before filtering (wefax.py:72): uint8 wraps modulo 256, int32 modulo
    2^32, float32 rounds.  The stage call with the extension handed over equals the oracle's filtfilt on the typed array; the
    call without it (extension formed from the float64 copy) differs near the ends for the wrapping dtypes."""
    from oracle import wefax_oracle as wo
    from wefax_amd import _native as nat
    from wefax_amd import hostparams as hp
    rng = np.random.default_rng(3)
    n = 5000
    if dtype == "uint8":
        x = rng.integers(0, 256, size=n).astype(np.uint8)
        x[0], x[-1] = 250, 3                                  # 2 x[0] - x[k] leaves [0, 255] at both ends
    elif dtype == "int32":
        x = rng.integers(-2**31, 2**31 - 1, size=n).astype(np.int32)
    else:
        x = (rng.standard_normal(n) * 0.3).astype(np.float32)
    b, a = wo.iirnotch(2600, 1, 11025)
    want = wo.filtfilt_biquad(b, a, x)
    ctx = nat.Context(0)
    got = ctx.notch_filtfilt(x.astype(np.float64), b, a, ext=hp.odd_extension(x))
    scale = np.max(np.abs(want))
    assert np.max(np.abs(got - want)) <= 1e-12 * scale
    plain = ctx.notch_filtfilt(x.astype(np.float64), b, a)
    assert np.max(np.abs(plain[40:-40] - want[40:-40])) <= 1e-12 * scale      # the interior never depended on it
    if dtype != "float32":
        assert np.max(np.abs(plain[:20] - want[:20])) > 1e-3 * scale
    ctx.close()


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["mono_noisy_240", "stereo_overflow_120", "stereo48k_image_240", "mono48k_noisy_120"])
def test_pipelined_file_upload_equals_read_then_upload(name, monkeypatch):
    """16-bit PCM files go from the page cache to the device as one pipeline (wfx_decode_upload_fd: slices read by a few threads,
    each on its way by DMA when it is complete); WEFAX_UPLOAD_PIPELINE=0 reads the whole file first (hostparams.read_wav) and uploads
    it then.  Same bytes on the device: same stream, same image, same messages."""
    from conftest import golden_cases
    from wefax_amd import Demodulator
    case = next(c for c in golden_cases() if c["name"] == name)
    res = []
    for pipe in ("1", "0"):
        monkeypatch.setenv("WEFAX_UPLOAD_PIPELINE", pipe)
        d = Demodulator(input_by_name(name), lines_per_minute=case["lpm"], quiet=True, tcp_stream=True)
        try:
            d.process()
            exc = None
        except (ValueError, IndexError) as e:
            exc = [type(e).__name__, str(e)]
        res.append((exc, d.digitalized_data.copy(), d.audio_data.copy(), None if exc else d.output_array.copy(), list(d.websocket_stack)))
        d.close()
    a, b = res
    assert a[0] == b[0] == case["exception"]
    assert np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2]) and a[4] == b[4]
    assert (a[3] is None and b[3] is None) or np.array_equal(a[3], b[3])
    assert np.array_equal(a[1], load_golden(name)["digitalized"])
