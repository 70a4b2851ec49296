"""CPU-side tests: host arithmetic of the product, the C ABI surface, the drop-in
API's argument handling.  No compute call reaches the GPU here."""
import ctypes
import os
import re
import subprocess

import numpy as np
import pytest

from conftest import GOLDEN, REPO, golden_cases, input_path
from oracle import wefax_oracle as wo
from wefax_amd import _native as nat
from wefax_amd import hostparams as hp
from wefax_amd import synth


def test_library_exports_every_declared_symbol():
    hdr = open(os.path.join(REPO, "include", "wefax_hip.h")).read()
    declared = set(re.findall(r"\b(wfx_[a-z0-9_]+)\s*\(", hdr))
    assert declared == set(nat.SYMBOLS)
    lib = nat.load()
    for name in sorted(declared):
        assert hasattr(lib, name), name
    assert lib.wfx_version().startswith(b"wefax_hip")
    assert lib.wfx_profile_kernel_count() > 10
    assert lib.wfx_profile_kernel_name(3) == b"fft_pass_fwd"


def test_ctypes_structs_match_the_c_header(tmp_path):
    src = tmp_path / "abi.c"
    src.write_text(r'''
#include <stdio.h>
#include <stddef.h>
#include "wefax_hip.h"
int main(void) {
  printf("%zu %zu %zu %zu ", sizeof(wfx_wire_entry), sizeof(wfx_wire_time), offsetof(wfx_wire_time, wait_us), offsetof(wfx_wire_time, timed));
  printf("%zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu\n", sizeof(wfx_decode_params), offsetof(wfx_decode_params, notch_b),
         offsetof(wfx_decode_params, rank_lo), offsetof(wfx_decode_params, mindistance),
         offsetof(wfx_decode_params, width), sizeof(wfx_decode_info), offsetof(wfx_decode_info, peak_pos),
         offsetof(wfx_decode_params, ext_left), sizeof(wfx_shard_layout), offsetof(wfx_shard_layout, in_lo),
         sizeof(wfx_synth_params), offsetof(wfx_synth_params, seed), offsetof(wfx_synth_params, iq));
  return 0; }''')
    exe = tmp_path / "abi"
    subprocess.run(["gcc", "-I", os.path.join(REPO, "include"), str(src), "-o", str(exe)], check=True)
    got = [int(v) for v in subprocess.run([str(exe)], capture_output=True, text=True, check=True).stdout.split()]
    P, I, L, S = nat.DecodeParams, nat.DecodeInfo, nat.ShardLayout, nat.SynthParams
    W, T = nat.WireEntry, nat.WireTime
    assert got[:4] == [ctypes.sizeof(W), ctypes.sizeof(T), T.wait_us.offset, T.timed.offset]
    got = got[4:]
    assert got == [ctypes.sizeof(P), P.notch_b.offset, P.rank_lo.offset, P.mindistance.offset, P.width.offset,
                   ctypes.sizeof(I), I.peak_pos.offset, P.ext_left.offset, ctypes.sizeof(L), L.in_lo.offset,
                   ctypes.sizeof(S), S.seed.offset, S.iq.offset]


def test_no_gpu_means_a_loud_failure_not_a_fallback():
    if nat.device_count() > 0:
        pytest.skip("a GPU is present")
    with pytest.raises(nat.NativeError, match="no HIP device"):
        nat.Context(0)
    from wefax_amd import Demodulator
    d = Demodulator(os.path.join(GOLDEN, "inputs", "ref_image.wav"), quiet=True)
    with pytest.raises(nat.NativeError):
        d.process()


def test_product_never_imports_the_oracle():
    pkg = os.path.join(REPO, "wefax_amd")
    for root, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                text = open(os.path.join(root, f)).read()
                assert "oracle" not in text.replace("no CPU fallback", ""), os.path.join(root, f)
    assert "oracle" not in open(os.path.join(REPO, "wefax.py")).read()


def test_bench_reaches_the_oracle_from_bench_py_alone():
    """bench.py is split into bench.py (command line, headline, the CPU baseline) and benchlib/ (round 6).  The oracle is the checker and
    the CPU baseline of bench.py's legs only: no module of benchlib imports it -- they ask bench.py's hooks -- and the names tests and tools
    use (`bench._LineGuard`, `bench.iq_recipe`, ...) are still bench's."""
    import re
    import subprocess
    import sys
    lib = os.path.join(REPO, "benchlib")
    for f in sorted(os.listdir(lib)):
        if f.endswith(".py"):
            text = open(os.path.join(lib, f)).read()
            assert not re.search(r"^\s*(from|import)\s+oracle", text, re.M), f
    r = subprocess.run([sys.executable, "-c", "import bench, benchlib; assert benchlib.ORACLE is bench._oracle and benchlib.CPU_BASELINE is bench.cpu_baseline; "
                        "print(all(hasattr(bench, n) for n in ('_LineGuard', 'iq_recipe', 'Ranks', 'GpuState', 'roofline_of', 'bench_c2', 'bench_iq', 'main')))"],
                       capture_output=True, text=True, cwd=REPO, timeout=120)
    assert r.returncode == 0 and r.stdout.strip() == "True", r.stderr


def test_constructor_contract():
    from wefax_amd import Demodulator
    with pytest.raises(Exception) as e:
        Demodulator("/nonexistent/x.wav")
    assert str(e.value) == "INVALID FILE: file at path: /nonexistent/x.wav does not exist"   # wefax.py:25
    with pytest.raises(Exception) as e:
        Demodulator(os.path.join(REPO, "README.md"))
    assert str(e.value) == "INVALID FILETYPE: only .wav files are supported at this moment"  # wefax.py:28
    d = Demodulator(os.path.join(GOLDEN, "inputs", "stereo_overflow_120.wav"), lines_per_minute=120, quiet=True)
    assert d.websocket_stack == [] and d.time_for_one_frame == 0.5
    d.update_lines_per_minute(240)
    assert d.lines_per_minute == 240 and d.time_for_one_frame == 0.25
    fi = d.file_info()
    assert fi == {"filename": "stereo_overflow_120.wav", "channels": 2, "sample_rate": 11025, "length": 13.0}


@pytest.mark.parametrize("case", golden_cases(), ids=[c["name"] for c in golden_cases()])
def test_wav_reader_matches_oracle_reader(case):
    p = input_path(case)
    sr0, d0 = wo.read_wav(p)
    sr1, d1 = hp.read_wav(p)
    assert sr0 == sr1 and d0.dtype == d1.dtype and np.array_equal(d0, d1)


@pytest.mark.parametrize("case", golden_cases(), ids=[c["name"] for c in golden_cases()])
def test_file_info_from_the_headers_equals_the_references(case):
    """file_info (wefax.py:342-346) is answered from the wav's headers alone; the manifest holds what the reference printed."""
    from wefax_amd import Demodulator
    d = Demodulator(input_path(case), lines_per_minute=case.get("lpm", 120), quiet=True)
    fi, ref = d.file_info(), case["file_info"]
    assert fi["filename"] == ref["filename"] and fi["channels"] == ref["channels"] and fi["sample_rate"] == ref["sample_rate"]
    assert fi["length"] == ref["length"]
    sr, frames, ch = hp.wav_info(input_path(case))
    sr1, data = hp.read_wav(input_path(case))
    assert (sr, frames, ch) == (sr1, data.shape[0], 1 if data.ndim == 1 else data.shape[1])


def test_wav_reader_threads_allocator_and_truncated_files(tmp_path):
    """Files beyond one read slice are copied out of the page cache by several threads, into memory the caller provides when it
    gives an allocator (the decoder: its context's page-locked staging buffer); chunks in front of `data` are skipped; a data
    chunk whose header promises more than the file holds yields the whole frames that are there (scipy: the same, with a
    warning); 24-bit PCM is widened as scipy does."""
    from scipy.io import wavfile
    rng = np.random.default_rng(4)
    x = rng.integers(-32768, 32767, size=(1_400_001, 2), dtype=np.int16)            # 5.6 MB: three slices
    p = str(tmp_path / "big.wav")
    wavfile.write(p, 48000, x)
    blob = open(p, "rb").read()
    q = str(tmp_path / "list.wav")
    pos = blob.index(b"data")
    with open(q, "wb") as fh:                                                        # a LIST chunk of odd size (padded) before the samples
        extra = b"LIST" + (5).to_bytes(4, "little") + b"abcde\0"
        body = blob[12:pos] + extra + blob[pos:]
        fh.write(b"RIFF" + (4 + len(body)).to_bytes(4, "little") + b"WAVE" + body)
    asked = []

    def alloc(nbytes):
        asked.append(nbytes)
        return np.zeros(nbytes + 100, np.uint8)

    for path in (p, q):
        sr, d = hp.read_wav(path, alloc=alloc)
        assert sr == 48000 and d.dtype == np.int16 and d.shape == x.shape and np.array_equal(d, x)
        assert d.flags.writeable
    assert asked == [x.nbytes, x.nbytes]
    t = str(tmp_path / "cut.wav")
    open(t, "wb").write(blob[:len(blob) - 4 * 1000 - 3])                             # 1000 frames and 3 bytes short
    sr, d = hp.read_wav(t)
    assert d.shape == (x.shape[0] - 1001, 2) and np.array_equal(d, x[:-1001])
    y = rng.integers(-2**23, 2**23 - 1, size=50_000, dtype=np.int32)
    raw = np.ascontiguousarray((y.astype("<i4").view(np.uint8).reshape(-1, 4))[:, :3]).tobytes()
    hdr = b"fmt " + (16).to_bytes(4, "little") + (1).to_bytes(2, "little") + (1).to_bytes(2, "little") + (11025).to_bytes(4, "little") + \
        (33075).to_bytes(4, "little") + (3).to_bytes(2, "little") + (24).to_bytes(2, "little")
    body = hdr + b"data" + len(raw).to_bytes(4, "little") + raw
    u = str(tmp_path / "p24.wav")
    open(u, "wb").write(b"RIFF" + (4 + len(body)).to_bytes(4, "little") + b"WAVE" + body)
    sr, d = hp.read_wav(u)
    sr0, d0 = wavfile.read(u)
    assert sr == sr0 == 11025 and d.dtype == d0.dtype and np.array_equal(d, d0)


def test_iirnotch_matches_oracle():
    for fs in (11025, 8000, 48000):
        b0, a0 = wo.iirnotch(2600, 1, fs)
        b1, a1 = hp.iirnotch(2600, 1, fs)
        assert np.array_equal(b0, b1) and np.array_equal(a0, a1)
    with pytest.raises(ValueError):
        hp.iirnotch(6000, 1, 11025)
    assert hp.load_notch_settings("/nonexistent.json") == (2600, 1)
    # a file that exists but lacks the keys raises, as the reference's Config()[...] lookup does (wefax.py:63-64)
    with pytest.raises(KeyError):
        hp.load_notch_settings(os.path.join(REPO, "tests", "golden", "manifest.json"))


def test_percentile_plan_reproduces_numpy():
    rng = np.random.default_rng(0)
    for n in (1, 2, 3, 10, 199, 200, 201, 11025, 250007, 7166250):
        v = np.sort(rng.standard_normal(min(n, 4000)))
        for q in (0.5, 99.5, 0.0, 100.0, 50.0):
            lo, hi, g = hp.percentile_plan(n, q)
            assert 0 <= lo <= hi <= n - 1 and hi - lo <= 1
            if n <= 4000:
                a, b = v[lo], v[hi]
                diff = b - a
                r = a + diff * g
                if g >= 0.5:
                    r = b - diff * (1 - g)
                assert r == np.percentile(v, q)


def test_sync_constants_table():
    # (n1, n0, mindistance, width) per lines-per-minute
    table = {60: (55, 11, 8820, 11025), 90: (36, 7, 5880, 7350), 100: (33, 6, 5292, 6615),
             120: (27, 5, 4410, 5512), 180: (18, 3, 2940, 3675), 240: (13, 2, 2205, 2756)}
    for lpm, (n1, n0, mind, w) in table.items():
        T = 1 / (lpm / 60)
        assert hp.sync_constants(11025, T) == (n1, n0, mind) == wo.sync_constants(11025, T)
        assert int(T * 11025) == w


def test_synthetic_workloads_have_the_baseline_sizes():
    f = synth.wefax_frequency_track(11025.0)
    assert f.shape[0] == 7166250
    assert set(np.unique(f[:55125])) == {1500.0, 2300.0}        # start tone toggles black/white
    x = synth.synth_capture(11025.0, phasing_lines=2, image_lines=2, start_tone_s=0.1, stop_tone_s=0.1,
                            black_tail_s=0.1)
    assert x.dtype == np.int16 and abs(int(np.abs(x).max()) - 16384) < 3
    iq = synth.synth_capture(48000.0, iq=True, phasing_lines=1, image_lines=1, start_tone_s=0.1,
                             stop_tone_s=0.1, black_tail_s=0.1)
    assert iq.ndim == 2 and iq.shape[1] == 2
    x5, lpm = synth.config_c5_member(1, noise=0.0)      # 240 LPM, IOC576: same duration as C2
    assert lpm == 240 and x5.shape[0] == 7166250
    x5, lpm = synth.config_c5_member(2, noise=0.0)      # 120 LPM, IOC288: 600 image lines
    assert lpm == 120 and x5.shape[0] == 7166250 - 600 * 5512.5


def test_decode_job_parameter_block(monkeypatch):
    """DecodeJob fills the ABI struct with the reference's scalar arithmetic (no GPU needed)."""
    from wefax_amd.wefax import DecodeJob

    class FakeCtx:
        def decode_upload(self, data, p):
            self.data, self.p = data, p

    fc = FakeCtx()
    x = np.zeros(48000 * 3 + 17, dtype=np.int16)
    job = DecodeJob(fc, x, 48000, 240)
    assert job.resampled and job.n == int(11025 * (x.shape[0] / 48000)) and job.width == 2756
    p = fc.p
    assert (p.in_kind, p.n0, p.n, p.resample) == (nat.WFX_IN_I16_MONO, x.shape[0], job.n, 1)
    assert (p.n1, p.n0_gap, p.mindistance, p.frame_samples) == (13, 2, 2205, 2756.25)
    b, a = wo.iirnotch(2600, 1, 11025)
    assert list(p.notch_b) == list(b) and list(p.notch_a) == list(a)
    st = np.zeros((1000, 2), dtype=np.int16)
    assert DecodeJob(fc, st, 11025, 120).params.in_kind == nat.WFX_IN_I16_STEREO
    u8 = np.full(1000, 128, dtype=np.uint8)
    assert DecodeJob(fc, u8, 11025, 120).params.in_kind == nat.WFX_IN_F64_MONO and fc.data.dtype == np.float64
    with pytest.raises(ValueError, match="greater than padlen"):
        DecodeJob(fc, np.zeros(9, dtype=np.int16), 11025, 120)


def test_fast_png_writer_round_trips_through_pillow(tmp_path):
    Image = pytest.importorskip("PIL.Image")
    from wefax_amd import pngio
    rng = np.random.default_rng(3)
    for shape in ((1, 1), (5, 3), (37, 5512), (440, 2756)):
        img = rng.integers(0, 256, size=shape, dtype=np.uint8)
        p = str(tmp_path / f"t{shape[0]}.png")
        pngio.write_png_gray8(p, img, threads=3)
        back = Image.open(p)
        assert back.mode == "L" and back.size == (shape[1], shape[0])
        assert np.array_equal(np.asarray(back), img)
    smooth = np.repeat((np.arange(300 * 64).reshape(300, 64) // 7 % 256).astype(np.uint8), 4, axis=0)
    assert np.array_equal(np.asarray(Image.open(__import__("io").BytesIO(pngio.encode_png_gray8(smooth, threads=1)))), smooth)
    with pytest.raises(ValueError):
        pngio.encode_png_gray8(np.zeros((0, 4), np.uint8))


def test_detector_settings_come_from_the_config_file(tmp_path, monkeypatch):
    """data_packet.py:21-42 reads tones_settings / sync_pulse_settings from config/config.json in the working directory; so does the
    packet class here (shipped values when the file is absent, an error when it is malformed)."""
    import json
    from wefax_amd import detect
    assert hp.load_detector_settings(str(tmp_path / "none.json")) == (None, None)
    cfg = {"notch_filter_settings": {"notch_filter_frequency": 2500, "notch_filter_quality_factor": 2},
           "tones_settings": {"start_tone_peaks_minimum_distance": 111, "stop_tone_peaks_minimum_distance": 222, "peaks_minimum_distance": 250,
                              "peaks_minimum_height": 0.07, "peaks_minimum_prominence": 0.3, "peaks_minimum_frequency": 700,
                              "peaks_maximum_frequency": 3300, "peaks_minimum_amount": 3, "peaks_maximum_amount": 7},
           "sync_pulse_settings": {"peaks_minimum_distance": 750, "peaks_minimum_height": 0.6, "peaks_minimum_prominence": 0.25,
                                   "peaks_minimum_frequency": 1300, "peaks_maximum_frequency": 1700}}
    (tmp_path / "config").mkdir()
    (tmp_path / "config" / "config.json").write_text(json.dumps(cfg))
    monkeypatch.chdir(tmp_path)
    tones, pulse = hp.load_detector_settings()
    assert tones == dict(start_distance=111, stop_distance=222, height=0.07, prominence=0.3, fmin=700, fmax=3300, amount_min=3, amount_max=7)
    assert pulse == dict(height=0.6, prominence=0.25, fmin=1300, fmax=1700)
    assert hp.load_notch_settings() == (2500, 2)
    assert set(tones) == set(detect.TONES) and set(pulse) == set(detect.SYNC_PULSE)
    (tmp_path / "config" / "config.json").write_text(json.dumps({"tones_settings": {}}))
    with pytest.raises(KeyError):
        hp.load_detector_settings()


def test_bench_guard_prints_the_headline_when_the_sharded_part_hangs(tmp_path):
    """bench.py at N > 1: the headline is measured before the one part of the line that runs data-path collectives; if that part
    does not come back (a peer died, a collective hangs) rank 0 prints the line it has with the reason in `c4_strong` and every rank
    leaves.  No GPU: the guard alone, around a sleep."""
    import json
    import subprocess
    import sys
    code = (
        "import sys, time, types\n"
        f"sys.path.insert(0, {REPO!r})\n"
        "import bench\n"
        "rk = types.SimpleNamespace(rank=int(sys.argv[1]), world=2)\n"
        "line = {'metric': 'm', 'value': 1.5}\n"
        "g = bench._LineGuard(line, rk, 0.3)\n"
        "time.sleep(30)\n")
    for rank, want_rc in ((0, 0), (1, 3)):
        r = subprocess.run([sys.executable, "-c", code, str(rank)], capture_output=True, text=True, timeout=60)
        assert r.returncode == want_rc, r.stderr
        if rank == 0:
            out = json.loads(r.stdout.strip().splitlines()[-1])
            assert out["value"] == 1.5 and "no result" in out["c4_strong"]["error"]
        else:
            assert r.stdout.strip() == ""
    # the part came back in time: the main thread claims the line, a late timer prints nothing
    code2 = code.replace("time.sleep(30)", "assert g.claim(); print('LINE'); g.printed_exit_only(); time.sleep(0.6)")
    r = subprocess.run([sys.executable, "-c", code2, "0"], capture_output=True, text=True, timeout=60)
    assert r.returncode == 0 and r.stdout.strip() == "LINE", (r.stdout, r.stderr)


def test_golden_inputs_regenerate_from_nothing_and_a_mismatch_costs_one_case(tmp_path):
    """A directory with the manifest and NO wav: every recipe input comes back with the manifest's SHA-256 (what a fresh checkout
    does for the inputs that are not kept in the repository); a recipe that does not reproduce its hash -- another NumPy build's
    sin / cos -- is reported for THAT case and leaves the others alone (round-4 verdict: it used to turn the whole session red)."""
    import json
    import shutil
    import sys
    sys.path.insert(0, GOLDEN)
    try:
        import recipes
    finally:
        sys.path.remove(GOLDEN)
    gold = tmp_path / "golden"
    (gold / "inputs").mkdir(parents=True)
    man = json.load(open(os.path.join(GOLDEN, "manifest.json")))
    small = [c for c in man["cases"] if "recipe" in c and c["recipe"] in ("mono8k_noisy_120", "mono_clean_120", "mono_u8_240", "stereo_overflow_120")]
    assert len(small) == 4
    bad_name = small[1]["name"]
    small[1] = dict(small[1], input_sha256="0" * 64)
    json.dump({"cases": small}, open(gold / "manifest.json", "w"))
    recipes.FAILED.pop(bad_name, None)
    try:
        bad = recipes.ensure_all(str(gold), strict=False)
        assert list(bad) == [bad_name] and "only this case is affected" in bad[bad_name]
        for c in small:
            p = gold / c["input"]
            if c["name"] == bad_name:
                assert not p.exists()
                with pytest.raises(recipes.GoldenInputMismatch):
                    recipes.ensure_input(str(gold), c)
            else:       # identical to the copy kept in the repository
                assert recipes.file_sha256(str(p)) == c["input_sha256"] == recipes.file_sha256(os.path.join(GOLDEN, c["input"]))
        with pytest.raises(recipes.GoldenInputMismatch):
            recipes.ensure_all(str(gold), strict=True)
    finally:
        recipes.FAILED.pop(bad_name, None)
    shutil.rmtree(gold)


def test_every_recipe_still_hashes_to_the_manifest(tmp_path):
    """All 22 recipes, the 12 MB IQ clip included, in a directory of their own (the kept copies are not consulted)."""
    import json
    import sys
    sys.path.insert(0, GOLDEN)
    try:
        import recipes
    finally:
        sys.path.remove(GOLDEN)
    gold = tmp_path / "golden"
    (gold / "inputs").mkdir(parents=True)
    man = json.load(open(os.path.join(GOLDEN, "manifest.json")))
    cases = [c for c in man["cases"] if "recipe" in c]
    assert len(cases) == 22
    json.dump({"cases": cases}, open(gold / "manifest.json", "w"))
    assert recipes.ensure_all(str(gold), strict=False) == {}
    assert sorted(os.listdir(gold / "inputs")) == sorted(os.path.basename(c["input"]) for c in cases)


def test_placed_allocation_keeps_the_best_candidate_and_frees_the_rest(monkeypatch):
    """Context.dev_malloc_placed (host logic only, a stand-in for the context): candidates are held side by side while they are timed,
    the first one that reaches the bar ends the search, the best is kept, every other one is freed; one try or a small buffer: no timing."""
    from wefax_amd import _native as nat

    class Fake:
        def __init__(self, rates):
            self.rates, self.live, self.freed, self.n = list(rates), [], [], 0

        def dev_malloc(self, nbytes):
            self.n += 1
            self.live.append(self.n)
            return self.n

        def dev_free(self, p):
            self.live.remove(p)
            self.freed.append(p)

        def d_stream_rate(self, p, nbytes):
            assert len(self.live) == p                      # (earlier candidates are still allocated: the next one lies elsewhere)
            return self.rates[p - 1]

    placed = nat.Context.dev_malloc_placed
    f = Fake([5200.0, 5900.0, 5600.0, 7000.0])
    p, rates = placed(f, 2 << 30, tries=3)
    assert p == 2 and rates == [5200.0, 5900.0, 5600.0] and f.live == [2] and sorted(f.freed) == [1, 3]
    f = Fake([5200.0, 6100.0, 7000.0])
    p, rates = placed(f, 2 << 30, tries=4)                  # the second candidate reaches the bar (6000 GB/s): the search ends there
    assert p == 2 and rates == [5200.0, 6100.0] and f.live == [2] and f.freed == [1]
    f = Fake([1.0])
    assert placed(f, 2 << 30, tries=1) == (1, []) and placed(f, 1 << 20, tries=4) == (2, [])      # one try / a small buffer: not timed
    monkeypatch.setenv("WFX_PLACE_TRIES", "2")
    f = Fake([10.0, 20.0, 30.0])
    p, rates = placed(f, 2 << 30)
    assert p == 2 and rates == [10.0, 20.0]
    f = Fake([3.0, 2.0])
    p, rates = placed(f, 1 << 20, tries=2, probe=lambda q: {1: 3.0, 2: 2.0}[q])      # with a probe of the caller's, any size is timed
    assert p == 1 and rates == [3.0, 2.0] and f.live == [1]


def test_the_shipped_library_reads_no_lab_switch_from_the_environment():
    """Round-5 verdict: diagnostics that change results (WFX_INGEST_DBG: "results are WRONG unless 0") were read by the production library on
    every launch.  They exist in variant builds only now (-DWFX_LAB: wfx_internal.h WFX_LAB_ENV); what the shipped sources still read with
    getenv() is this list -- deployment settings and test hooks that select another, equivalent code path."""
    import glob
    import re
    allowed = {"WFX_LINK_GBS", "WFX_LINK_LAT_US", "WFX_SHARD_CHUNKS", "WFX_SHARD_ROWS", "WFX_PNG_SLICE_KB", "WFX_PNG_THREADS", "WFX_DEBUG",
               "WFX_INGEST_NI", "WFX_INGEST_TILE", "WFX_NO_CZT", "WFX_NO_I16_RESAMPLE", "WFX_MR_NT", "WFX_PICK_SEG", "WFX_COMM_ASYNC"}
    root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "wefax_amd", "csrc")
    seen = set()
    for f in glob.glob(os.path.join(root, "*.hip")) + glob.glob(os.path.join(root, "*.h")):
        for line in open(f):
            if "#define WFX_LAB_ENV(name) getenv(name)" in line:
                continue
            seen |= set(re.findall(r'(?<!LAB_ENV\()\bgetenv\("([A-Z0-9_]+)"\)', line))
    assert seen <= allowed, sorted(seen - allowed)
    lab = set()
    for f in glob.glob(os.path.join(root, "*.hip")):
        lab |= set(re.findall(r'WFX_LAB_ENV\("([A-Z0-9_]+)"\)', open(f).read()))
    assert "WFX_INGEST_DBG" in lab and not (lab & allowed)
