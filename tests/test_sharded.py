"""ONE capture over several GPUs (wefax_amd/sharded.py, csrc/wfx_shard.hip, wfx_dist.hip, wfx_comm.hip).

CPU (no GPU needed): how a capture is cut for N ranks, the host-only consistency check of every rank's exchange lists
(also at the full sizes of BASELINE configs[2] and [3], which no test GPU run covers), the error behaviour for captures
that cannot be sharded, and the RCCL unique-id bootstrap between two real processes over a loopback socket.

GPU (-m gpu): the sharded exact decode with every rank emulated on one GPU (in-process communicator: a collective
completes when the last rank has posted its part; a message whose two ends disagree on the size is an error) against
the oracle and the single-GPU exact path, for world sizes 1, 2, 3 and 8; the RCCL transport with the one rank a
one-GPU box has; the percentile select's overflow path; the device test-signal kernels against the NumPy generator."""
import multiprocessing as mp
import os
import socket
import tempfile

import numpy as np
import pytest

from wefax_amd import _native as nat
from wefax_amd import sharded, synth
from wefax_amd.wefax import build_params

KW130 = dict(start_tone_s=5.0, phasing_lines=20, image_lines=220, stop_tone_s=2.0, black_tail_s=3.0)      # 130 s: N = 1 433 250
KW30 = dict(start_tone_s=2.0, phasing_lines=20, image_lines=30, stop_tone_s=1.0, black_tail_s=2.0)        # 30 s


# ---------------------------------------------------------------------------------------------------------------
# CPU
# ---------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("plan", ["dist", "rows"])
@pytest.mark.parametrize("n0,sr,kind", [(1433250, 11025, 0), (7166250, 11025, 0), (1440000, 48000, 0), (1440000, 48000, 1),
                                       (172800000, 48000, 0), (79380000, 22050, 2), (286650, 11025, 0), (330750, 11025, 0), (9922500, 22050, 2),
                                       (52920000, 14700, 2), (588000, 14700, 2), (57600000, 16000, 2), (640000, 16000, 2)])
def test_layout_tiles_the_capture_and_the_exchange_lists_are_consistent(n0, sr, kind, plan):
    """Both layouts of the distributed form ("dist": the columns layout of round 4 wherever it applies; "rows": rounds 2-3)."""
    p, meta = build_params(kind, n0, sr, 0.5, shard_plan=sharded.plan_code(plan))
    n = meta["n"]
    for world in (1, 2, 3, 4, 8):
        lays = [nat.shard_layout(p, world, r) for r in range(world)]
        r1 = lays[0].first_radix[0] * lays[0].first_radix[1]
        assert r1 > 0 and all(lay.plan_forced == 1 for lay in lays)
        assert sum(lay.own_samples for lay in lays) == n
        if lays[0].plan == 2 and world > 1:
            # COLUMNS layout: every rank holds the same columns of each of the r1 rows -- r1 segments, one row apart
            assert plan == "dist" and all(lay.plan == 2 and lay.nseg == r1 for lay in lays)
            stride = lays[0].own_seg_stride
            assert r1 * stride == n and all(lay.own_seg_stride == stride for lay in lays)
            assert lays[0].own_lo == 0 and lays[-1].own_lo + lays[-1].own_seg_len == stride                 # the segments of the ranks tile a row
            assert all(lays[i].own_lo + lays[i].own_seg_len == lays[i + 1].own_lo for i in range(world - 1))
            assert all(lay.own_lo % 8 == 0 and lay.own_seg_len % 2 == 0 and lay.own_seg_len >= 256 for lay in lays)
            lens = [lay.own_seg_len for lay in lays]
            assert max(lens) - min(lens) <= 16                                                               # balanced to a few columns
            assert all(lay.own_hi == lay.own_lo + (r1 - 1) * stride + lay.own_seg_len for lay in lays)
            if meta["resampled"]:   # input: the rank's columns of the input's own arrangement, no halo (the resampler is global)
                istr = lays[0].in_seg_stride
                assert r1 * istr == n0 and all(lay.in_halo == 0 and lay.in_seg_stride == istr for lay in lays)
                assert lays[0].in_lo == 0 and lays[-1].in_lo + lays[-1].in_seg_len == istr
                assert all(lays[i].in_lo + lays[i].in_seg_len == lays[i + 1].in_lo for i in range(world - 1))
            else:                   # the notch needs 24 + 2 samples beyond every segment: 32 are handed over
                assert all(lay.in_halo == 32 and lay.in_lo == lay.own_lo and lay.in_seg_len == lay.own_seg_len and lay.in_seg_stride == stride for lay in lays)
            assert all(lay.in_frames == r1 * (lay.in_seg_len + 2 * lay.in_halo) for lay in lays)
            if n <= 2000000:        # (the index arrays of the small cases: every sample owned exactly once)
                owned = np.concatenate([lay.own_index() for lay in lays])
                assert owned.shape[0] == n and np.array_equal(np.sort(owned), np.arange(n))
        else:
            assert all(lay.nseg == 1 for lay in lays) and (world == 1 or all(lay.plan == 1 for lay in lays))
            assert lays[0].own_lo == 0 and lays[-1].own_hi == n
            assert all(lays[i].own_hi == lays[i + 1].own_lo for i in range(world - 1))
            assert all(lay.own_lo % 2 == 0 and lay.own_hi % 2 == 0 for lay in lays)             # packed pairs stay together
            sizes = [lay.own_hi - lay.own_lo for lay in lays]
            assert max(sizes) - min(sizes) <= n // r1                                             # balanced to one row of the first radix
            if meta["resampled"]:
                assert lays[0].in_lo == 0 and lays[-1].in_hi == n0
                assert all(lays[i].in_hi == lays[i + 1].in_lo for i in range(world - 1))
                # the input rows are the same share of the capture as the output rows
                assert all(abs((lay.in_hi - lay.in_lo) / n0 - (lay.own_hi - lay.own_lo) / n) < 1e-12 for lay in lays)
            else:   # the notch needs 24 + 2 samples beyond the own range: the slice carries 32, clipped at the capture's ends
                assert all(lay.in_lo == max(0, lay.own_lo - 32) and lay.in_hi == min(n0, lay.own_hi + 32) for lay in lays)
        nat.shard_dry_run(p, world)          # raises if two ends of a message disagree, a receive leaves its buffer, ...


def _wire_total(p, world, only=None):
    return sum(e["bytes"] for e in nat.shard_wire_plan(p, world) if only is None or only(e["name"]))


def test_columns_layout_halves_the_bytes_on_the_wire():
    """BASELINE configs[3] behind the front end (57.6 M samples at 16 kHz -> 39.69 M at 11 025 Hz) on 8 ranks: the rows layout of
    rounds 2-3 moves eight array transposes (2 x 461 + 6 x 317.5 MB, 7/8 of each over the links), the columns layout four
    (461 + 3 x 317.5 MB) plus two halo exchanges of a few hundred KB -- asserted from the plan's own exchange lists (no GPU)."""
    n0, n = 57600000, 39690000
    rows, _ = build_params(2, n0, 16000, 0.5, n_out=n, shard_plan=sharded.plan_code("rows"))
    cols, _ = build_params(2, n0, 16000, 0.5, n_out=n, shard_plan=sharded.plan_code("dist"))
    a_f, a_h = 16 * (n0 // 2), 16 * (n // 2)                     # bytes of the resampler's forward array and of the 11 025 Hz arrays
    for world in (2, 4, 8):
        wr, wc = nat.shard_wire_plan(rows, world), nat.shard_wire_plan(cols, world)
        names_c = [e["name"] for e in wc]
        # (every transpose travels in 4 k1 subsets, each a collective of its own: the exchange of one overlaps the passes of another)
        nsub = names_c.count("hilbert E2")
        assert nsub == 4
        dedup = [n for i, n in enumerate(names_c) if i == 0 or names_c[i - 1] != n]
        assert dedup == ["resample fwd E2", "resample inv E3", "resample inv halo", "hilbert E2", "hilbert E3", "hilbert halo",
                         "select level 0", "select level 1", "select candidates", "stream gather"]
        assert [e["name"] for e in wr][:8] == ["resample fwd E1", "resample fwd E2", "resample inv E3", "resample inv E4", "hilbert E1", "hilbert E2",
                                               "hilbert E3", "hilbert E4"]
        tr = _wire_total(rows, world, lambda nm: " E" in nm)
        tc = _wire_total(cols, world, lambda nm: " E" in nm)
        share = (world - 1) / world
        assert abs(tr - share * (2 * a_f + 6 * a_h)) <= 0.002 * tr          # (column ranges are multiples of four: not exactly 1 / world)
        assert abs(tc - share * (a_f + 3 * a_h)) <= 0.002 * tc
        halo = _wire_total(cols, world, lambda nm: nm.endswith("halo"))
        assert halo <= world * 225 * (16 + 16 + 2 + 2) * 16
        total_r, total_c = _wire_total(rows, world), _wire_total(cols, world)
        assert total_c <= 0.52 * total_r
        # a transpose puts array / world^2 on every directed link, a quarter of it per subset
        link = sum(e["max_link_bytes"] for e in wc if e["name"] == "hilbert E2")
        assert abs(link - a_h / world ** 2) <= 0.2 * a_h / world ** 2         # (the maxima of four subsets, whose k1 sets differ by a unit or two)
    assert _wire_total(cols, 8) <= 1.31e9 and _wire_total(rows, 8) >= 2.5e9


def test_the_number_of_k1_subsets_follows_the_cost_model(monkeypatch):
    """Without WFX_SHARD_CHUNKS the plan takes the subset count its model prices cheapest among 1 .. 4: more subsets hide more of the
    slab passes behind the wire and cost one exchange latency each -- long captures on few ranks take more, short captures or many
    ranks fewer; a cheaper exchange (WFX_LINK_LAT_US) moves the choice up, never down."""
    def subsets(params, world):
        return [e["name"] for e in nat.shard_wire_plan(params, world)].count("hilbert E2")
    monkeypatch.delenv("WFX_SHARD_CHUNKS", raising=False)
    monkeypatch.delenv("WFX_LINK_LAT_US", raising=False)
    c3, _ = build_params(2, 57600000, 16000, 0.5, n_out=39690000, shard_plan=sharded.plan_code("dist"))
    c1, _ = build_params(0, 7166250, 11025, 0.5, shard_plan=sharded.plan_code("dist"))
    at20 = [subsets(c3, w) for w in (2, 4, 8)]
    assert all(1 <= k <= 4 for k in at20) and at20[0] >= at20[1] >= at20[2] and at20[0] >= 3
    assert subsets(c1, 4) == 1                                    # 0.3 ms of work: nothing to hide an exchange behind
    assert "k1 subset" in nat.shard_layout(c3, 8, 0).plan_reason.decode()
    monkeypatch.setenv("WFX_LINK_LAT_US", "5")
    at5 = [subsets(c3, w) for w in (2, 4, 8)]
    assert all(a >= b for a, b in zip(at5, at20)) and at5[0] == 4
    for w in (2, 4, 8):
        nat.shard_dry_run(c3, w)
    monkeypatch.setenv("WFX_SHARD_CHUNKS", "4")
    assert [subsets(c3, w) for w in (2, 4, 8)] == [4, 4, 4]


def test_the_cost_model_declines_a_distributed_plan_that_would_lose(monkeypatch):
    """shard_plan 0 (the default): with 50 GB/s links two ranks would spend longer exchanging configs[3]'s arrays than one GPU
    needs for the whole decode -- the single plan is taken and says why; eight ranks get a sharded plan (since round 6 the chunk-local
    one); links that are fast enough flip the choice; the caller can force either."""
    n0, n = 57600000, 39690000
    monkeypatch.delenv("WFX_LINK_GBS", raising=False)
    monkeypatch.delenv("WFX_LINK_LAT_US", raising=False)
    monkeypatch.delenv("WFX_SHARD_CHUNKS", raising=False)
    p, _ = build_params(2, n0, 16000, 0.5, n_out=n)
    lay2, lay8 = nat.shard_layout(p, 2, 0), nat.shard_layout(p, 8, 0)
    assert lay2.plan == 0 and lay2.plan_forced == 0 and lay2.plan_reason.decode().startswith("cost model")
    assert lay2.model_dist_compute_s + lay2.model_dist_wire_s > lay2.model_single_s > 0
    assert (lay2.own_lo, lay2.own_hi) == (0, n) and nat.shard_layout(p, 2, 1).own_samples == 0
    # (round 6: eight ranks take plan 3 -- both multipole forms, kilobytes on the wire; the transposing plan, when asked for, moves 1.3 GB)
    assert lay8.plan == 3 and lay8.model_dist_compute_s + lay8.model_dist_wire_s < 0.5 * lay8.model_single_s and lay8.model_wire_bytes < 64e6
    dist8, _ = build_params(2, n0, 16000, 0.5, n_out=n, shard_plan=sharded.plan_code("dist"))
    lay8d = nat.shard_layout(dist8, 8, 0)
    assert lay8d.plan == 2 and lay8d.model_dist_compute_s + lay8d.model_dist_wire_s < lay8d.model_single_s
    assert 1.2e9 < lay8d.model_wire_bytes < 1.35e9
    monkeypatch.setenv("WFX_LINK_GBS", "400")
    assert nat.shard_layout(p, 2, 0).plan == 2
    monkeypatch.delenv("WFX_LINK_GBS")
    forced, _ = build_params(2, n0, 16000, 0.5, n_out=n, shard_plan=sharded.plan_code("dist"))
    assert nat.shard_layout(forced, 2, 0).plan == 2 and nat.shard_layout(forced, 2, 0).plan_forced == 1
    single, _ = build_params(2, n0, 16000, 0.5, n_out=n, shard_plan=sharded.plan_code("single"))
    assert nat.shard_layout(single, 8, 0).plan == 0 and nat.shard_layout(single, 8, 3).own_samples == 0
    nat.shard_dry_run(single, 8)
    # the 10-minute capture of configs[1]: 0.33 ms on one GPU -- only eight ranks' exchanges are short enough, and only just
    c2, _ = build_params(0, 7166250, 11025, 0.5)
    assert [nat.shard_layout(c2, w, 0).plan for w in (2, 4, 8)] == [0, 0, 2]
    monkeypatch.setenv("WFX_LINK_LAT_US", "40")
    assert [nat.shard_layout(c2, w, 0).plan for w in (2, 4, 8)] == [0, 0, 0]


@pytest.mark.parametrize("n0", [2 * 1000003, 1433252, 7166252, 2 * 3583126 + 2, 9000 * 2 + 2, 39690002, 2 * 104729, 600000 + 2 * 7919,
                                1433251, 7166251, 1000003, 39690001, 20001])
def test_any_length_at_the_native_rate_gets_a_padded_plan(n0):
    """An 11 025 Hz capture whose half-length has a prime factor above 13 (i.e. almost every real recording) is sharded too: the
    Hilbert convolution is embedded in a 13-smooth transform of Kp >= n - 1 points whose rows are dealt to the ranks; a rank owns
    the samples of its rows that lie inside the capture (the ranks whose rows are all padding own none and still take part in
    every exchange).  The layouts tile the capture and every rank's exchange lists agree (host-only dry run).
    ODD lengths (scipy's kernel has taps on every lag) take the same form as a REAL convolution on packed transforms: a point is a
    pair of samples here too, Kp >= n, and a glue step between the forward and inverse slab passes (round 4; before: one point per sample)."""
    for world in (1, 2, 3, 8):
        if n0 < 40000 and world == 8:
            continue
        # the rows layout (rounds 2-3; what one rank still takes): contiguous ranges that tile the capture
        p, meta = build_params(0, n0, 11025, 0.5, shard_plan=sharded.plan_code("rows"))
        lays = [nat.shard_layout(p, world, r) for r in range(world)]
        assert all(lay.plan == 1 and lay.nseg == 1 for lay in lays)
        assert lays[0].own_lo == 0 and lays[-1].own_hi == n0
        assert all(lays[i].own_hi == lays[i + 1].own_lo for i in range(world - 1))
        sizes = [lay.own_hi - lay.own_lo for lay in lays]
        assert min(sizes) >= 4096 and (world == 1 or max(sizes) - min(sizes) <= n0 // 12)     # only rows that hold samples are dealt
        assert all(lay.in_lo == max(0, lay.own_lo - 32) and lay.in_hi == min(n0, lay.own_hi + 32) for lay in lays if lay.own_hi > lay.own_lo)
        nat.shard_dry_run(p, world)
        # the columns layout (later in round 4, more than one rank): a rank's columns of EVERY row of the padded arrangement -- the
        # segments reach past the capture's end (the slots behind it hold nothing), their halos are 192 samples (filtfilt's exact
        # edge where the capture ends inside a segment), and every sample of the capture has exactly one owner
        p, meta = build_params(0, n0, 11025, 0.5, shard_plan=sharded.plan_code("dist"))
        lays = [nat.shard_layout(p, world, r) for r in range(world)]
        if world == 1 or lays[0].plan != 2:
            assert all(lay.plan == 1 for lay in lays)
        else:
            assert all(lay.plan == 2 and lay.nseg == lays[0].nseg and lay.in_halo == 192 for lay in lays)
            stride = int(lays[0].own_seg_stride)
            assert lays[0].own_lo == 0 and lays[-1].own_lo + lays[-1].own_seg_len == stride
            assert all(lays[i].own_lo + lays[i].own_seg_len == lays[i + 1].own_lo for i in range(world - 1))
            assert 2 * (n0 - 1) <= int(lays[0].nseg) * stride < 2.2 * n0 + stride       # the padded arrangement: Kp >= n - 1 POINTS
            owned = np.concatenate([lay.own_index() for lay in lays])
            owned = np.sort(owned[owned < n0])
            assert owned.shape[0] == n0 and owned[0] == 0 and owned[-1] == n0 - 1 and np.all(np.diff(owned) == 1)
        nat.shard_dry_run(p, world)



@pytest.mark.parametrize("n", [7166250, 1433250, 661500, 39690000, 32768, 2 * 1000003])
def test_multipole_plan_cuts_the_capture_at_leaf_workgroups_and_keeps_the_wire_in_kilobytes(n):
    """Plan 3 (round 6; csrc/wfx_shard.hip run_phase_fmm, csrc/wfx_fmm.hip): contiguous ranges whose ends are leaf-workgroup boundaries of a tree
    that depends on n alone; 320 frames of halo round the circle; the boxes a rank reads beyond its range are the boxes delivered (dry run);
    everything but the select's candidates and the final gather of the stream stays in kilobytes -- the gather level's weights + three boxes
    per finer level and side: ~25 KB per rank for the 10-minute capture on eight ranks (the transposing plans: ~100 MB)."""
    p, meta = build_params(0, n, 11025, 0.5, shard_plan=sharded.plan_code("fmm"))
    assert meta["n"] == n
    cuts = None
    for world in (1, 2, 3, 8):
        lays = [nat.shard_layout(p, world, r) for r in range(world)]
        assert all(lay.plan == 3 and lay.plan_forced == 1 and lay.nseg == 1 and lay.in_halo == 320 for lay in lays)
        assert lays[0].own_lo == 0 and lays[-1].own_hi == n and all(lays[i].own_hi == lays[i + 1].own_lo for i in range(world - 1))
        assert all(lay.in_lo == lay.own_lo and lay.in_hi == lay.own_hi and lay.in_frames == lay.own_samples + 640 for lay in lays)
        assert all(lay.in_index()[0] == int(lay.own_lo) - 320 and lay.in_index().shape[0] == lay.in_frames for lay in lays)
        sizes = [lay.own_samples for lay in lays]
        assert min(sizes) >= 2048 and max(sizes) <= 2 * min(sizes) + 4096
        if world == 8:
            cuts = {int(lay.own_lo) for lay in lays}
        nat.shard_dry_run(p, world)
        wire = {e["name"]: e for e in nat.shard_wire_plan(p, world)}
        assert list(wire) == ["fmm weights", "fmm seams", "select level 0", "select level 1", "select candidates", "stream gather"]
        if world == 1:
            assert all(e["bytes"] == 0 for e in wire.values())
            continue
        assert wire["fmm weights"]["max_rank_bytes"] <= 96 * 1024 and wire["fmm seams"]["max_rank_bytes"] == 64
        assert wire["stream gather"]["bytes"] == n - lays[0].own_samples + 8 * (world - 1)
        small = sum(wire[k]["max_rank_bytes"] for k in ("fmm weights", "fmm seams", "select level 0", "select level 1", "select candidates"))
        if n <= 7166250:
            assert small <= 600 * 1024, small
    # the cuts of 2 ranks are cuts of 8 ranks: the tree's boxes, not the world size, decide where a range may end
    lays2 = [nat.shard_layout(p, 2, r) for r in range(2)]
    assert {int(lay.own_lo) for lay in lays2} <= cuts


def test_the_cost_model_takes_the_multipole_plan_where_the_capture_is_long_enough(monkeypatch):
    """`plan="auto"`: a 60-minute capture at 11 025 Hz (39.69 M samples) is cut by plan 3 at every world size (its exchanges are kilobytes:
    model 1.7 / 1.0 / 0.64 ms on 2 / 4 / 8 ranks against 2.8 ms on one GPU and more for the transposing plan); the 10-minute capture is too
    short to be worth it below 8 ranks, where the transposing plan's model is still a little ahead."""
    for v in ("WFX_LINK_GBS", "WFX_LINK_LAT_US", "WFX_SHARD_CHUNKS", "WFX_SHARD_ROWS"):
        monkeypatch.delenv(v, raising=False)
    p, _ = build_params(0, 39690000, 11025, 0.5)
    for world in (2, 4, 8):
        lay = nat.shard_layout(p, world, 0)
        assert lay.plan == 3 and lay.plan_forced == 0
        assert lay.model_dist_compute_s + lay.model_dist_wire_s < 0.7 * lay.model_single_s
    p, _ = build_params(0, 7166250, 11025, 0.5)
    assert [nat.shard_layout(p, w, 0).plan for w in (2, 4, 8)] == [0, 0, 2]
    # resampled captures (round 6: the resampler has its multipole form too): the 60-minute IQ stream's hand-over signal (57.6 M -> 39.69 M) from
    # four ranks on; the 60-minute 48 kHz capture (172.8 M sources: the tree over them is the cost) on four ranks, where the transposing plan's
    # exchanges are still too long -- on eight that plan's model is a little ahead again
    p, _ = build_params(2, 57600000, 16000, 0.5, n_out=39690000)
    assert [nat.shard_layout(p, w, 0).plan for w in (4, 8)] == [3, 3]
    p, _ = build_params(0, 172800000, 48000, 0.5)
    assert [nat.shard_layout(p, w, 0).plan for w in (2, 4, 8)] == [0, 3, 2]


def test_an_oversampled_capture_left_open_is_sharded_from_two_ranks_on(monkeypatch):
    """The library's cost model sees the capture at the hand-over rate; the front end in front of it (the 22 GB ingest of configs[3]: 3.7 ms of
    the 6.7) is divided by the world size only under a sharded plan.  `FrontEndShardedDecoder(plan="auto")` therefore chooses with that time
    added (sharded.choose_plan_with_front_end, host only): the chunk-local plan from TWO ranks on, where the library alone would leave rank 0
    to ingest the whole stream."""
    for v in ("WFX_LINK_GBS", "WFX_LINK_LAT_US", "WFX_SHARD_CHUNKS", "WFX_SHARD_ROWS"):
        monkeypatch.delenv(v, raising=False)
    raw_bytes = 5529600000 * 4
    for world, want in ((2, 3), (4, 3), (8, 3)):
        name, fig = sharded.choose_plan_with_front_end(57600000, 16000, 120, None, 39690000, world, raw_bytes)
        assert fig["candidates"][name]["plan"] == want
        assert fig["candidates"][name]["model_s"] == min(c["model_s"] for c in fig["candidates"].values())
        assert abs(fig["front_end_s"] - raw_bytes / 6.0e12) < 1e-12
    name, fig = sharded.choose_plan_with_front_end(57600000, 16000, 120, None, 39690000, 2, raw_bytes)
    assert name == "fmm" and fig["candidates"]["auto"]["plan"] == 0 and fig["candidates"]["auto"]["model_s"] > 1.3 * fig["candidates"]["fmm"]["model_s"]


def test_multipole_plan_refuses_what_it_cannot_shard():
    p, _ = build_params(0, 480000, 8000, 0.5, shard_plan=sharded.plan_code("fmm"))         # upsampling: the resampler's multipole form is built for downsampling
    with pytest.raises(nat.NativeError, match="no multipole form of the resampler"):
        nat.shard_layout(p, 2, 0)
    p, _ = build_params(0, 1440000 - 3, 48000, 0.5, shard_plan=sharded.plan_code("fmm"))   # an odd count at 11 025 Hz
    assert _["n"] % 2 == 1
    with pytest.raises(nat.NativeError, match="no multipole form"):
        nat.shard_layout(p, 2, 0)
    p, _ = build_params(0, 100001, 11025, 0.5, shard_plan=sharded.plan_code("fmm"))        # odd: no multipole form
    with pytest.raises(nat.NativeError, match="no multipole form"):
        nat.shard_layout(p, 2, 0)
    p, _ = build_params(0, 40000, 11025, 0.5, shard_plan=sharded.plan_code("fmm"))         # 16 boxes at the gather level
    with pytest.raises(nat.NativeError, match="boxes at the gather level"):
        nat.shard_layout(p, 32, 0)


@pytest.mark.parametrize("kind,n0,sr,n_out", [(0, 2880000, 48000, None), (2, 57600000, 16000, 39690000), (0, 172800000, 48000, None), (1, 960000, 16000, None),
                                              (0, 1440001, 48000, None)])
def test_multipole_plan_in_front_of_a_resampler(kind, n0, sr, n_out):
    """Plan 3 for a capture at another rate (round 6): the resampler's own multipole form on a tree over the INPUT samples, the Hilbert transform's
    on a tree over the resampled ones; the ranks are dealt boxes of a level common to both, so a rank's input range and its range at 11 025 Hz
    are the same arc of the circle.  64 input frames of halo round the circle; two more exchanges (the resampler's weights and the parts of its
    constant; 320 resampled samples per seam), kilobytes both."""
    p, meta = build_params(kind, n0, sr, 0.5, shard_plan=sharded.plan_code("fmm"), **({"n_out": n_out} if n_out else {}))
    n = meta["n"]
    for world in (1, 2, 3, 8):
        lays = [nat.shard_layout(p, world, r) for r in range(world)]
        assert all(lay.plan == 3 and lay.nseg == 1 and lay.in_halo == 64 for lay in lays)
        assert lays[0].own_lo == 0 and lays[-1].own_hi == n and all(lays[i].own_hi == lays[i + 1].own_lo for i in range(world - 1))
        assert lays[0].in_lo == 0 and lays[-1].in_hi == n0 and all(lays[i].in_hi == lays[i + 1].in_lo for i in range(world - 1))
        for lay in lays:
            # the same arc: the first input sample at or behind the arc's start is the rank's first, likewise its first resampled sample
            assert abs(int(lay.in_lo) / n0 - int(lay.own_lo) / n) <= 1.0 / n and abs(int(lay.in_hi) / n0 - int(lay.own_hi) / n) <= 1.0 / n
            assert lay.in_frames == int(lay.in_hi - lay.in_lo) + 128 and lay.in_index()[0] == int(lay.in_lo) - 64
        nat.shard_dry_run(p, world)
        wire = {e["name"]: e for e in nat.shard_wire_plan(p, world)}
        assert list(wire) == ["resampler weights", "resampled halos", "fmm weights", "fmm seams", "select level 0", "select level 1", "select candidates", "stream gather"]
        if world == 1:
            assert all(e["bytes"] == 0 for e in wire.values())
            continue
        assert wire["resampler weights"]["max_rank_bytes"] <= 320 * 1024 and wire["resampled halos"]["max_rank_bytes"] == 2 * 320 * 8
        assert wire["stream gather"]["bytes"] == n - lays[0].own_samples + 8 * (world - 1)


def test_captures_without_a_distributed_form_get_the_single_plan():
    """A resampled capture whose half-lengths are odd or not 13-smooth (its inverse transform's length is the reference's to
    choose), or a capture too short for the world size, is not refused: rank 0 owns it whole and decodes it alone (first_radix
    (0, 0)), the other ranks own nothing.  Every rank reaches the same verdict from the description alone."""
    for n0, sr, world in [(1440001, 48000, 2),             # odd length, resampled: its transforms are packed
                          (4000, 11025, 8),                 # too short for the world size
                          (749700, 22050, 2),               # 34 s at 22 050 Hz: a factor 17 in both transforms
                          (1440002, 48000, 3), (792000, 44100, 8)]:      # (int(11025 * (792000 / 44100)) = 197999: odd)
        p, meta = build_params(0 if sr == 11025 else 2, n0, sr, 0.5, shard_plan=sharded.plan_code("dist"))
        lays = [nat.shard_layout(p, world, r) for r in range(world)]
        assert all(tuple(lay.first_radix) == (0, 0) and lay.plan == 0 for lay in lays)
        assert (lays[0].own_lo, lays[0].own_hi, lays[0].in_lo, lays[0].in_hi) == (0, meta["n"], 0, n0)
        assert all(lay.own_lo == lay.own_hi == meta["n"] and lay.in_lo == lay.in_hi == n0 for lay in lays[1:])
        nat.shard_dry_run(p, world)
        assert sharded.layout_supported(n0, sr, world, kind=0 if sr == 11025 else 2) and not sharded.layout_distributed(n0, sr, world, kind=0 if sr == 11025 else 2)
    assert sharded.layout_distributed(1433250, 11025, 8, kind=0)
    p, _ = build_params(0, 1433250, 11025, 0.5)
    with pytest.raises(nat.NativeError):
        nat.shard_layout(p, 2, 2)                                       # rank out of range
    p.hilbert_mode = nat.WFX_HILBERT_BLUESTEIN
    with pytest.raises(nat.NativeError):
        nat.shard_layout(p, 2, 0)                                       # the convolution form only


def _free_port():
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def _bootstrap_worker(rank, world, port, out_dir):
    uid = sharded.bootstrap_unique_id(rank, world, port=port, timeout=30.0, make_id=lambda: bytes(range(128)))
    # every rank also derives its own slice from the layout alone: nothing else is shared between the processes
    p, _ = build_params(0, 1433250, 11025, 0.5, shard_plan=sharded.plan_code("rows"))
    lay = nat.shard_layout(p, world, rank)
    with open(os.path.join(out_dir, f"r{rank}"), "wb") as fh:
        fh.write(uid + int(lay.own_lo).to_bytes(8, "little") + int(lay.own_hi).to_bytes(8, "little"))


def test_unique_id_bootstrap_between_processes(tmp_path):
    """world_size 3, real processes, loopback TCP: the ranks that start before rank 0 listens retry; everyone ends with
    rank 0's 128 bytes.  (ncclGetUniqueId itself needs a GPU: the id is a stand-in here, the transport is the real one.)"""
    port = _free_port()
    ctx = mp.get_context("spawn")
    procs = [ctx.Process(target=_bootstrap_worker, args=(r, 3, port, str(tmp_path))) for r in (2, 1, 0)]     # rank 0 last
    for pr in procs:
        pr.start()
    for pr in procs:
        pr.join(60)
        assert pr.exitcode == 0
    blobs = [open(os.path.join(tmp_path, f"r{r}"), "rb").read() for r in range(3)]
    assert all(b[:128] == bytes(range(128)) for b in blobs)
    own = [(int.from_bytes(b[128:136], "little"), int.from_bytes(b[136:144], "little")) for b in blobs]
    assert own[0][0] == 0 and own[2][1] == 1433250 and own[0][1] == own[1][0] and own[1][1] == own[2][0]


def _gloo_worker(rank, world, port, uid_port, out_dir):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        # what a rank of the sharded decode derives on its own: its slice of the capture, of the 60-minute 48 kHz configuration
        p, meta = build_params(0, 172800000, 48000, 0.5, shard_plan=sharded.plan_code("dist"))
        lay = nat.shard_layout(p, world, rank)
        # columns layout: the rank's columns of every row -- (first sample of its first segment, segment length, segments, stride)
        mine = dict(rank=rank, own=(int(lay.own_lo), int(lay.own_lo + lay.own_seg_len)), inp=(int(lay.in_lo), int(lay.in_lo + lay.in_seg_len)),
                    radix=tuple(lay.first_radix), nseg=int(lay.nseg), stride=(int(lay.own_seg_stride), int(lay.in_seg_stride)), plan=int(lay.plan))
        uid = sharded.bootstrap_unique_id(rank, world, port=uid_port, timeout=30.0, make_id=lambda: bytes(range(128)))
        mine["uid_ok"] = uid == bytes(range(128))
        everyone = [None] * world
        dist.all_gather_object(everyone, mine)
        dist.barrier()
        if rank == 0:
            import json
            with open(os.path.join(out_dir, "gathered.json"), "w") as fh:
                json.dump({"n": meta["n"], "ranks": everyone}, fh)
    finally:
        dist.destroy_process_group()


def test_two_gloo_processes_agree_on_the_partition(tmp_path):
    """world_size 2 over gloo on the CPU (torch only here, in the test): each process derives its own rows from the capture's
    parameters alone, fetches the communicator id over the product's TCP bootstrap, and the gathered pieces tile the capture."""
    pytest.importorskip("torch")
    import json
    ctx = mp.get_context("spawn")
    port, uid_port = _free_port(), _free_port()
    procs = [ctx.Process(target=_gloo_worker, args=(r, 2, port, uid_port, str(tmp_path))) for r in (1, 0)]
    for pr in procs:
        pr.start()
    for pr in procs:
        pr.join(120)
        assert pr.exitcode == 0
    g = json.load(open(os.path.join(tmp_path, "gathered.json")))
    r0, r1 = sorted(g["ranks"], key=lambda r: r["rank"])
    assert r0["uid_ok"] and r1["uid_ok"] and r0["radix"] == r1["radix"]
    # the two ranks' segments tile a row of the 11 025 Hz arrangement, and of the input's, and the rows tile the capture
    nseg = r0["nseg"]
    assert r0["plan"] == r1["plan"] == 2 and nseg == r1["nseg"] == r0["radix"][0] * r0["radix"][1] and r0["stride"] == r1["stride"]
    assert r0["own"][0] == 0 and r0["own"][1] == r1["own"][0] and r1["own"][1] == r0["stride"][0] and nseg * r0["stride"][0] == g["n"]
    assert r0["inp"][0] == 0 and r0["inp"][1] == r1["inp"][0] and r1["inp"][1] == r0["stride"][1] and nseg * r0["stride"][1] == 172800000


def _nonce_worker(rank, world, port, nonce, uid_byte, out_dir):
    uid = sharded.bootstrap_unique_id(rank, world, port=port, timeout=30.0, make_id=lambda: bytes([uid_byte]) * 128, nonce=nonce)
    with open(os.path.join(out_dir, f"{nonce}_{rank}"), "wb") as fh:
        fh.write(uid)


def test_two_jobs_on_one_port_range_do_not_serve_each_others_ranks(tmp_path):
    """Two decodes on one host probing the SAME port range: each rank 0 answers only peers that present its job's nonce, so the
    ranks of job A never end up with job B's communicator id (they skip to the next port of the range)."""
    port = _free_port()
    ctx = mp.get_context("spawn")
    procs = [ctx.Process(target=_nonce_worker, args=(r, 2, port, nonce, b, str(tmp_path)))
             for nonce, b in (("jobA", 0xA1), ("jobB", 0xB2)) for r in (1, 0)]
    for pr in procs:
        pr.start()
    for pr in procs:
        pr.join(90)
        assert pr.exitcode == 0
    for nonce, b in (("jobA", 0xA1), ("jobB", 0xB2)):
        for r in range(2):
            assert open(os.path.join(tmp_path, f"{nonce}_{r}"), "rb").read() == bytes([b]) * 128


def test_bootstrap_times_out_loudly_without_rank_zero():
    with pytest.raises(nat.NativeError):
        sharded.bootstrap_unique_id(1, 2, port=_free_port(), timeout=0.5)


# ---------------------------------------------------------------------------------------------------------------
# GPU
# ---------------------------------------------------------------------------------------------------------------
def _oracle(x, sr, lpm):
    from oracle import wefax_oracle as wo
    with tempfile.TemporaryDirectory() as td:
        path = os.path.join(td, "x.wav")
        synth.write_wav(path, sr, x)
        return wo.process(path, lpm, want_messages=False)


CASES = {
    "mono_11025_130s": lambda: (synth.synth_capture(11025.0, noise=0.05, seed=3, **KW130), 11025, 120),
    "mono_11025_240lpm": lambda: (synth.synth_capture(11025.0, noise=0.05, seed=6, lpm=240, start_tone_s=5.0, phasing_lines=40, image_lines=440,
                                                      stop_tone_s=2.0, black_tail_s=3.0), 11025, 240),
    # 30 s at 11 025 Hz: 165 375 packed points = 225 x 735, and 735 = 3 5 7^2 has no radix-pair decomposition (per-prime passes)
    "mono_11025_30s": lambda: (synth.synth_capture(11025.0, noise=0.05, seed=12, **KW30), 11025, 120),
    "mono_48000_30s": lambda: (synth.synth_capture(48000.0, noise=0.05, seed=4, **KW30), 48000, 120),
    "stereo_48000_30s": lambda: (synth.synth_capture(48000.0, noise=0.05, seed=5, iq=True, **KW30), 48000, 120),
    # two channels at the native rate: the merge of wefax.py:360-373 (int16 wrap) on every rank's own frames, no resampler
    "stereo_11025_130s": lambda: (synth.synth_capture(11025.0, noise=0.05, seed=13, iq=True, amplitude=0.9, **KW130), 11025, 120),
    "float_11025_130s": lambda: (synth.synth_capture(11025.0, noise=0.02, seed=8, **KW130).astype(np.float64) * 0.37, 11025, 120),
    # sample formats whose filtfilt odd extension scipy evaluates in the file's own dtype: uint8 wraps (the capture starts and ends
    # near the top of the range), int32 wraps, float32 rounds -- the sharded path takes the 9 + 9 numbers from the host like DecodeJob
    "uint8_11025_130s": lambda: (np.clip(synth.synth_capture(11025.0, noise=0.05, seed=9, **KW130).astype(np.int32) // 160 + 150, 0, 255).astype(np.uint8), 11025, 120),
    "int32_11025_130s": lambda: (synth.synth_capture(11025.0, noise=0.05, seed=10, **KW130).astype(np.int32) * 98000, 11025, 120),
    "float32_11025_130s": lambda: ((synth.synth_capture(11025.0, noise=0.05, seed=11, **KW130).astype(np.float64) / 32768.0 * 1.0000001).astype(np.float32), 11025, 120),
}


@pytest.mark.gpu
@pytest.mark.parametrize("layout", ["dist", "rows"])
@pytest.mark.parametrize("case", sorted(CASES))
def test_sharded_exact_decode_equals_the_oracle_for_every_world_size(case, layout):
    """max |delta pixel| 0 and equal start_frame / peaks against the oracle, for 1, 2, 3 and 8 ranks; every float stage
    bit-identical across world sizes; every rank ends with the same percentiles on its device.  Both layouts of the stencil
    stages: "dist" = the columns layout (round 4: no rows <-> columns transposes), "rows" = rounds 2-3."""
    x, sr, lpm = CASES[case]()
    if x.dtype == np.float64:
        import scipy.io.wavfile  # noqa: F401  (float wav: the oracle reads it like scipy does)
    ref = _oracle(x, sr, lpm)
    first = None
    for world in (1, 2, 3, 8):
        r = sharded.decode_emulated(x, sr, world, lpm, plan=layout)
        assert r["plan"] == (2 if layout == "dist" else 1)
        names = [e["name"] for e in r["wire"][0]]
        assert ("hilbert E1" in names) == (layout == "rows") and ("hilbert halo" in names) == (layout == "dist")
        assert np.array_equal(r["digitalized"], ref["digitalized"]), f"{case} world {world}: uint8 stream differs"
        assert np.array_equal(r["digitalized"], r["digitalized_blocks"])          # the gathered stream is the ranks' blocks
        assert r["sync"]["start_frame"] == ref["start_frame"] and r["sync"]["peaks"] == [int(v) for v in ref["peaks"]]
        assert r["sync"]["phasing"] == [int(v) for v in ref["phasing_signals"]]
        assert np.array_equal(r["image"], ref["image"]), f"{case} world {world}: image differs"
        assert len(set(r["lows"])) == 1 and len(set(r["highs"])) == 1
        assert abs(r["low"] - ref["low"]) <= 1e-9 * abs(ref["low"]) and abs(r["high"] - ref["high"]) <= 1e-9 * abs(ref["high"])
        scale = np.max(np.abs(ref["demod"]))
        assert np.max(np.abs(r["envelope"] - ref["demod"])) <= 1e-9 * scale
        assert np.max(np.abs(r["audio"] - ref["audio"])) <= 1e-9 * np.max(np.abs(ref["audio"]))
        if first is None:
            first = r
        else:
            for k in ("envelope", "audio", "digitalized", "image"):
                assert np.array_equal(r[k], first[k]), f"{case}: {k} depends on the world size ({world})"
            assert r["low"] == first["low"] and r["high"] == first["high"]


@pytest.mark.gpu
@pytest.mark.parametrize("case,layout", [("mono_11025_130s", "dist"), ("stereo_48000_30s", "dist"), ("mono_48000_30s", "rows")])
def test_bytes_counted_by_the_communicator_equal_the_plan(case, layout):
    """What every rank's communicator counted during a decode (wfx_comm_wire_stats: per collective, bytes to other ranks) adds up
    to what the host-only wire plan says for that capture and world size -- the figures bench.py prints are the plan's."""
    x, sr, lpm = CASES[case]()
    p, _ = build_params(sharded.capture_kind(x), x.shape[0], sr, 1 / (lpm / 60), shard_plan=sharded.plan_code(layout))
    for world in (2, 3, 8):
        r = sharded.decode_emulated(x, sr, world, lpm, plan=layout, want=("stream",))
        plan = nat.shard_wire_plan(p, world)
        seen = {}
        for rank_stats in r["wire"]:
            for e in rank_stats:
                seen[e["name"]] = seen.get(e["name"], 0) + e["sent"]
        want = {}
        for e in plan:                                    # (a transpose appears once per k1 subset)
            want[e["name"]] = want.get(e["name"], 0) + e["bytes"]
        assert seen == want, (world, seen, want)
        assert [e["name"] for e in r["wire"][0]] == [e["name"] for e in plan]      # the same collectives in the same order


@pytest.mark.gpu
def test_full_size_ten_minute_capture_on_eight_emulated_ranks():
    """BASELINE configs[1] at full size (7 166 250 samples): 8 ranks against the single-GPU exact path (itself pinned to the
    oracle by test_gpu_parity) -- identical stream, peaks, start_frame, image."""
    from wefax_amd.wefax import DecodeJob
    x = synth.config_c2(noise=0.05, seed=0)
    ctx = nat.Context(0)
    job = DecodeJob(ctx, x, 11025, 120)
    job.run()
    info = job.result()
    stream, img = job.fetch("digitalized"), job.fetch("image")
    ctx.close()
    r = sharded.decode_emulated(x, 11025, 8, 120, want=("image", "stream"))
    assert r["first_radix"] == (15, 15)
    assert np.array_equal(r["digitalized"], stream) and np.array_equal(r["image"], img)
    assert r["sync"]["start_frame"] == info.start_frame and r["sync"]["npeaks"] == info.npeaks



@pytest.mark.gpu
@pytest.mark.parametrize("case", ["mono_11025_130s", "mono_11025_30s", "float_11025_130s", "uint8_11025_130s", "float32_11025_130s", "stereo_11025_130s"])
def test_multipole_plan_gives_the_one_gpu_decodes_bytes_for_every_world_size(case):
    """Plan 3: every world size gives the bytes of the one-GPU decode in the same Hilbert form -- filtered audio, envelope, stream, start frame,
    image -- and the oracle's stream and image (max |delta pixel| 0); what the communicator counted is what the plan says."""
    from wefax_amd.wefax import DecodeJob
    x, sr, lpm = CASES[case]()
    ref = _oracle(x, sr, lpm)
    c = nat.Context(0)
    job = DecodeJob(c, x, sr, lpm, hilbert_mode=nat.WFX_HILBERT_FMM)
    job.run()
    info = job.result()
    one = {"digitalized": job.fetch("digitalized"), "envelope": job.fetch("envelope"), "audio": job.fetch("audio"), "image": job.fetch("image")}
    assert np.array_equal(one["digitalized"], ref["digitalized"]) and info.start_frame == ref["start_frame"]
    kind = sharded.capture_kind(x)
    p, _ = build_params(kind, x.shape[0], sr, 1 / (lpm / 60), shard_plan=sharded.plan_code("fmm"))
    for world in (1, 2, 3, 8):
        r = sharded.decode_emulated(x, sr, world, lpm, plan="fmm")
        assert r["plan"] == 3
        for k in ("digitalized", "envelope", "audio", "image"):
            assert np.array_equal(r[k], one[k]), f"{case} world {world}: {k} differs from the one-GPU decode"
        assert np.array_equal(r["digitalized"], r["digitalized_blocks"])
        assert r["sync"]["start_frame"] == ref["start_frame"] and r["sync"]["peaks"] == [int(v) for v in ref["peaks"]]
        assert np.array_equal(r["image"], ref["image"])
        assert len(set(r["lows"])) == 1 and len(set(r["highs"])) == 1 and r["low"] == info.low and r["high"] == info.high
        plan = nat.shard_wire_plan(p, world)
        counted = {}
        for ws in r["wire"]:
            for e in ws:
                counted[e["name"]] = counted.get(e["name"], 0) + int(e["sent"])
        for e in plan:
            assert counted.get(e["name"], 0) == e["bytes"], (world, e["name"], counted.get(e["name"]), e["bytes"])


RS_CASES = {
    "mono_48000_30s": CASES["mono_48000_30s"],
    "stereo_48000_30s": CASES["stereo_48000_30s"],
    "float_16000_60s": lambda: (synth.synth_capture(16000.0, noise=0.03, seed=21, start_tone_s=3.0, phasing_lines=20, image_lines=80, stop_tone_s=2.0,
                                                    black_tail_s=3.0).astype(np.float64) * 0.41, 16000, 120),
    "mono_44100_odd_count": lambda: (synth.synth_capture(44100.0, noise=0.05, seed=22, **KW30)[:-7], 44100, 120),
}


@pytest.mark.gpu
@pytest.mark.parametrize("case", sorted(RS_CASES))
def test_multipole_plan_in_front_of_a_resampler_gives_the_same_bytes_for_every_world_size(case):
    """Plan 3 for captures at another rate: resampler and Hilbert transform both chunk-local (csrc/wfx_shard.hip run_phase_rs + run_phase_fmm).
    Every world size gives the same bytes -- resampled + filtered audio, envelope, stream, image; against the oracle (whose resampler is scipy's
    transform over the capture: the same sums in another order, 1e-13 apart) the stream within the parity bar with the start frame equal; what
    the communicator counted is what the plan says."""
    from wefax_amd.wefax import DecodeJob
    x, sr, lpm = RS_CASES[case]()
    ref = _oracle(x, sr, lpm)
    kind = sharded.capture_kind(x)
    p, meta = build_params(kind, x.shape[0], sr, 1 / (lpm / 60), shard_plan=sharded.plan_code("fmm"))
    # the one-GPU decode on the multipole route (hilbert_mode 4 takes the resampler's multipole form too): the bytes every world size must give
    c = nat.Context(0)
    job = DecodeJob(c, x, sr, lpm, hilbert_mode=nat.WFX_HILBERT_FMM)
    job.run()
    info = job.result()
    one = {"digitalized": job.fetch("digitalized"), "envelope": job.fetch("envelope"), "audio": job.fetch("audio"), "image": job.fetch("image")}
    assert info.start_frame == ref["start_frame"]
    first = None
    for world in (1, 2, 3, 8):
        r = sharded.decode_emulated(x, sr, world, lpm, plan="fmm")
        assert r["plan"] == 3 and r["n"] == meta["n"]
        assert np.array_equal(r["digitalized"], r["digitalized_blocks"])
        assert r["sync"]["start_frame"] == ref["start_frame"]
        for k in ("digitalized", "envelope", "audio", "image"):
            assert np.array_equal(r[k], one[k]), f"{case} world {world}: {k} differs from the one-GPU decode on the multipole route"
        if first is None:
            first = r
            d = np.abs(r["digitalized"].astype(np.int16) - ref["digitalized"].astype(np.int16))
            assert d.max() <= 1 and np.count_nonzero(d) <= 2, (int(d.max()), int(np.count_nonzero(d)))
            assert np.abs(r["image"].astype(np.int16) - ref["image"].astype(np.int16)).max() <= 1
            scale = np.max(np.abs(ref["audio"])) if "audio" in ref else None
            if scale:
                assert np.max(np.abs(r["audio"] - ref["audio"])) <= 1e-11 * scale
        else:
            for k in ("digitalized", "envelope", "audio", "image"):
                assert np.array_equal(r[k], first[k]), f"{case} world {world}: {k} differs from world 1"
            assert r["low"] == first["low"] and r["high"] == first["high"]
        plan = nat.shard_wire_plan(p, world)
        counted = {}
        for ws in r["wire"]:
            for e in ws:
                counted[e["name"]] = counted.get(e["name"], 0) + int(e["sent"])
        for e in plan:
            assert counted.get(e["name"], 0) == e["bytes"], (world, e["name"], counted.get(e["name"]), e["bytes"])


@pytest.mark.gpu
def test_multipole_plan_full_size_ten_minute_capture_on_eight_emulated_ranks():
    from wefax_amd.wefax import DecodeJob
    x = synth.config_c2(noise=0.05, seed=2)
    c = nat.Context(0)
    job = DecodeJob(c, x, 11025, 120)                     # the default (transform) route
    job.run()
    info = job.result()
    r = sharded.decode_emulated(x, 11025, 8, 120, plan="fmm", want=("image", "stream"))
    assert r["plan"] == 3 and np.array_equal(r["digitalized"], job.fetch("digitalized")) and r["sync"]["start_frame"] == info.start_frame
    assert np.array_equal(r["image"], job.fetch("image"))
    sent = [sum(int(e["sent"]) for e in ws if e["name"] != "stream gather") for ws in r["wire"]]
    assert max(sent) <= 600 * 1024, sent                  # (the transposing plans: ~100 MB per rank)
    assert max(int(e["sent"]) for ws in r["wire"] for e in ws if e["name"] == "fmm weights") <= 32 * 1024


@pytest.mark.gpu
def test_rccl_communicator_with_one_rank():
    """The real transport as far as a one-GPU box can run it: librccl bound by the library, unique id from rank 0, every
    collective of the decode enqueued on the context's stream; the result equals the fused single-GPU decode."""
    from wefax_amd.wefax import DecodeJob
    x = synth.synth_capture(11025.0, noise=0.05, seed=3, **KW130)
    ctx = nat.Context(0)
    job = DecodeJob(ctx, x, 11025, 120)
    job.run()
    img, stream = job.fetch("image"), job.fetch("digitalized")
    uid = sharded.bootstrap_unique_id(0, 1)
    assert len(uid) == nat.WFX_COMM_ID_BYTES
    comm = nat.Comm.rccl(ctx, uid, 1, 0)
    assert comm.is_rccl and comm.world == 1 and comm.rank == 0
    dec = sharded.ShardedDecoder(ctx, comm, x.shape[0], 11025, 120, nat.WFX_IN_I16_MONO, data=x)
    for _ in range(3):                      # the shard's buffers and plans are reused across decodes
        dec.run()
    info = dec.result()
    assert np.array_equal(dec.fetch("image"), img) and np.array_equal(dec.fetch("stream"), stream)
    assert info.start_frame == job.result().start_frame
    dec.close()
    comm.close()
    ctx.close()


@pytest.mark.gpu
@pytest.mark.parametrize("rate,stereo", [(11025, False), (48000, True)])
def test_exchanges_of_k1_subsets_on_the_communicators_own_stream(monkeypatch, rate, stereo):
    """RCCL with the one rank a one-GPU box has, the k1 set cut into 4 subsets: every E2 / E3 goes to the communicator's stream
    (an event on the context's stream gates it, an event per slot releases the slab passes that consume it), the passes of one
    subset are enqueued while the exchange of the next is in flight -- and the decode is the fused one's, decode after decode.
    WFX_COMM_ASYNC=0 runs the same exchanges in stream order: same result."""
    from wefax_amd.wefax import DecodeJob
    kw = dict(KW30) if rate != 11025 else dict(KW130)
    x = synth.synth_capture(float(rate), noise=0.05, seed=5, iq=stereo, **kw)
    ctx = nat.Context(0)
    job = DecodeJob(ctx, x, rate, 120)
    job.run()
    img, stream, start = job.fetch("image"), job.fetch("digitalized"), job.result().start_frame
    monkeypatch.setenv("WFX_SHARD_CHUNKS", "4")
    uid = sharded.bootstrap_unique_id(0, 1)
    comm = nat.Comm.rccl(ctx, uid, 1, 0)
    dec = sharded.ShardedDecoder(ctx, comm, x.shape[0], rate, 120, sharded.capture_kind(x), data=x, plan="dist")
    pairs = 2 if rate != 11025 else 1                   # distributed transform pairs per decode: resampler + Hilbert, or Hilbert alone
    assert dec.shard.phases == (4 * 4 + 7 if rate != 11025 else 2 * 4 + 6)
    for rep in range(3):
        dec.run()
    info = dec.result()
    assert comm.async_exchanges == 3 * pairs * 2 * 4    # E2 and E3 of 4 subsets per pair and decode
    assert np.array_equal(dec.fetch("image"), img) and np.array_equal(dec.fetch("stream"), stream) and info.start_frame == start
    before = comm.async_exchanges
    monkeypatch.setenv("WFX_COMM_ASYNC", "0")
    dec.run()
    assert comm.async_exchanges == before and np.array_equal(dec.fetch("stream"), stream)
    dec.close()
    comm.close()
    ctx.close()


@pytest.mark.gpu
def test_a_bad_unique_id_is_a_comm_error_not_a_crash():
    ctx = nat.Context(0)
    with pytest.raises(ValueError):
        nat.Comm.rccl(ctx, b"short", 1, 0)
    with pytest.raises(nat.NativeError) as e:
        nat.Comm.rccl(ctx, bytes(128), 1, 3)         # rank outside the world
    assert "rank" in str(e.value)
    ctx.close()


def _arbitrary_even_lengths(count, seed, lo, hi):
    rng = np.random.default_rng(seed)
    return [int(2 * rng.integers(lo // 2, hi // 2)) for _ in range(count)]


def _arbitrary_odd_lengths(count, seed, lo, hi):
    return [n + 1 for n in _arbitrary_even_lengths(count, seed, lo, hi)]


@pytest.mark.gpu
@pytest.mark.parametrize("n", _arbitrary_even_lengths(6, 21, 300000, 900000) + [2 * 200003, 1433252] + _arbitrary_odd_lengths(4, 22, 300000, 900000) + [400009, 1433251])
def test_sharded_decode_of_any_length_equals_the_oracle(n):
    """Arbitrary lengths at 11 025 Hz (random even and odd ones, a prime and a prime times two, the 130-s capture plus one and two
    samples) through the padded distributed convolution, worlds 1 / 2 / 3 / 8: stream, peaks, start_frame, image equal to the
    oracle's; the float stages do not depend on the world size; decoding twice (the second decode skips the kernel's transform)
    gives the same again."""
    lines = max(12, int(n / 5512.5) - 40)
    x = synth.synth_capture(11025.0, noise=0.05, seed=n % 1000, start_tone_s=2.0, phasing_lines=20, image_lines=lines, stop_tone_s=1.0, black_tail_s=1.0)
    x = np.concatenate([x, x[:max(0, n - x.shape[0])]])[:n]
    assert x.shape[0] == n
    ref = _oracle(x, 11025, 120)
    first = None
    for world in (1, 2, 3, 8):
        r = sharded.decode_emulated(x, 11025, world, 120, repeat=2)
        assert np.array_equal(r["digitalized"], ref["digitalized"]), f"world {world}: uint8 stream differs"
        assert np.array_equal(r["digitalized"], r["digitalized_blocks"])
        if ref.get("exception") is None:
            assert r["sync"]["start_frame"] == ref["start_frame"] and r["sync"]["peaks"] == [int(v) for v in ref["peaks"]]
            assert np.array_equal(r["image"], ref["image"])
        else:
            assert r["sync"]["no_group"]
        assert len(set(r["lows"])) == 1 and len(set(r["highs"])) == 1
        scale = np.max(np.abs(ref["demod"]))
        assert np.max(np.abs(r["envelope"] - ref["demod"])) <= 1e-9 * scale
        if first is None:
            first = r
        else:
            assert np.array_equal(r["envelope"], first["envelope"]) and r["low"] == first["low"] and r["high"] == first["high"]


@pytest.mark.gpu
def test_sixty_minute_native_rate_capture_of_arbitrary_length_on_eight_ranks():
    """Full size: 60 minutes at 11 025 Hz plus one sample (odd: packed real transforms, 39.7 M-point padded arrangement) and plus two
    (even, non-smooth half) in the columns layout on 4 and 8 emulated ranks -- stream, start frame and image of the one-GPU decode."""
    from wefax_amd.wefax import DecodeJob
    x0 = synth.synth_capture(11025.0, noise=0.05, seed=3, image_lines=7110, black_tail_s=5.0)
    assert x0.shape[0] == 39690000
    for n in (39690001, 39690002):
        x = np.concatenate([x0, x0[:2]])[:n]
        ctx = nat.Context(0)
        # (the transposing plans' arithmetic is the transform route's: asked for by name -- a one-GPU decode of an even capture this long
        # takes the multipole route by default, 8e-14 from scipy where the transform route is 3e-15: over 39.7 M samples a byte or two may
        # sit one grey level apart)
        job = DecodeJob(ctx, x, 11025, 120, hilbert_mode=nat.WFX_HILBERT_FFT)
        job.run()
        info = job.result()
        stream, img = job.fetch("digitalized").copy(), job.fetch("image").copy()
        del job
        if n % 2 == 0:
            dflt = DecodeJob(ctx, x, 11025, 120)
            assert dflt.hilbert_mode == nat.WFX_HILBERT_FMM
            dflt.run()
            i2 = dflt.result()
            d = np.abs(dflt.fetch("digitalized").astype(np.int16) - stream.astype(np.int16))
            assert d.max() <= 1 and np.count_nonzero(d) <= 64 and i2.start_frame == info.start_frame
            del dflt
        ctx.close()
        for world in (4, 8):
            r = sharded.decode_emulated(x, 11025, world, 120, want=("image", "stream"))
            assert r["plan"] == 2
            assert np.array_equal(r["digitalized"], stream) and np.array_equal(r["image"], img)
            assert r["sync"]["start_frame"] == info.start_frame


@pytest.mark.gpu
@pytest.mark.parametrize("world,n", [(2, 600570), (2, 600599), (2, 601420), (2, 601421), (2, 633824), (2, 633877),
                                     (3, 602946), (3, 602975), (3, 602976), (3, 602977), (3, 602986), (3, 603039),
                                     (4, 601420), (4, 602394), (4, 602447), (8, 600599), (8, 601421), (8, 602394),
                                     (2, 112001), (2, 112010), (2, 112063), (3, 112002), (4, 112033), (8, 112062)])
def test_capture_that_ends_at_a_segment_boundary(world, n):
    """Padded forms in the columns layout: the capture ends INSIDE the padded arrangement.  These lengths put the end exactly at a
    rank's first column of a row, one sample before or behind it, 10 / 30 / 63 samples to either side -- where filtfilt's exact
    edge (the last 64 outputs, made of the last 127 samples) straddles two ranks' columns and each of them has to compute its part
    from its own segment and halo.  The 112 0xx lengths end 1..63 samples behind a ROW boundary (rank 0's first column): part of the
    edge then lies among the LAST rank's own samples of the row before, which sees the end through its right halo (round-4 advisor
    finding).  Found by `nat.shard_layout` for these world sizes; the test checks that they still are."""
    p, _ = build_params(0, n, 11025, 0.5, shard_plan=sharded.plan_code("dist"))
    lays = [nat.shard_layout(p, world, r) for r in range(world)]
    assert lays[0].plan == 2 and lays[0].nseg > 1 and lays[0].in_halo == 192
    stride = int(lays[0].own_seg_stride)
    d = min(min((n - int(lay.own_lo)) % stride, (int(lay.own_lo) - n) % stride) for lay in (lays if n < 200000 else lays[1:]))
    assert d <= 63
    lines = max(12, int(n / 5512.5) - 40)
    x = synth.synth_capture(11025.0, noise=0.05, seed=n % 1000, start_tone_s=2.0, phasing_lines=20, image_lines=lines, stop_tone_s=1.0, black_tail_s=1.0)
    x = np.concatenate([x, x[:max(0, n - x.shape[0])]])[:n]
    ref = _oracle(x, 11025, 120)
    r = sharded.decode_emulated(x, 11025, world, 120, repeat=2)
    assert np.array_equal(r["digitalized"], ref["digitalized"]), "uint8 stream differs"
    assert np.array_equal(r["digitalized"], r["digitalized_blocks"])
    scale = np.max(np.abs(ref["audio"]))
    assert np.max(np.abs(r["audio"] - ref["audio"])) <= 1e-9 * scale                    # the notch's output up to the very last sample
    if ref.get("exception") is None:
        assert r["sync"]["start_frame"] == ref["start_frame"] and np.array_equal(r["image"], ref["image"])


@pytest.mark.gpu
@pytest.mark.parametrize("fs,n0,stereo", [(48000, 1440001, False), (48000, 1200002 + 2 * 7919, True), (22050, 749700, False), (44100, 792000, False),
                                          (11025, 6000, False)])
def test_captures_on_the_single_plan_decode_like_the_one_gpu_path(fs, n0, stereo):
    """Resampled captures of arbitrary length (odd; a prime factor above 13; a reference length int(11025 n0 / fs) that comes out
    odd) and a capture too short for eight ranks: the sharded interface takes them -- rank 0 decodes alone, the others receive
    the scalars -- and the result is the oracle's for every world size."""
    lines = max(12, int(n0 / fs * 2) - 30)
    x = synth.synth_capture(float(fs), noise=0.03, seed=n0 % 997, start_tone_s=1.0, phasing_lines=20, image_lines=lines, stop_tone_s=0.5, black_tail_s=0.5)
    x = np.concatenate([x, x[:max(0, n0 - x.shape[0])]])[:n0]
    data = np.stack([x, (x // 3).astype(np.int16)], axis=1) if stereo else x
    ref = _oracle(data, fs, 120)
    for world in (1, 2, 3, 8):
        assert not sharded.layout_distributed(n0, fs, world, kind=sharded.capture_kind(data)) or fs == 11025 and world < 8
        r = sharded.decode_emulated(data, fs, world, 120, repeat=2)
        assert np.array_equal(r["digitalized"], ref["digitalized"]), f"world {world}: uint8 stream differs"
        assert np.array_equal(r["digitalized"], r["digitalized_blocks"])
        if ref.get("exception") is None:
            assert r["sync"]["start_frame"] == ref["start_frame"] and r["sync"]["peaks"] == [int(v) for v in ref["peaks"]]
            assert np.array_equal(r["image"], ref["image"])
        else:
            assert r["sync"]["no_group"]
        assert len(set(r["lows"])) == 1 and len(set(r["highs"])) == 1           # the scalars reached every rank


@pytest.mark.gpu
def test_candidate_overflow_of_the_percentile_select_is_reported_and_recovered():
    """A constant capture has ONE envelope value: every key lands in the same bin and the candidate lists of the select
    outgrow what travels in the all-gather.  The decode says so -- on every rank alike, so no rank runs ahead --, raises the
    capacity, and a later decode of the same shard completes: low == high, and the quantiser reports the NaNs for which the
    reference raises ValueError (int(nan), wefax.py:216)."""
    x = np.full(1433250, 1200, dtype=np.int16)
    ref = _oracle(x, 11025, 120)
    assert isinstance(ref.get("exception"), ValueError)
    comms = nat.Comm.local(2)
    ctxs = [nat.Context(0) for _ in range(2)]
    decs = [sharded.ShardedDecoder(ctxs[r], comms[r], x.shape[0], 11025, 120, nat.WFX_IN_I16_MONO, data=x, plan="dist") for r in range(2)]
    overflows = 0
    infos = None
    for attempt in range(5):
        for ph in range(decs[0].shard.phases):
            for d in decs:
                d.shard.phase(ph)
        errs = []
        got = []
        for d in decs:
            try:
                got.append(d.result())
            except nat.NativeError as e:
                errs.append(str(e))
        assert len(errs) in (0, 2)                       # both ranks or neither
        if not errs:
            infos = got
            break
        assert all("overflow" in e for e in errs)
        overflows += 1
    assert overflows >= 1 and infos is not None
    assert infos[0].low == infos[0].high == infos[1].low == infos[1].high
    assert infos[0].nan_count > 1000000                  # nearly every sample equals the one percentile value: 0 / 0
    for d in decs:
        d.close()
    for c in comms:
        c.close()
    for c in ctxs:
        c.close()


@pytest.mark.gpu
def test_driving_a_local_world_out_of_step_is_an_error():
    x = synth.synth_capture(11025.0, noise=0.05, seed=3, **KW130)
    comms = nat.Comm.local(2)
    ctxs = [nat.Context(0) for _ in range(2)]
    decs = [sharded.ShardedDecoder(ctxs[r], comms[r], x.shape[0], 11025, 120, nat.WFX_IN_I16_MONO, data=x, plan="dist") for r in range(2)]
    with pytest.raises(nat.NativeError):
        decs[0].run()                          # all phases of one rank while the other has not started: refused, not a hang
    with pytest.raises(nat.NativeError):
        decs[0].result()                       # nothing has run
    for d in decs:
        d.close()
    for c in comms:
        c.close()
    for c in ctxs:
        c.close()


@pytest.mark.gpu
@pytest.mark.parametrize("fs,iq", [(11025.0, False), (192000.0, True)])
def test_device_synthesis_equals_the_numpy_generator(fs, iq):
    from wefax_amd import synth_device
    ctx = nat.Context(0)
    kw = dict(start_tone_s=1.0, phasing_lines=10, image_lines=20, stop_tone_s=1.0, black_tail_s=1.0)
    ref = synth.synth_capture(fs, noise=0.0, seed=0, iq=iq, **kw)
    p = synth_device.synth_params(fs, noise=0.0, iq=iq, **kw)
    n0 = int(ctx.lib.wfx_synth_frames(p))
    assert n0 == ref.shape[0] == synth_device.capture_frames(fs, **kw)
    ptr = synth_device.synth_slice(ctx, p, 0, n0)
    got = ctx.dev_download(ptr, ref.shape, np.int16)
    d = np.abs(got.astype(np.int32) - ref.astype(np.int32))
    assert d.max() <= 1 and np.count_nonzero(d) <= 1e-5 * d.size       # rounding ties of the phase sum at most
    lo, hi = n0 - 1000, n0 + 1500                                      # a slice that wraps around the end of the capture
    ptr2 = synth_device.synth_slice(ctx, p, lo, hi)
    got2 = ctx.dev_download(ptr2, (hi - lo,) + ref.shape[1:], np.int16)
    assert np.array_equal(got2, got[np.arange(lo, hi) % n0])
    pn = synth_device.synth_params(fs, noise=0.05, seed=7, iq=iq, **kw)
    ptr3 = synth_device.synth_slice(ctx, pn, 0, n0)
    noise = ctx.dev_download(ptr3, ref.shape, np.int16).astype(np.float64) - got
    assert abs(noise.std() / (0.05 * 32767) - 1) < 0.02 and abs(noise.mean()) < 20
    ptr4 = synth_device.synth_slice(ctx, pn, 5000, 9000)               # same frames from another range: same noise
    part = ctx.dev_download(ptr4, (4000,) + ref.shape[1:], np.int16)
    assert np.array_equal(part, ctx.dev_download(ptr3, ref.shape, np.int16)[5000:9000])
    for q in (ptr, ptr2, ptr3, ptr4):
        ctx.dev_free(q)
    ctx.close()
