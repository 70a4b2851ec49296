#!/usr/bin/env python3
"""Turn rocprofv3 --pmc counter CSVs into profiles/pmc_traffic.json.

    python tools/pmc_summary.py <dir with FETCH_SIZE run> <dir with WRITE_SIZE run> <out.json>

Per kernel group (the names bench.py reports): average FETCH_SIZE and WRITE_SIZE per
launch, converted to bytes (rocprofv3 reports KiB) with the gfx950 correction of
/opt/skills/guides/MI355X_MICROARCH.md section HBM applied: FETCH_SIZE counts 64 B per
128-byte request on wide coalesced streaming reads, so it is doubled; WRITE_SIZE is exact
for 16-byte-per-lane streaming stores.
"""
import csv
import glob
import json
import os
import re
import sys
from collections import defaultdict

GROUPS = [
    (r"fmm_up_leaf", "fmm_notch_p2m_m2m"), (r"fmm_up_tier|fmm_top|fmm_down_tier", "fmm_tiers_and_top"),
    (r"fmm_tree_leaf_env|fmm_leaf_env|fmm_edge_median", "fmm_near_l2p_env_median"), (r"fmm_tree_leaf", "fmm_tree_levels"),
    (r"rs_up_leaf|rs_csum|rs_halo", "resample_fmm_p2m_m2m"), (r"rs_leaf", "resample_fmm_near_l2p"),
    (r"fft_pass<\d+, 1,", "fft_pass_fwd"), (r"fft_pass<\d+, -1,", "fft_pass_inv"),
    (r"mr_pass<\d+, \d+, 0>", "fft_pass_fwd"), (r"mr_pass<\d+, \d+, 1>", "fft_pass_inv"),
    (r"mr2_pass<\d+, \d+, \d+, \d+, 0(, \d+)?>", "fft_pass_fwd"), (r"mr2_pass<\d+, \d+, \d+, \d+, 1(, \d+)?>", "fft_pass_inv"),      # (RA, RB, IN_MODE, OUT_MODE, INVERSE[, NTL])
    (r"notch_kernel", "notch_filtfilt"), (r"hconv_env_median|hilbert_abs|hconv_env\b", "env_median"),
    (r"select_l0|select_l1|select_compact", "select_hist"), (r"select_finish|select_lerp", "select_scan"),
    (r"quantise_kernel|quantise_corr_kernel", "quantise"), (r"sync_corr_kernel", "sync_corr"), (r"sync_pick_kernel", "sync_pick"),
    (r"image_kernel", "lines_to_image"), (r"median5_kernel", "median5"), (r"fir_hilbert", "fir_analytic"),
    (r"merge_kernel|i16_to_f64", "merge_channels"), (r"bs_|hilbert_mid|hconv_fill|hconv_pack", "bluestein_pointwise"),
    (r"resample_", "resample_pointwise"),
    (r"ingest_stream_kernel<[01], [0-9], false|decimate_kernel<[01],", "polyphase_ingest"), (r"ingest_stream_kernel<[01], [0-9], true", "polyphase_ingest_tail"), (r"decimate_kernel<[23],|rational_kernel", "polyphase_stages"),
    (r"mr_padded_fill", "bluestein_pointwise"),
    (r"select_level_kernel", "select_hist"), (r"quantise_kernel", "quantise"),
]


def group_of(name):
    for pat, g in GROUPS:
        if re.search(pat, name):
            return g
    return None


def collect(d, counter):
    tot = defaultdict(float)
    cnt = defaultdict(int)
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        with open(f) as fh:
            for row in csv.DictReader(fh):
                if row.get("Counter_Name") != counter:
                    continue
                g = group_of(row.get("Kernel_Name", ""))
                if g is None:
                    continue
                tot[g] += float(row["Counter_Value"])
                cnt[g] += 1
    for f in glob.glob(os.path.join(d, "pmc_*_per_kernel.csv")) if os.path.isdir(d) else [d]:
        with open(f) as fh:
            for row in csv.DictReader(fh):
                if row.get("counter") != counter:
                    continue
                g = group_of(row["kernel"])
                if g is None:
                    continue
                tot[g] += float(row["mean_value"]) * int(row["dispatches"])
                cnt[g] += int(row["dispatches"])
    return {g: (tot[g] / cnt[g], cnt[g]) for g in tot}


def main():
    fetch_dir, write_dir, out = sys.argv[1:4]
    fetch = collect(fetch_dir, "FETCH_SIZE")
    write = collect(write_dir, "WRITE_SIZE")
    res = {}
    for g in sorted(set(fetch) | set(write)):
        f_kib, nf = fetch.get(g, (0.0, 0))
        w_kib, nw = write.get(g, (0.0, 0))
        res[g] = {
            "fetch_size_kib_per_launch_raw": round(f_kib, 1),
            "write_size_kib_per_launch_raw": round(w_kib, 1),
            "launches_seen": [nf, nw],
            "hbm_bytes_per_launch": int(round((2.0 * f_kib + w_kib) * 1024)),
            "note": "FETCH_SIZE doubled (gfx950 wide-read correction), WRITE_SIZE as is",
        }
    # where and when these counters were taken: bench.py quotes it next to `roofline.traffic` (the line itself does not run rocprofv3)
    res["_collected_at"] = os.environ.get("WFX_EVIDENCE_TAG", "unknown tree") + ", per-launch means over " + ", ".join(
        f"{g}: {v['launches_seen'][0]}" for g, v in list(res.items())[:3]) + " ... launches"
    with open(out, "w") as fh:
        json.dump(res, fh, indent=1)
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
