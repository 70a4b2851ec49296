#!/usr/bin/env python3
"""Summarise an SQ counter pass of rocprofv3 per kernel (mean per dispatch, and ratios to SQ_WAVE_CYCLES).
    rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU \
              SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d DIR -- python3 bench.py ...
    python tools/pmc_sq.py DIR [name filter regex]"""
import csv, glob, os, re, sys
from collections import defaultdict
d = sys.argv[1]
pat = re.compile(sys.argv[2]) if len(sys.argv) > 2 else None
acc = defaultdict(lambda: defaultdict(float))
cnt = defaultdict(lambda: defaultdict(int))
for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        name = re.sub(r"\(.*", "", r["Kernel_Name"].replace("void ", "").replace("(anonymous namespace)::", ""))
        if pat and not pat.search(name):
            continue
        acc[name][r["Counter_Name"]] += float(r["Counter_Value"])
        cnt[name][r["Counter_Name"]] += 1
for name in sorted(acc):
    m = {k: acc[name][k] / cnt[name][k] for k in acc[name]}
    wc = m.get("SQ_WAVE_CYCLES", 0.0) or 1.0
    parts = [f"{k.replace('SQ_', '')}={v / wc:.3f}" for k, v in sorted(m.items()) if k != "SQ_WAVE_CYCLES"]
    print(f"{name[:44]:44s} n={max(cnt[name].values()):4d} WAVE_CYCLES={wc:.3g}  " + " ".join(parts))
