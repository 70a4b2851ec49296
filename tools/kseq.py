#!/usr/bin/env python3
"""Print the kernels of ONE step of a rocprofv3 --kernel-trace run in launch order with their durations and the gaps between
them: the dispatches between the last two launches of an anchor kernel (default: the notch).
    python tools/kseq.py <dir or kernel_trace.csv> [anchor regex]"""
import csv, glob, os, re, sys
path = sys.argv[1]
files = [path] if os.path.isfile(path) else glob.glob(os.path.join(path, "**", "*kernel_trace.csv"), recursive=True)
anchor = re.compile(sys.argv[2] if len(sys.argv) > 2 else "notch_kernel")
rows = []
for f in files:
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
idx = [i for i, r in enumerate(rows) if anchor.search(r[2])]
if len(idx) < 2:
    sys.exit("fewer than two launches of the anchor kernel")
a, b = idx[-2], idx[-1]
prev_end = None
tot = 0
for s, e, name in rows[a:b]:
    short = re.sub(r"\(.*", "", name.replace("void ", "").replace("(anonymous namespace)::", ""))[:58]
    gap = (s - prev_end) / 1e3 if prev_end is not None else 0.0
    print(f"{short:58s} {(e - s) / 1e3:9.1f} us   gap {gap:6.1f}")
    prev_end = e
    tot += e - s
print(f"sum of kernel durations {tot / 1e3:.1f} us, span {(rows[b][0] - rows[a][0]) / 1e3:.1f} us")
