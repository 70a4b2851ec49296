#!/usr/bin/env python3
"""Print the kernels of ONE step of a rocprofv3 --kernel-trace run in launch order with their durations and the gaps between
them: the dispatches between the last two launches of an anchor kernel (default: the notch).
    python tools/kseq.py <dir or kernel_trace.csv> [anchor regex]
Each line also carries the HSA queue of the dispatch and its start relative to the step's first kernel, and a `*` where it began
before the previous dispatch had ended -- two streams at work (the sharded decode's exchanges on the communicator's own stream)."""
import csv, glob, os, re, sys
path = sys.argv[1]
files = [path] if os.path.isfile(path) else glob.glob(os.path.join(path, "**", "*kernel_trace.csv"), recursive=True)
anchor = re.compile(sys.argv[2] if len(sys.argv) > 2 else "notch_kernel")
rows = []
for f in files:
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "?")))
rows.sort()
idx = [i for i, r in enumerate(rows) if anchor.search(r[2])]
if len(idx) < 2:
    sys.exit("fewer than two launches of the anchor kernel")
a, b = idx[-2], idx[-1]
prev_end = None
tot = 0
t0 = rows[a][0]
for s, e, name, q in rows[a:b]:
    short = re.sub(r"\(.*", "", name.replace("void ", "").replace("(anonymous namespace)::", ""))[:58]
    gap = (s - prev_end) / 1e3 if prev_end is not None else 0.0
    print(f"{short:58s} {(e - s) / 1e3:9.1f} us   gap {gap:7.1f}   queue {q:>3s}   start {(s - t0) / 1e3:9.1f} {'*' if gap < 0 else ''}")
    prev_end = max(prev_end or e, e)
    tot += e - s
print(f"sum of kernel durations {tot / 1e3:.1f} us, span {(rows[b][0] - rows[a][0]) / 1e3:.1f} us")
