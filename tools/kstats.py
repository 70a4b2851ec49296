#!/usr/bin/env python3
"""Print a rocprofv3 kernel_stats.csv compactly: short kernel name, calls, average and total microseconds.
    python tools/kstats.py <dir or csv> [name filter regex]"""
import csv, glob, os, re, sys
path = sys.argv[1]
files = [path] if os.path.isfile(path) else glob.glob(os.path.join(path, "**", "*kernel_stats.csv"), recursive=True)
pat = re.compile(sys.argv[2]) if len(sys.argv) > 2 else None
for f in files:
    for r in csv.DictReader(open(f)):
        name = r["Name"]
        if pat and not pat.search(name):
            continue
        short = re.sub(r"\(.*", "", name.replace("void ", "").replace("(anonymous namespace)::", ""))
        print(f"{short[:60]:60s} calls {int(r['Calls']):6d}  avg {float(r['AverageNs'])/1e3:10.1f} us  total {float(r['TotalDurationNs'])/1e6:9.2f} ms")
