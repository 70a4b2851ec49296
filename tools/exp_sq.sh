#!/bin/bash
# SQ counter pass for one kernel family:  bash tools/exp_sq.sh '<name regex>' ['VAR=a VAR2=b']
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/exp_sq; rm -rf "$OUT"; mkdir -p "$OUT"
for kv in $2; do export "$kv"; done
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU --output-format csv -d "$OUT/a" -o run -- python3 bench.py --steps 3 --warmup 1 --no-cpu --no-c4 --no-pcie --no-extras $EXP_ARGS > /dev/null 2> "$OUT/err.txt"
python3 tools/pmc_sq.py "$OUT/a" "$1"
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE --output-format csv -d "$OUT/b" -o run -- python3 bench.py --steps 3 --warmup 1 --no-cpu --no-c4 --no-pcie --no-extras $EXP_ARGS > /dev/null 2>> "$OUT/err.txt"
python3 - "$OUT/b" "$1" <<'PY'
import csv, glob, os, re, sys
from collections import defaultdict
acc = defaultdict(lambda: defaultdict(float)); cnt = defaultdict(lambda: defaultdict(int))
pat = re.compile(sys.argv[2])
for f in glob.glob(os.path.join(sys.argv[1], "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        name = re.sub(r"\(.*", "", r["Kernel_Name"].replace("void ", ""))
        if not pat.search(name): continue
        acc[name][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[name][r["Counter_Name"]] += 1
for name in sorted(acc):
    print(name[:44], " ".join(f"{k}={acc[name][k]/cnt[name][k]:.4g}" for k in sorted(acc[name])))
PY
tail -3 "$OUT/err.txt" | cut -c1-200
# third pass: matrix-core activity
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES --output-format csv -d "$OUT/c" -o run -- python3 bench.py --steps 3 --warmup 1 --no-cpu --no-c4 --no-pcie --no-extras $EXP_ARGS > /dev/null 2>> "$OUT/err.txt"
python3 tools/pmc_sq.py "$OUT/c" "$1"
