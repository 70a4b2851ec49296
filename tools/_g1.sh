#!/bin/bash
set -u
OUT=$PWD/gpurun_out/head1
mkdir -p "$OUT"
python -m pytest tests -x -q -m gpu > "$OUT/pytest_gpu.log" 2>&1; tail -4 "$OUT/pytest_gpu.log"
/usr/bin/time -v python -c "import __graft_entry__ as g; g.smoke()" > "$OUT/smoke.log" 2>&1; grep -E 'Elapsed|smoke|max' "$OUT/smoke.log" | head -5
/usr/bin/time -v python bench.py > "$OUT/bench.json" 2> "$OUT/bench.time"; grep -E 'Elapsed' "$OUT/bench.time"; head -c 400 "$OUT/bench.json"; echo
