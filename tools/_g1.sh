#!/bin/bash
set -u
OUT=$PWD/gpurun_out/skip1
mkdir -p "$OUT"
python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py -x -q -m gpu > "$OUT/pytest.log" 2>&1; tail -3 "$OUT/pytest.log"
time python bench.py --no-cpu > "$OUT/bench.json" 2> "$OUT/err.txt"
python - "$OUT/bench.json" <<'PY'
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("c2", d["ms_per_step"], "gl", {k:(v.get("ms_per_step"), v.get("digitalized_mismatches")) for k,v in d["general_length"].items() if isinstance(v,dict)}, "c3", d["c3"]["ms_per_step"], "c4", d["c4_strong"]["ms_per_step"])
PY
