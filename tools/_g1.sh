#!/bin/bash
set -u
OUT=$PWD/gpurun_out/odd2
mkdir -p "$OUT"
python -m pytest tests/test_sharded.py tests/test_comm_shm.py -x -q -m gpu > "$OUT/pytest_shard.log" 2>&1; tail -3 "$OUT/pytest_shard.log"
export WFX_BENCH_OVERSUBSCRIBE=1
for lib in new old; do
  if [ $lib = old ]; then export WFX_LIB=$PWD/wefax_amd/csrc/build/old_odd.so; else unset WFX_LIB; fi
  for g in 4 8; do
    timeout 600 python bench.py --gpus $g --shard --trim 1 --plan dist --no-c4 --no-pcie --steps 5 --warmup 2 > "$OUT/bench_shard_trim1_shm${g}_$lib.json" 2> "$OUT/err_${g}_$lib.txt"
    python - "$OUT/bench_shard_trim1_shm${g}_$lib.json" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[1].split('/')[-1], d["ms_per_step"], (d.get("wire") or {}).get("total_bytes"), d.get("parity_vs_oracle"))
except Exception as e:
    print(sys.argv[1], "ERR", e)
PY
  done
done
for f in "$OUT"/err_*.txt; do tail -n 2 "$f"; done
