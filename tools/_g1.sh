#!/bin/bash
set -u
OUT=$PWD/gpurun_out/czt3
mkdir -p "$OUT"
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py tests/test_sharded.py -x -q -m gpu > "$OUT/pytest.log" 2>&1; tail -5 "$OUT/pytest.log"
timeout 600 python tools/resample_any_length.py --minutes 10 > "$OUT/any_10min.jsonl" 2> "$OUT/err10.txt"; cat "$OUT/any_10min.jsonl"; tail -n 3 "$OUT/err10.txt"
timeout 900 python tools/resample_any_length.py --minutes 60 > "$OUT/any_60min.jsonl" 2> "$OUT/err60.txt"; cat "$OUT/any_60min.jsonl"; tail -n 3 "$OUT/err60.txt"
