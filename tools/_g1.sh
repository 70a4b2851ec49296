#!/bin/bash
set -u
OUT=$PWD/gpurun_out/anytest
mkdir -p "$OUT"
timeout 1500 python -m pytest tests/test_gpu_configs.py -x -q -m gpu -k "arbitrary_length" > "$OUT/pytest.log" 2>&1; tail -5 "$OUT/pytest.log"
