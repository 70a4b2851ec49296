#!/bin/bash
set -u
OUT=$PWD/gpurun_out/czt4
mkdir -p "$OUT"
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "resample or chirp" > "$OUT/pytest.log" 2>&1; tail -5 "$OUT/pytest.log"
for f in 1 0; do
  echo "WFX_FUSED_SPECTRUM=$f"
  WFX_FUSED_SPECTRUM=$f timeout 900 python tools/resample_any_length.py --minutes 60 2> "$OUT/err60_$f.txt" | cut -c1-100
  WFX_FUSED_SPECTRUM=$f timeout 600 python tools/resample_any_length.py --minutes 10 2> "$OUT/err10_$f.txt" | cut -c1-100
done
echo default; timeout 900 python tools/resample_any_length.py --minutes 60 2> "$OUT/err60_d.txt" | cut -c1-100
