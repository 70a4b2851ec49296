#!/bin/bash
set -u
OUT=$PWD/gpurun_out/pred3
mkdir -p "$OUT"
timeout 900 python -m pytest tests/test_pred_select.py -q > "$OUT/pytest_pred.log" 2>&1; tail -3 "$OUT/pytest_pred.log"
cd /tmp && export TMPDIR=/tmp && cd - > /dev/null
WFX_PRED_MIN_N=1 rocprofv3 --kernel-trace --output-format csv -d "$OUT/tr_c3_1" -o run -- python3 bench.py --workload c3 --steps 3 --warmup 1 --no-cpu > /dev/null 2> "$OUT/err_tr.txt"
python tools/kseq.py "$OUT/tr_c3_1" > "$OUT/kseq_c3_pred1.txt" 2>&1; rm -rf "$OUT/tr_c3_1"
grep -E 'pred_|select_|env_median|quantise|sum of' "$OUT/kseq_c3_pred1.txt"
for rep in 1 2 3; do for m in 0 1; do
  WFX_PRED_MIN_N=$m python bench.py --workload c3 --no-cpu --steps 20 --warmup 3 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('c3 pred$m', d['ms_per_step'])"
done; done
for rep in 1 2; do for m in 0 1; do
  WFX_PRED_MIN_N=$m python bench.py --workload iq --no-cpu --steps 10 --warmup 2 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('iq pred$m', d['ms_per_step'])"
  WFX_PRED_MIN_N=$m python bench.py --no-cpu --no-c4 --no-pcie --no-extras 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('c2 pred$m', d['ms_per_step'])"
done; done
