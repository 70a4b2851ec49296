#!/bin/bash
# What kind of box is this, and how fast does it run the configs[3] ingest?  (round-3 verdict: the ingest kernel measures 4.4 ms on
# some boxes and 5.1 on others.)  Clocks / power limits as rocm-smi reports them, the f64 issue rate (tracks the shader clock), the
# plain read-stream ceiling (tracks HBM), and the ingest kernel itself on 16 GiB -- while sampling the clocks under that load.
#   gpurun -- 'bash tools/box_probe.sh <tag>'  ->  gpurun_out/box_probe_<tag>.txt
set -u
TAG=${1:-box}
OUT=$PWD/gpurun_out/box_probe_$TAG.txt
mkdir -p "$PWD/gpurun_out"
{
  echo "== identity"; rocm-smi --showuniqueid --showserial 2>/dev/null | grep -E 'Unique|Serial' | head -4
  echo "== limits and idle clocks"; rocm-smi --showmaxpower --showpower --showclocks --showperflevel 2>/dev/null | grep -vE '^=|^$|WARNING' | head -30
  echo "== f64 issue rate"; [ -x tools/micro/build/f64_rate ] && timeout 60 tools/micro/build/f64_rate 2>&1 | head -8
  echo "== read-stream ceiling"; [ -x tools/micro/build/stream_big ] && timeout 120 tools/micro/build/stream_big 2>&1 | tail -4
  echo "== ingest probe (16 GiB), clocks sampled every 0.5 s while it runs"
  ( for i in $(seq 1 40); do rocm-smi --showclocks --showpower 2>/dev/null | grep -E 'sclk|mclk|fclk|Power' | tr -s ' ' | tr '\n' ';'; echo; sleep 0.5; done ) > /tmp/clk_samples.txt &
  SAMPLER=$!
  timeout 300 python tools/ingest_lab.py taps 2>&1 | tail -5
  kill $SAMPLER 2>/dev/null; wait $SAMPLER 2>/dev/null
  echo "-- clock samples under load (distinct lines, with counts)"; sort /tmp/clk_samples.txt | uniq -c | sort -rn | head -12
  echo "== configs[3] on this box"; python bench.py --workload iq --no-cpu --steps 10 --warmup 2 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernels']
print('ms_per_step', d['ms_per_step'], 'ingest_us', k.get('polyphase_in',{}).get('avg_us'), 'stage2_us', k.get('polyphase',{}).get('avg_us'), 'roofline', d['roofline']['frac'])"
} > "$OUT" 2>&1
cat "$OUT"
