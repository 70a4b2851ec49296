#!/bin/bash
# The core of tools/collect_evidence.sh at the FINAL tree of a round: GPU test suite, smoke, the default bench line, per-kernel durations and
# the PMC traffic summaries of configs[1] / [2] / [3] (what bench.py's `roofline.traffic` quotes, labelled with the csrc hash).
#   gpurun --timeout 1800 -- 'bash tools/collect_core.sh r05_v4'
set -u
TAG=${1:-core}
OUT=$PWD/gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
export WFX_EVIDENCE_TAG="profiles/$TAG ($(python -c "import hashlib,glob;h=hashlib.sha1();[h.update(open(f,'rb').read()) for f in sorted(glob.glob('wefax_amd/csrc/*.h*'))];print('csrc sha1 '+h.hexdigest()[:12])"))"
python -m pytest tests -m gpu -q > "$OUT/pytest_gpu.log" 2>&1
grep -E "passed|failed" "$OUT/pytest_gpu.log" | tail -1
python -c "import __graft_entry__ as g; g.smoke()" > "$OUT/smoke.log" 2>&1; tail -1 "$OUT/smoke.log"
python bench.py > "$OUT/bench.json" 2> "$OUT/bench.err"
head -c 300 "$OUT/bench.json"; echo
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -o run -- python3 bench.py --steps 10 --warmup 2 --no-cpu --no-c4 --no-pcie --no-extras > "$OUT/prof_bench.json" 2> "$OUT/prof_bench.err"
find "$OUT/trace" -name '*kernel_stats.csv' -exec cp {} "$OUT/kernel_stats.csv" \;
rm -rf "$OUT/trace"
for w in c2 iq c3; do
  case $w in
    c2) args="--steps 3 --warmup 1 --no-cpu --no-c4 --no-pcie --no-extras"; sfx="";;
    iq) args="--workload iq --steps 2 --warmup 1 --no-cpu"; sfx="_iq";;
    c3) args="--workload c3 --steps 2 --warmup 1 --no-cpu --no-extras"; sfx="_c3";;
  esac
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_fetch$sfx" -o run -- python3 bench.py $args > /dev/null 2>> "$OUT/bench.err"
  rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_write$sfx" -o run -- python3 bench.py $args > /dev/null 2>> "$OUT/bench.err"
  python tools/pmc_summary.py "$OUT/pmc_fetch$sfx" "$OUT/pmc_write$sfx" "$OUT/pmc_traffic$sfx.json" > "$OUT/pmc_summary$sfx.log" 2>&1
  rm -rf "$OUT/pmc_fetch$sfx" "$OUT/pmc_write$sfx"
done
for w in iq c3; do
  [ $w = iq ] && args="--workload iq --steps 5 --warmup 1 --no-cpu" || args="--workload c3 --steps 5 --warmup 1 --no-cpu --no-extras"
  rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace_$w" -o run -- python3 bench.py $args > /dev/null 2>> "$OUT/bench.err"
  python tools/kstats.py "$OUT/trace_$w" "ingest_stream|decimate|select_|notch|median|image|quantise|sync|mr2_pass|mr_pass|resample" > "$OUT/kernel_stats_$w.txt"
  rm -rf "$OUT/trace_$w"
done
ls "$OUT"
