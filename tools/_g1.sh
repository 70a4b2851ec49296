#!/bin/bash
set -u
TAG=r04_v3
OUT=$PWD/gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
export WFX_EVIDENCE_TAG="profiles/$TAG ($(python -c "import hashlib,glob;h=hashlib.sha1();[h.update(open(f,'rb').read()) for f in sorted(glob.glob('wefax_amd/csrc/*.h*'))];print('csrc sha1 '+h.hexdigest()[:12])"))"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace_c3" -o run -- python3 bench.py --workload c3 --steps 5 --warmup 1 --no-cpu --no-extras > /dev/null 2>> "$OUT/bench2.err"
python tools/kstats.py "$OUT/trace_c3" "select_|notch|median|image|quantise|sync|mr2_pass|mr_pass|resample" > "$OUT/kernel_stats_c3.txt"
rm -rf "$OUT/trace_c3"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_fetch_c3" -o run -- python3 bench.py --workload c3 --steps 2 --warmup 1 --no-cpu --no-extras > /dev/null 2>> "$OUT/bench2.err"
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_write_c3" -o run -- python3 bench.py --workload c3 --steps 2 --warmup 1 --no-cpu --no-extras > /dev/null 2>> "$OUT/bench2.err"
python tools/pmc_summary.py "$OUT/pmc_fetch_c3" "$OUT/pmc_write_c3" "$OUT/pmc_traffic_c3.json" > /dev/null 2>&1
rm -rf "$OUT/pmc_fetch_c3" "$OUT/pmc_write_c3"
head -c 600 "$OUT/pmc_traffic_c3.json"; echo; head -12 "$OUT/kernel_stats_c3.txt"
