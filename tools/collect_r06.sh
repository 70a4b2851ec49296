#!/bin/bash
# Round-6 evidence set on an MI355X box:   gpurun --timeout 3000 -- 'bash tools/collect_r06.sh r06_v1'
# Everything lands in gpurun_out/<tag>/; what should be judged is copied to profiles/<tag>/ afterwards.
set -u
TAG=${1:-r06}
OUT=$PWD/gpurun_out/$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
CSRC=$(python -c "import hashlib,glob;h=hashlib.sha1();[h.update(open(f,'rb').read()) for f in sorted(glob.glob('wefax_amd/csrc/*.h*'))];print('csrc sha1 '+h.hexdigest()[:12])")
export WFX_EVIDENCE_TAG="profiles/$TAG ($CSRC)"
echo "$CSRC" > "$OUT/csrc_hash.txt"

python -m pytest tests -m gpu -q > "$OUT/pytest_gpu.log" 2>&1
grep -E "passed|failed" "$OUT/pytest_gpu.log" | tail -1
python __graft_entry__.py smoke > "$OUT/smoke.log" 2>&1; tail -1 "$OUT/smoke.log"

# the default bench line, three times (un-profiled)
for k in 1 2 3; do python bench.py > "$OUT/bench$k.json" 2> "$OUT/bench$k.err"; done
cp "$OUT/bench1.json" "$OUT/bench.json"
python tools/show_bench.py "$OUT/bench.json" > "$OUT/bench_summary.txt" 2>&1; head -3 "$OUT/bench_summary.txt"

# per-kernel durations: the headline on the transform route and on the multipole route; configs[2]; configs[3]
for mode in fft fmm; do
  export WEFAX_HILBERT=$mode
  rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace_$mode" -o run -- python3 bench.py --steps 10 --warmup 2 --no-cpu --no-c4 --no-c5 --no-pcie --no-extras --no-e2e > "$OUT/prof_bench_$mode.json" 2> "$OUT/prof_bench_$mode.err"
  find "$OUT/trace_$mode" -name '*kernel_stats.csv' -exec cp {} "$OUT/kernel_stats_$mode.csv" \;
  rm -rf "$OUT/trace_$mode"
done
unset WEFAX_HILBERT
cp "$OUT/kernel_stats_fft.csv" "$OUT/kernel_stats.csv"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace_c3" -o run -- python3 bench.py --workload c3 --steps 5 --warmup 2 --no-cpu --no-e2e > "$OUT/prof_bench_c3.json" 2> "$OUT/prof_bench_c3.err"
find "$OUT/trace_c3" -name '*kernel_stats.csv' -exec cp {} "$OUT/kernel_stats_c3.csv" \;
rm -rf "$OUT/trace_c3"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace_iq" -o run -- python3 bench.py --workload iq --steps 5 --warmup 2 --no-cpu > "$OUT/prof_bench_iq.json" 2> "$OUT/prof_bench_iq.err"
find "$OUT/trace_iq" -name '*kernel_stats.csv' -exec cp {} "$OUT/kernel_stats_iq.csv" \;
rm -rf "$OUT/trace_iq"

# HBM traffic, one counter per pass, no trace domains besides the kernel trace: headline (both routes)
for mode in fft fmm; do
  export WEFAX_HILBERT=$mode
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_fetch" -o run -- python3 bench.py --steps 3 --warmup 1 --no-cpu --no-c4 --no-c5 --no-pcie --no-extras --no-e2e > /dev/null 2> "$OUT/pmc_fetch_$mode.err"
  rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_write" -o run -- python3 bench.py --steps 3 --warmup 1 --no-cpu --no-c4 --no-c5 --no-pcie --no-extras --no-e2e > /dev/null 2> "$OUT/pmc_write_$mode.err"
  python tools/pmc_summary.py "$OUT/pmc_fetch" "$OUT/pmc_write" "$OUT/pmc_traffic_$mode.json" > "$OUT/pmc_summary_$mode.log" 2>&1
  rm -rf "$OUT/pmc_fetch" "$OUT/pmc_write"
done
unset WEFAX_HILBERT
cp "$OUT/pmc_traffic_fft.json" "$OUT/pmc_traffic.json"
# ... configs[2] and configs[3] (what their `roofline.traffic` quotes)
if [ "${COLLECT_PMC_BIG:-1}" = "1" ]; then
  for w in iq c3; do
    [ $w = iq ] && args="--workload iq --steps 2 --warmup 1 --no-cpu" || args="--workload c3 --steps 2 --warmup 1 --no-cpu --no-extras --no-e2e"
    rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_fetch_$w" -o run -- python3 bench.py $args > /dev/null 2> "$OUT/pmc_fetch_$w.err"
    rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_write_$w" -o run -- python3 bench.py $args > /dev/null 2> "$OUT/pmc_write_$w.err"
    python tools/pmc_summary.py "$OUT/pmc_fetch_$w" "$OUT/pmc_write_$w" "$OUT/pmc_traffic_$w.json" > "$OUT/pmc_summary_$w.log" 2>&1
    rm -rf "$OUT/pmc_fetch_$w" "$OUT/pmc_write_$w"
  done
fi

# SQ counters of the multipole kernels
EXP_ARGS="--no-e2e --no-c5" bash tools/exp_sq.sh "fmm_|mr2_pass|notch|hconv" "WEFAX_HILBERT=fmm" > "$OUT/sq_counters_fmm.txt" 2>&1
# f64 matrix / vector rates of this box
tools/micro/build/mfma_f64_rate > "$OUT/mfma_f64_rate.txt" 2>&1

# plan 3 of the sharded decode: every world size against the one-GPU decode, bytes on the wire
python tools/shard_fmm_check.py 650 1 2 3 8 > "$OUT/shard_fmm_check.txt" 2>&1; tail -1 "$OUT/shard_fmm_check.txt"
python bench.py --shard --plan fmm --no-c4 --no-cpu > "$OUT/bench_shard_fmm_rccl1.json" 2>> "$OUT/bench.err"
# the resampler's multipole form: accuracy + time on the device, plan 3 in front of it in emulated worlds (48 kHz and 16 kHz captures),
# the 60-minute IQ stream through the sharded interface at ONE rank on plan 3 and on the transposing plan (= each plan's total work), and
# plan 3's kernels under rocprofv3
python tools/rs_fmm_check.py --big > "$OUT/rs_fmm_check.txt" 2>&1; tail -2 "$OUT/rs_fmm_check.txt"
python tools/shard_rs_check.py 48000 200 1 2 3 8 > "$OUT/shard_rs_check_48k.txt" 2>&1; tail -1 "$OUT/shard_rs_check_48k.txt"
python tools/shard_rs_check.py 16000 300 1 2 5 8 > "$OUT/shard_rs_check_16k.txt" 2>&1; tail -1 "$OUT/shard_rs_check_16k.txt"
for pl in fmm dist; do
  python bench.py --workload iq --iq-form sharded --plan $pl --steps 10 --warmup 3 --no-cpu > "$OUT/bench_iq_world1_$pl.json" 2> "$OUT/bench_iq_world1_$pl.err"
done
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace_iq_fmm" -o run -- python3 bench.py --workload iq --iq-form sharded --plan fmm --steps 5 --warmup 2 --no-cpu > "$OUT/prof_bench_iq_fmm.json" 2> "$OUT/prof_bench_iq_fmm.err"
find "$OUT/trace_iq_fmm" -name '*kernel_stats.csv' -exec cp {} "$OUT/kernel_stats_iq_plan3.csv" \;
rm -rf "$OUT/trace_iq_fmm"
# two ranks on this one GPU over the shm transport: the N > 1 bench path end to end (protocol, not speed: the ranks share the GPU)
WFX_BENCH_OVERSUBSCRIBE=1 WFX_BENCH_COMM=shm python bench.py --gpus 2 --steps 5 --warmup 2 --no-cpu > "$OUT/bench_n2_shm_one_gpu.json" 2> "$OUT/bench_n2_shm_one_gpu.err"
ls "$OUT"
