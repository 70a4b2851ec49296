#!/bin/bash
# Collect the judged evidence set on an MI355X box (run through gpurun from the
# repo root):   gpurun --timeout 2400 -- 'bash tools/collect_evidence.sh r02_v1'
# Everything lands in gpurun_out/<tag>/; copy what should be judged to profiles/<tag>/.
set -u
TAG=${1:-evidence}
OUT=$PWD/gpurun_out/$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
export WFX_EVIDENCE_TAG="profiles/$TAG ($(python -c "import hashlib,glob;h=hashlib.sha1();[h.update(open(f,'rb').read()) for f in sorted(glob.glob('wefax_amd/csrc/*.h*'))];print('csrc sha1 '+h.hexdigest()[:12])"))"

python -m pytest tests -m gpu -q > "$OUT/pytest_gpu.log" 2>&1
grep -E "passed|failed" "$OUT/pytest_gpu.log" | tail -1

# un-profiled default bench line (the contract's N=1 run: configs[1] + the c4_strong object)
python bench.py > "$OUT/bench.json" 2> "$OUT/bench.err"
head -c 600 "$OUT/bench.json"; echo

# per-kernel durations of the same workload (configs[1] only: --no-c4 keeps the 22 GB stream out of the trace)
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -o run -- python3 bench.py --steps 10 --warmup 2 --no-cpu --no-c4 --no-pcie --no-extras > "$OUT/prof_bench.json" 2> "$OUT/prof_bench.err"
find "$OUT/trace" -name '*kernel_stats.csv' -exec cp {} "$OUT/kernel_stats.csv" \;

# HBM traffic, one counter per pass, no trace domains besides the kernel trace
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_fetch" -o run -- python3 bench.py --steps 3 --warmup 1 --no-cpu --no-c4 --no-pcie --no-extras > /dev/null 2> "$OUT/pmc_fetch.err"
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_write" -o run -- python3 bench.py --steps 3 --warmup 1 --no-cpu --no-c4 --no-pcie --no-extras > /dev/null 2> "$OUT/pmc_write.err"
python tools/pmc_summary.py "$OUT/pmc_fetch" "$OUT/pmc_write" "$OUT/pmc_traffic.json" > "$OUT/pmc_summary.log" 2>&1
for d in fetch write; do
  f=$(find "$OUT/pmc_$d" -name '*counter_collection.csv' | head -1)
  [ -n "$f" ] && python - "$f" "$OUT/pmc_${d}_per_kernel.csv" <<'EOF'
import csv, sys, collections
acc = collections.OrderedDict()
for r in csv.DictReader(open(sys.argv[1])):
    k = (r["Kernel_Name"], r["Counter_Name"])
    a = acc.setdefault(k, [0, 0.0])
    a[0] += 1
    a[1] += float(r["Counter_Value"])
with open(sys.argv[2], "w") as f:
    f.write("kernel,counter,dispatches,mean_value\n")
    for (k, c), (n, s) in acc.items():
        f.write('"%s",%s,%d,%.1f\n' % (k, c, n, s / n))
EOF
done
rm -rf "$OUT/trace" "$OUT/pmc_fetch" "$OUT/pmc_write"

# the other configurations
for b in 2 4 8; do python bench.py --batch $b --no-cpu > "$OUT/bench_batch$b.json" 2>> "$OUT/bench.err"; done
# the sharded exact path with the ONE rank a one-GPU box has, over the real transport (RCCL bound by the library)
WFX_BENCH_FORCE_DIST=1 python bench.py --shard --no-c4 > "$OUT/bench_shard_rccl1.json" 2>> "$OUT/bench.err"
WFX_BENCH_FORCE_DIST=1 python bench.py --workload iq --iq-seconds 450 --no-cpu > "$OUT/bench_iq_450s_rccl1.json" 2>> "$OUT/bench.err"
python bench.py --workload iq --iq-seconds 450 --no-cpu > "$OUT/bench_iq_450s_fused.json" 2>> "$OUT/bench.err"
WFX_BENCH_FORCE_DIST=1 python bench.py --workload iq --no-cpu > "$OUT/bench_iq_3600s_rccl1.json" 2>> "$OUT/bench.err"
# BASELINE configs[2] (60 minutes at 48 kHz) and configs[3] (60 minutes at 1.536 MS/s IQ) as lines of their own, CPU leg included
python bench.py --workload c3 > "$OUT/bench_c3.json" 2>> "$OUT/bench.err"
WFX_BENCH_FORCE_DIST=1 python bench.py --workload c3 --no-cpu > "$OUT/bench_c3_rccl1.json" 2>> "$OUT/bench.err"
python bench.py --workload iq > "$OUT/bench_iq_3600s.json" 2>> "$OUT/bench.err"
python tools/e2e.py > "$OUT/e2e_c2.json" 2>> "$OUT/bench.err"
python tools/e2e_breakdown.py > "$OUT/e2e_breakdown.txt" 2>> "$OUT/bench.err"
# the PNG encoded on the device (deflate) against zlib: decodes to the image, sizes, times
python tools/png_deflate_check.py --full > "$OUT/png_deflate.jsonl" 2>> "$OUT/bench.err"
# every rank of 1 / 2 / 3 / 8 emulated on this GPU, against the oracle
python tools/shard_check.py --cases plain,resample,stereo,lpm240,c2 > "$OUT/shard_check.jsonl" 2>> "$OUT/bench.err"
python tools/shard_check2.py iq > "$OUT/shard_check_iq.jsonl" 2>> "$OUT/bench.err"

# BASELINE configs[3] under the profiler
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace_iq" -o run -- python3 bench.py --workload iq --steps 5 --warmup 1 --no-cpu > /dev/null 2>> "$OUT/bench.err"
python tools/kstats.py "$OUT/trace_iq" "ingest_stream|decimate|rational|select_|notch|median|image|quantise|sync|mr2_pass|mr_pass|resample|synth|dist_" > "$OUT/kernel_stats_iq.txt"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_fetch_iq" -o run -- python3 bench.py --workload iq --steps 2 --warmup 1 --no-cpu > /dev/null 2>> "$OUT/bench.err"
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_write_iq" -o run -- python3 bench.py --workload iq --steps 2 --warmup 1 --no-cpu > /dev/null 2>> "$OUT/bench.err"
python tools/pmc_summary.py "$OUT/pmc_fetch_iq" "$OUT/pmc_write_iq" "$OUT/pmc_traffic_iq.json" > /dev/null 2>&1
rm -rf "$OUT/trace_iq" "$OUT/pmc_fetch_iq" "$OUT/pmc_write_iq"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace_c3" -o run -- python3 bench.py --workload c3 --steps 5 --warmup 1 --no-cpu --no-extras > /dev/null 2>> "$OUT/bench.err"
python tools/kstats.py "$OUT/trace_c3" "select_|notch|median|image|quantise|sync|mr2_pass|mr_pass|resample" > "$OUT/kernel_stats_c3.txt"
rm -rf "$OUT/trace_c3"
# BASELINE configs[2] asks for "rocprof HBM GB/s": FETCH_SIZE / WRITE_SIZE per kernel, separate passes
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_fetch_c3" -o run -- python3 bench.py --workload c3 --steps 2 --warmup 1 --no-cpu --no-extras > /dev/null 2>> "$OUT/bench.err"
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_write_c3" -o run -- python3 bench.py --workload c3 --steps 2 --warmup 1 --no-cpu --no-extras > /dev/null 2>> "$OUT/bench.err"
python tools/pmc_summary.py "$OUT/pmc_fetch_c3" "$OUT/pmc_write_c3" "$OUT/pmc_traffic_c3.json" > /dev/null 2>&1
rm -rf "$OUT/pmc_fetch_c3" "$OUT/pmc_write_c3"

# N real processes on this ONE GPU (shared-memory transport): spawn -> rendezvous -> every phase -> gather, checked against the oracle
# (--plan dist: the cost model would decline the distributed plan at 2 and 4 ranks, and this is about running it)
for g in 2 4 8; do
  WFX_BENCH_OVERSUBSCRIBE=1 timeout 900 python bench.py --gpus $g --shard --plan dist --steps 3 --warmup 1 --no-c4 --no-cpu-loops > "$OUT/bench_shard_shm$g.json" 2>> "$OUT/bench.err"
done
# BASELINE configs[3] on 8 processes: the columns layout (4 array transposes, in k1 subsets) and the rows layout of rounds 2-3 (8): the
# `wire` object has every collective's bytes; the host-staged transport's time follows them
WFX_BENCH_OVERSUBSCRIBE=1 timeout 900 python bench.py --gpus 8 --workload iq --plan dist --steps 3 --warmup 1 --no-cpu > "$OUT/bench_c4_shm8.json" 2>> "$OUT/bench.err"
WFX_BENCH_OVERSUBSCRIBE=1 timeout 900 python bench.py --gpus 8 --workload iq --plan rows --steps 3 --warmup 1 --no-cpu > "$OUT/bench_c4_shm8_rows.json" 2>> "$OUT/bench.err"
# the default line at 2 ranks: the cost model takes the single plan for c4_strong (rank 0 alone) instead of a distributed one that loses
WFX_BENCH_OVERSUBSCRIBE=1 timeout 900 python bench.py --gpus 2 --steps 3 --warmup 1 --no-pcie --no-cpu > "$OUT/bench_default_shm2.json" 2>> "$OUT/bench.err"
# the driver's own launch for N > 1 (torch.distributed.run: ranks from the environment), here with the ranks sharing the one GPU
for g in 2 4; do
  WFX_BENCH_OVERSUBSCRIBE=1 timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node $g --master-addr 127.0.0.1 --master-port 2955$g bench.py --gpus $g --steps 5 --warmup 2 > "$OUT/bench_torchrun_shm$g.json" 2>> "$OUT/bench.err"
done
# a padded sharded decode (the 10-minute capture less two samples): only the sample-bearing rows dealt vs all rows
for g in 8; do
  WFX_BENCH_OVERSUBSCRIBE=1 timeout 600 python bench.py --gpus $g --shard --plan dist --trim 2 --steps 5 --warmup 2 --no-c4 --no-cpu --no-extras --no-pcie > "$OUT/bench_shard_trim2_shm${g}.json" 2>> "$OUT/bench.err"
done
# the real transport with the one rank there is, exchanges in 4 k1 subsets on the communicator's own stream / in stream order
WFX_SHARD_CHUNKS=4 WFX_BENCH_FORCE_DIST=1 python bench.py --workload iq --no-cpu > "$OUT/bench_iq_rccl1_chunks4.json" 2>> "$OUT/bench.err"
WFX_SHARD_CHUNKS=4 WFX_COMM_ASYNC=0 WFX_BENCH_FORCE_DIST=1 python bench.py --workload iq --no-cpu > "$OUT/bench_iq_rccl1_chunks4_inorder.json" 2>> "$OUT/bench.err"

# the same, one decode in launch order with the queue of every dispatch: the exchanges (copies on the communicator's stream) against the passes
WFX_SHARD_CHUNKS=4 WFX_BENCH_FORCE_DIST=1 rocprofv3 --kernel-trace --output-format csv -d "$OUT/tr_ov" -o run -- python3 bench.py --workload iq --iq-seconds 450 --steps 3 --warmup 1 --no-cpu > /dev/null 2>> "$OUT/bench.err"
python tools/kseq.py "$OUT/tr_ov" 'ingest_stream' > "$OUT/kseq_iq450_rccl1_chunks4.txt" 2>&1
rm -rf "$OUT/tr_ov"
# captures of ANY length through the resampler (chirp-z form) against the whole-second ones
timeout 600 python tools/resample_any_length.py --minutes 10 > "$OUT/any_length_10min.jsonl" 2>> "$OUT/bench.err"
timeout 900 python tools/resample_any_length.py --minutes 60 > "$OUT/any_length_60min.jsonl" 2>> "$OUT/bench.err"
python -c "import __graft_entry__ as g; g.smoke()" > "$OUT/smoke.log" 2>&1

# randomised whole-path parity sweep against the oracle
timeout 1200 python tools/random_parity.py --cases 200 --seed 14 > "$OUT/random_parity.jsonl" 2>> "$OUT/bench.err"; tail -1 "$OUT/random_parity.jsonl"

timeout 2400 python tools/random_fe_parity.py --cases 120 --seed 9 > "$OUT/random_fe_parity.jsonl" 2>> "$OUT/bench.err"; tail -1 "$OUT/random_fe_parity.jsonl" | cut -c1-400
timeout 1200 python tools/random_shard_parity.py --cases 80 --seed 6 > "$OUT/random_shard_parity.jsonl" 2>> "$OUT/bench.err"; tail -1 "$OUT/random_shard_parity.jsonl"

# round 5: the changed launch forms of bench.py (default line with e2e, 2 ranks over shm with both plans and timed collectives, the RCCL
# self-test failing -> fallback, one RCCL rank with event-timed collectives), the ingest kernel taken apart, instruction rates
bash tools/r05_bench_paths.sh "$TAG/paths" > "$OUT/bench_paths.txt" 2>&1
timeout 600 python tools/ingest_lab.py 16 > "$OUT/ingest_lab_16GiB.txt" 2>> "$OUT/bench.err"
timeout 600 python tools/ingest_lab.py stream > "$OUT/ingest_lab_60min_stream.txt" 2>&1
timeout 600 python tools/ingest_lab.py where > "$OUT/ingest_lab_placement.txt" 2>> "$OUT/bench.err"
timeout 600 python tools/ingest_lab.py allocations > "$OUT/ingest_lab_allocations.txt" 2>> "$OUT/bench.err"
timeout 600 python tools/ingest_lab.py realloc 16 > "$OUT/alloc_lab.txt" 2>> "$OUT/bench.err"
[ -x tools/micro/build/valu_rate ] && timeout 120 tools/micro/build/valu_rate > "$OUT/valu_rate.txt" 2>&1
[ -x tools/micro/build/stream_pattern ] && timeout 300 tools/micro/build/stream_pattern > "$OUT/stream_pattern.txt" 2>&1
python tools/e2e_c3_debug.py 2>&1 | grep -v sync_pick > "$OUT/e2e_c3_stages.txt"
# where the ingest kernel's cycles go (SQ counters of the IQ workload, one pass)
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d "$OUT/sq_iq" -o run -- python3 bench.py --workload iq --steps 2 --warmup 1 --no-cpu > /dev/null 2>> "$OUT/bench.err"
python tools/pmc_sq.py "$OUT/sq_iq" > "$OUT/sq_counters_iq.txt"
rm -rf "$OUT/sq_iq"

# the read-streaming ceiling of this box and the ingest stage against it
if [ -x tools/micro/build/stream_big ]; then timeout 120 tools/micro/build/stream_big > "$OUT/stream_ceiling.txt" 2>&1; fi
timeout 600 python tools/ingest_lab.py taps > "$OUT/ingest_probe.txt" 2>> "$OUT/bench.err"

# where the waves' cycles go (SQ counters, one pass)
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d "$OUT/sq" -o run -- python3 bench.py --steps 3 --warmup 1 --no-cpu --no-c4 --no-pcie --no-extras > /dev/null 2>> "$OUT/bench.err"
python tools/pmc_sq.py "$OUT/sq" > "$OUT/sq_counters.txt"
rm -rf "$OUT/sq"
ls -la "$OUT"
