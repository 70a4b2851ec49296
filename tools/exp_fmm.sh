#!/bin/bash
# Headline step on the transform route and on the fast-multipole route, kernel by kernel (rocprofv3 --kernel-trace --stats):
#   bash tools/exp_fmm.sh          (on the GPU box; writes gpurun_out/exp_fmm/{fft,fmm}/)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for mode in fft fmm; do
    OUT="gpurun_out/exp_fmm/$mode"
    rm -rf "$OUT"; mkdir -p "$OUT"
    export WEFAX_HILBERT=$mode
    python3 bench.py --steps 20 --warmup 3 --no-cpu --no-c4 --no-pcie --no-extras --no-e2e > "$OUT/bench_plain.json" 2> "$OUT/err_plain.txt"
    rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT" -o run -- python3 bench.py --steps 10 --warmup 2 --no-cpu --no-c4 --no-pcie --no-extras --no-e2e > "$OUT/bench.json" 2> "$OUT/err.txt"
    echo "== [$mode] ms_per_step $(python3 -c "import json; d=json.loads(open('$OUT/bench_plain.json').read().strip().splitlines()[-1]); print(d['ms_per_step'])" 2>/dev/null)"
    python3 tools/kstats.py "$OUT"
done
