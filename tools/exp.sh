#!/bin/bash
# Kernel-time A/B on the GPU box: one rocprofv3 --kernel-trace --stats run of the default bench per setting.
#   bash tools/exp.sh '<name filter regex>' 'VAR=a' 'VAR=b VAR2=c' ...      (an empty string = defaults)
# Settings are exported into this shell (never `env` after `--`: see the rocprofv3 note in DESIGN.md section 5).
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
FILTER="$1"; shift
i=0
for setting in "$@"; do
    i=$((i + 1))
    OUT="gpurun_out/exp/$i"
    rm -rf "$OUT"; mkdir -p "$OUT"
    (
        for kv in $setting; do export "$kv"; done
        rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT" -o run -- python3 bench.py --steps 10 --warmup 2 --no-cpu --no-c4 --no-pcie --no-extras $EXP_ARGS > "$OUT/bench.json" 2> "$OUT/err.txt"
    )
    echo "== [$setting]  $(python3 -c "import json,sys; d=json.loads(open('$OUT/bench.json').read().strip().splitlines()[-1]); print('ms_per_step', d['ms_per_step'])" 2>/dev/null)"
    python3 tools/kstats.py "$OUT" "$FILTER"
done
