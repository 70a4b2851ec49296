#!/bin/bash
# Round 5: the streaming ingest kernel (csrc/wfx_ingest.hip) against the tile kernels of rounds 1-4 on ONE box, back to back,
# alternating (A/B/A/B/A/B), on the 60-minute IQ stream.   gpurun -- 'bash tools/r05_ab_ingest.sh <tag>'
TAG=${1:-r05_a}
OUT=$PWD/gpurun_out/$TAG
mkdir -p "$OUT"
line() { python - "$1" "$2" <<'PY'
import json, sys
name, path = sys.argv[1], sys.argv[2]
try:
    d = json.loads(open(path).read().strip().splitlines()[-1])
    k = d["kernels"]
    c = d.get("per_step", {}) if "per_step" in d else {}
    print("%-22s ms %.4f ingest %s stages %s frac %s start %s" % (name, d["ms_per_step"], k.get("polyphase_ingest", {}).get("us_per_step"),
          k.get("polyphase_stages", {}).get("us_per_step"), d["roofline"].get("frac"), d["config"].get("start_frame")))
except Exception as e:
    print(name, "ERR", e)
PY
}
run() { name=$1; shift; env "$@" timeout 600 python bench.py --workload iq --no-cpu --steps 10 --warmup 2 > "$OUT/$name.json" 2> "$OUT/$name.err"; line "$name" "$OUT/$name.json"; }
{
for rep in 1 2 3; do
  run stream_fused_$rep WFX_DUMMY=1
  run tile_$rep WFX_INGEST_TILE=1
done
run stream_unfused WFX_FE_UNFUSED=1
for ni in 2 4 8 32 64; do run stream_ni$ni WFX_INGEST_NI=$ni; done
} 2>&1 | tee "$OUT/summary.txt"
