#!/bin/bash
# Round 5: the launch forms of bench.py that changed this round, on one box (ranks share the one GPU where there are several):
#   default line (with the e2e objects), 2 ranks over the host-staged transport with BOTH plans of c4_strong timed and every
#   collective timed, the RCCL self-test failing (two ranks on one GPU) -> fallback, one RCCL rank with event-timed collectives.
TAG=${1:-r05_paths}
OUT=$PWD/gpurun_out/$TAG
mkdir -p "$OUT"
t0=$(date +%s.%N); python bench.py > "$OUT/bench.json" 2> "$OUT/bench.err"; echo "default line: rc $? in $(echo "$(date +%s.%N) - $t0" | bc) s"
WFX_BENCH_OVERSUBSCRIBE=1 timeout 900 python bench.py --gpus 2 --steps 3 --warmup 1 --no-pcie --no-cpu > "$OUT/bench_default_shm2.json" 2> "$OUT/bench_default_shm2.err"; echo "shm2 rc $?"
WFX_BENCH_COMM=rccl WFX_BENCH_OVERSUBSCRIBE=1 WFX_BENCH_RCCL_PROBE_S=40 timeout 900 python bench.py --gpus 2 --steps 3 --warmup 1 --no-pcie --no-cpu --no-c4 --no-extras > "$OUT/bench_rccl_probe_fails_shm2.json" 2> "$OUT/bench_rccl_probe_fails_shm2.err"; echo "probe-fail rc $?"
WFX_BENCH_FORCE_DIST=1 python bench.py --shard --no-c4 --no-cpu > "$OUT/bench_shard_rccl1.json" 2> "$OUT/bench_shard_rccl1.err"; echo "rccl1 rc $?"
WFX_BENCH_OVERSUBSCRIBE=1 timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29552 bench.py --gpus 2 --steps 3 --warmup 1 --no-cpu > "$OUT/bench_torchrun_shm2.json" 2> "$OUT/bench_torchrun_shm2.err"; echo "torchrun2 rc $?"
python - "$OUT" <<'PY'
import json, sys, os
out = sys.argv[1]
def last(name):
    try:
        return json.loads(open(os.path.join(out, name)).read().strip().splitlines()[-1])
    except Exception as e:
        return {"_err": str(e)}
d = last("bench.json")
print("default:", d.get("value"), d.get("ms_per_step"), "c4", d.get("c4_strong", {}).get("ms_per_step"), d.get("c4_strong", {}).get("roofline", {}).get("frac"),
      "c3", d.get("c3", {}).get("ms_per_step"), "e2e c2", d.get("e2e", {}).get("ms"), d.get("e2e", {}).get("stages_ms"), "e2e c3", d.get("c3", {}).get("e2e", {}).get("ms"), d.get("c3", {}).get("e2e", {}).get("stages_ms"), d.get("c3", {}).get("e2e", {}).get("png_bytes"))
for n in ("bench_default_shm2.json", "bench_torchrun_shm2.json"):
    d = last(n)
    c4 = d.get("c4_strong", {})
    print(n, d.get("value"), d.get("_err"), "c4:", c4.get("ms_per_step"), c4.get("model_ms"), (c4.get("wire") or {}).get("layout"), "forced:", (c4.get("forced_dist") or {}).get("ms_per_step"), (c4.get("forced_dist") or {}).get("model_ms"), (c4.get("forced_dist") or {}).get("error"), c4.get("error"))
    fw = ((c4.get("forced_dist") or {}).get("wire") or {})
    print("   forced this_rank_us", fw.get("this_rank_us"), (fw.get("this_rank") or [None])[:2])
d = last("bench_rccl_probe_fails_shm2.json")
print("probe-fail:", d.get("value"), d.get("config", {}).get("transport"), d.get("rccl_probe"), d.get("_err"))
d = last("bench_shard_rccl1.json")
w = d.get("wire") or {}
print("rccl1:", d.get("value"), w.get("this_rank_us"), (w.get("this_rank") or [None])[:3], d.get("_err"))
PY
