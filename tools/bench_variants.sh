mkdir -p gpurun_out/r02
run() { name=$1; shift; timeout 600 "$@" > gpurun_out/r02/$name.json 2> gpurun_out/r02/$name.err; echo "$name rc=$?"; python - "$name" <<'PY'
import json,sys
n=sys.argv[1]
try:
    d=json.load(open(f"gpurun_out/r02/{n}.json"))
    k=d.pop("kernels",{}); c4=d.pop("c4_strong",None)
    print(n, "ms", d["ms_per_step"], "value", d["value"], "rccl", d["config"].get("ranks_rccl"), "roof", d.get("roofline",{}).get("kernel"), d.get("roofline",{}).get("frac"), "start", d["config"].get("start_frame"), d.get("parity_vs_oracle"))
    print("   ", {a:b["us_per_step"] for a,b in k.items()})
except Exception as e:
    print(n, "ERR", e); print(open(f"gpurun_out/r02/{n}.err").read()[-800:])
PY
}
WFX_BENCH_FORCE_DIST=1 run b_shard_rccl1 python bench.py --shard --no-c4 --steps 20
run b_shard_local1 python bench.py --shard --no-c4 --no-cpu --steps 20
WFX_BENCH_FORCE_DIST=1 run b_iq450_rccl1 python bench.py --workload iq --iq-seconds 450 --no-cpu --steps 10
run b_iq450_fused python bench.py --workload iq --iq-seconds 450 --no-cpu --steps 10
run b_iq3600_sharded1 python bench.py --workload iq --iq-form sharded --no-cpu --steps 5
run b_c3 python bench.py --workload c3 --no-cpu --steps 5
WFX_BENCH_FORCE_DIST=1 run b_c3_rccl1 python bench.py --workload c3 --no-cpu --steps 5
