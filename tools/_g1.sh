cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04a
O=gpurun_out/r04a
timeout 2400 python -m pytest tests -m gpu -q > $O/pytest_all.log 2>&1
tail -5 $O/pytest_all.log
timeout 600 python bench.py --no-c4 --no-cpu --no-pcie > $O/bench_odd2.json 2> $O/err2.log
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r04a/bench_odd2.json").read().strip().splitlines()[-1])
print(d["ms_per_step"])
for k in ("n_plus_1","n_plus_2"):
    g=d["general_length"][k]
    print(k, g["ms_per_step"], g["form"][:80])
PY
