cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04a
O=gpurun_out/r04a
timeout 2400 python -m pytest tests/test_sharded.py -m gpu -q -x -k "own_stream or rccl" > $O/pytest_async.log 2>&1
grep -v "^  File \"/usr" $O/pytest_async.log | tail -25
WFX_SHARD_CHUNKS=4 WFX_BENCH_FORCE_DIST=1 timeout 600 python bench.py --workload iq --no-cpu --steps 5 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('iq rccl1 chunks4', d['ms_per_step'], d['config']['start_frame'])"
WFX_SHARD_CHUNKS=4 WFX_COMM_ASYNC=0 WFX_BENCH_FORCE_DIST=1 timeout 600 python bench.py --workload iq --no-cpu --steps 5 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('iq rccl1 chunks4 sync', d['ms_per_step'], d['config']['start_frame'])"
WFX_BENCH_FORCE_DIST=1 timeout 600 python bench.py --workload iq --no-cpu --steps 5 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('iq rccl1 chunks1', d['ms_per_step'], d['config']['start_frame'])"
