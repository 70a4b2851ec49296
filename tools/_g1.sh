cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_sharded.py tests/test_comm_shm.py -m gpu -q > gpurun_out/g1.log 2>&1
grep -v "^  File \"/usr" gpurun_out/g1.log | tail -40
