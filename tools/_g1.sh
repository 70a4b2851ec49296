cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04a
O=gpurun_out/r04a
timeout 2400 python -m pytest tests -m gpu -q -x > $O/pytest_all.log 2>&1
grep -v "^  File \"/usr" $O/pytest_all.log | tail -25
