cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04a
O=gpurun_out/r04a
timeout 1500 python -m pytest tests/test_polyphase.py tests/test_gpu_configs.py -m gpu -q -x > $O/pytest_fe.log 2>&1
tail -5 $O/pytest_fe.log
WFX_BENCH_FORCE_DIST=1 timeout 600 python bench.py --workload iq --no-cpu --steps 5 > $O/bench_iq_rccl1_cols.json 2> $O/err1.log; head -c 700 $O/bench_iq_rccl1_cols.json; echo
WFX_BENCH_FORCE_DIST=1 timeout 600 python bench.py --workload iq --no-cpu --steps 5 --plan rows > $O/bench_iq_rccl1_rows.json 2>> $O/err1.log; head -c 300 $O/bench_iq_rccl1_rows.json; echo
WFX_BENCH_OVERSUBSCRIBE=1 timeout 900 python bench.py --gpus 8 --workload iq --steps 3 --warmup 1 --no-cpu --plan dist > $O/bench_iq_shm8_cols.json 2>> $O/err1.log; head -c 400 $O/bench_iq_shm8_cols.json; echo
WFX_BENCH_OVERSUBSCRIBE=1 timeout 900 python bench.py --gpus 8 --workload iq --steps 3 --warmup 1 --no-cpu --plan rows > $O/bench_iq_shm8_rows.json 2>> $O/err1.log; head -c 400 $O/bench_iq_shm8_rows.json; echo
tail -5 $O/err1.log
