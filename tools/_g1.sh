#!/bin/bash
set -u
OUT=$PWD/gpurun_out/r04_v3_sweeps
mkdir -p "$OUT"
timeout 2400 python tools/random_parity.py --cases 600 --seed 77 > "$OUT/random_parity_600.jsonl" 2> "$OUT/err1.txt"; tail -1 "$OUT/random_parity_600.jsonl"
timeout 2400 python tools/random_shard_parity.py --cases 200 --seed 31 > "$OUT/random_shard_parity_200.jsonl" 2> "$OUT/err2.txt"; tail -1 "$OUT/random_shard_parity_200.jsonl"
timeout 3000 python tools/random_fe_parity.py --cases 150 --seed 33 > "$OUT/random_fe_parity_150.jsonl" 2> "$OUT/err3.txt"; tail -1 "$OUT/random_fe_parity_150.jsonl" | cut -c1-400
