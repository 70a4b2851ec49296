#!/bin/bash
set -u
OUT=$PWD/gpurun_out/r04_v3_sweeps2
mkdir -p "$OUT"
timeout 3000 python tools/random_parity.py --cases 1000 --seed 101 > "$OUT/random_parity_1000.jsonl" 2> "$OUT/err1.txt"; tail -1 "$OUT/random_parity_1000.jsonl"
timeout 3000 python tools/random_shard_parity.py --cases 300 --seed 55 > "$OUT/random_shard_parity_300.jsonl" 2> "$OUT/err2.txt"; tail -1 "$OUT/random_shard_parity_300.jsonl"
