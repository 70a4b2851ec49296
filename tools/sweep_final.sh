#!/bin/bash
# Large parity sweeps at the final tree (profiles/r04_v4): fresh seeds, against the oracle.
set -u
OUT=$PWD/gpurun_out/r04_v4_sweeps
mkdir -p "$OUT"
timeout 3000 python tools/random_shard_parity.py --cases 400 --seed 91 > "$OUT/random_shard_parity_400.jsonl" 2> "$OUT/err2.txt"; tail -1 "$OUT/random_shard_parity_400.jsonl"
timeout 3000 python tools/random_parity.py --cases 500 --seed 202 > "$OUT/random_parity_500.jsonl" 2> "$OUT/err1.txt"; tail -1 "$OUT/random_parity_500.jsonl"
