#!/bin/bash
set -u
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/s14
mkdir -p "$OUT"
cd "$REPO"
IQGPU_FUZZ_SEEDS=4000 timeout 2400 python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "random_chain or agc_random" > "$OUT/soak.log" 2>&1
tail -6 "$OUT/soak.log"
