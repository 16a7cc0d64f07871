#!/bin/bash
set -u
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/s9
mkdir -p "$OUT"
cd "$REPO"
python3 tools/debug_interp.py > "$OUT/debug_interp.txt" 2>&1
timeout 900 python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "agc" > "$OUT/pytest_agc.log" 2>&1
tail -15 "$OUT/pytest_agc.log"
python3 tools/bench_chain.py --log2-frames 28 --shift 200e3 --steps 200 > "$OUT/chain_noagc.txt" 2>&1
python3 tools/bench_chain.py --log2-frames 28 --shift 200e3 --agc --steps 200 > "$OUT/chain_agc.txt" 2>&1
IQGPU_AGC_NOFUSE=1 python3 tools/bench_chain.py --log2-frames 28 --shift 200e3 --agc --steps 200 > "$OUT/chain_agc_nofuse.txt" 2>&1
cat "$OUT/debug_interp.txt" "$OUT/chain_noagc.txt" "$OUT/chain_agc.txt" "$OUT/chain_agc_nofuse.txt"
