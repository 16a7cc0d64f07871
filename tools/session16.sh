#!/bin/bash
set -u
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/s16
mkdir -p "$OUT"
cd "$REPO"
timeout 1500 python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "agc or submit" > "$OUT/pytest.log" 2>&1
tail -4 "$OUT/pytest.log"
python3 tools/bench_chain.py --out-rate 46511.71875 --log2-frames 28 --steps 100 > "$OUT/chain_am.txt" 2>&1
python3 tools/bench_chain.py --out-rate 46511.71875 --log2-frames 28 --steps 100 --agc > "$OUT/chain_am_agc.txt" 2>&1
python3 tools/bench_chain.py --log2-frames 28 --shift 200e3 --agc --steps 200 > "$OUT/chain_nrsc5_agc.txt" 2>&1
cat "$OUT"/chain_*.txt | grep -v amdgpu
