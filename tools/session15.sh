#!/bin/bash
set -u
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/s15
mkdir -p "$OUT"
cd "$REPO"
timeout 1500 python3 -m pytest tests -x -q -m gpu > "$OUT/pytest.log" 2>&1
tail -4 "$OUT/pytest.log"
python3 tools/bench_chain.py --out-rate 46511.71875 --log2-frames 28 --steps 100 > "$OUT/chain_am.txt" 2>&1
python3 tools/bench_chain.py --out-rate 46511.71875 --log2-frames 28 --steps 100 --agc > "$OUT/chain_am_agc.txt" 2>&1
IQGPU_AGC_NOFUSE=1 python3 tools/bench_chain.py --out-rate 46511.71875 --log2-frames 28 --steps 100 --agc > "$OUT/chain_am_agc_nofuse.txt" 2>&1
cat "$OUT"/chain_*.txt | grep -v amdgpu
