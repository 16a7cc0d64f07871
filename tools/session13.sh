#!/bin/bash
set -u
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/s13
mkdir -p "$OUT"
cd "$REPO"
timeout 1500 python3 -m pytest tests -x -q -m gpu > "$OUT/pytest.log" 2>&1
tail -4 "$OUT/pytest.log"
python3 tools/bench_chain.py --log2-frames 28 --agc --steps 200 > "$OUT/chain_cs16_noshift_agc.txt" 2>&1
python3 tools/bench_chain.py --log2-frames 28 --steps 200 > "$OUT/chain_cs16_noshift.txt" 2>&1
python3 tools/bench_chain.py --in-format cu8 --out-format cu8 --out-rate 1488375 --log2-frames 28 --agc --steps 200 > "$OUT/chain_cu8_agc.txt" 2>&1
python3 tools/bench_chain.py --in-format cu8 --out-format cu8 --out-rate 1488375 --log2-frames 28 --steps 200 > "$OUT/chain_cu8.txt" 2>&1
IQGPU_AGC_NOFUSE=1 python3 tools/bench_chain.py --in-format cu8 --out-format cu8 --out-rate 1488375 --log2-frames 28 --agc --steps 200 > "$OUT/chain_cu8_agc_nofuse.txt" 2>&1
cat "$OUT"/chain_*.txt | grep -v amdgpu
