#!/bin/bash
set -u
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/s8
mkdir -p "$OUT"
cd "$REPO"
timeout 1500 python3 -m pytest tests -x -q -m gpu > "$OUT/pytest.log" 2>&1
tail -5 "$OUT/pytest.log"
python3 bench.py --steps 20 --warmup 5 > "$OUT/bench.json" 2> "$OUT/bench.err"
python3 tools/bench_hostcall.py > "$OUT/hostcall.txt" 2>&1
python3 tools/bench_chain.py --log2-frames 28 --shift 200e3 > "$OUT/chain_noagc.txt" 2>&1
python3 tools/bench_chain.py --log2-frames 28 --shift 200e3 --agc > "$OUT/chain_agc.txt" 2>&1
cat "$OUT/bench.json" "$OUT/hostcall.txt" "$OUT/chain_noagc.txt" "$OUT/chain_agc.txt"
