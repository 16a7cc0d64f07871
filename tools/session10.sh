#!/bin/bash
set -u
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/s10
mkdir -p "$OUT"
cd "$REPO"
timeout 1500 python3 -m pytest tests -x -q -m gpu > "$OUT/pytest.log" 2>&1
tail -8 "$OUT/pytest.log"
python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --config 4 > "$OUT/bench4.json" 2> "$OUT/bench.err"
python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --config 3 > "$OUT/bench3.json" 2>> "$OUT/bench.err"
python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --config preset > "$OUT/bench_preset.json" 2>> "$OUT/bench.err"
python3 tools/bench_chain.py --in-format cu8 --out-format cu8 --in-rate 61.44e6 --out-rate 1488375 --log2-frames 29 --steps 50 > "$OUT/chain_c4.txt" 2>&1
python3 tools/bench_chain.py --in-format cs16 --out-format cs16 --in-rate 8e6 --out-rate 0.9e6 --log2-frames 28 --steps 50 > "$OUT/chain_s3.txt" 2>&1
cat "$OUT"/bench4.json "$OUT"/bench3.json "$OUT"/bench_preset.json "$OUT"/chain_c4.txt "$OUT"/chain_s3.txt
