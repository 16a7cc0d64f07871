#!/bin/bash
set -u
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/s5
mkdir -p "$OUT"
cd "$REPO"
python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline > "$OUT/bench.json" 2> "$OUT/bench.err"
IQGPU_LIB=$REPO/iq_tool_amd/lib/libiqgpu_clock.so python3 tools/clock.py > "$OUT/clock.txt" 2>&1
python3 tools/bench_hostcall.py > "$OUT/hostcall.txt" 2>&1
timeout 1500 python3 -m pytest tests -x -q -m gpu > "$OUT/pytest.log" 2>&1
tail -5 "$OUT/pytest.log"
cat "$OUT/bench.json" "$OUT/clock.txt" "$OUT/hostcall.txt"
