#!/bin/bash
set -u
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/s19
mkdir -p "$OUT"
cd "$REPO"
timeout 900 python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "agc or submit" > "$OUT/pytest.log" 2>&1
tail -3 "$OUT/pytest.log"
python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --config preset > "$OUT/bench_preset.json" 2> "$OUT/bench.err"
python3 -c "import json; d=json.loads(open('$OUT/bench_preset.json').read()); print(d['ms_per_step'], d['config']['workload'][-60:])"
gcc -O2 -Wall -I include tools/hostcall_bench.c -o tools/hostcall_bench -L iq_tool_amd/lib -liqgpu -Wl,-rpath,$REPO/iq_tool_amd/lib
./tools/hostcall_bench 14 16 18 20 22 > "$OUT/hostcall_c.txt" 2>&1
python3 tools/bench_hostcall.py > "$OUT/hostcall_py.txt" 2>&1
cat "$OUT/hostcall_c.txt" "$OUT/hostcall_py.txt"
