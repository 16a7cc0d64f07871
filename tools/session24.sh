#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd "$REPO"; mkdir -p gpurun_out/s24
gcc -O2 -I include tools/hostcall_bench.c -o /tmp/hostcall_bench -L iq_tool_amd/lib -liqgpu -Wl,-rpath,$REPO/iq_tool_amd/lib
/tmp/hostcall_bench 14 16 18 19 20 22 24 2>&1 | tee gpurun_out/s24/hostcall.txt
python3 tools/bench_hostcall.py 2>&1 | tee gpurun_out/s24/hostcall_py.txt
timeout 1500 python3 -m pytest tests -x -q -m gpu > gpurun_out/s24/pytest.log 2>&1; tail -3 gpurun_out/s24/pytest.log
python3 bench.py --steps 20 --warmup 5 2>/dev/null | tee gpurun_out/s24/bench.json
