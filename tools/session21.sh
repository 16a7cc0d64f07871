#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd "$REPO"; mkdir -p gpurun_out/s21
timeout 900 python3 -m pytest tests -x -q -m gpu -k "submit or harness or probe or service or agc_fused_through" > gpurun_out/s21/pytest.log 2>&1
tail -3 gpurun_out/s21/pytest.log
gcc -O2 -Wall -I include tools/hostcall_bench.c -o tools/hostcall_bench -L iq_tool_amd/lib -liqgpu -Wl,-rpath,$REPO/iq_tool_amd/lib
./tools/hostcall_bench 14 16 18 20 22 24 > gpurun_out/s21/hostcall_c.txt 2>&1
cat gpurun_out/s21/hostcall_c.txt
python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/s21/bench.json 2> gpurun_out/s21/bench.err
python3 -c "import json; d=json.loads(open('gpurun_out/s21/bench.json').read()); print(d['ms_per_step'], d['host_end_to_end'])"
