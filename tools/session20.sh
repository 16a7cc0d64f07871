#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd "$REPO"; mkdir -p gpurun_out/s20
gcc -O2 -Wall -I include tools/hostcall_bench.c -o tools/hostcall_bench -L iq_tool_amd/lib -liqgpu -Wl,-rpath,$REPO/iq_tool_amd/lib
./tools/hostcall_bench 14 18 20 > gpurun_out/s20/hostcall_c.txt 2>&1
cat gpurun_out/s20/hostcall_c.txt
