#!/bin/bash
set -u
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/s6
mkdir -p "$OUT"
cd "$REPO"
for rep in 1 2; do
for v in 0 1; do
  echo "=== steal $v (rep $rep)" >> "$OUT/steal.txt"
  IQGPU_LIB=$REPO/iq_tool_amd/lib/libiqgpu_st$v.so python3 tools/clock.py 2>&1 | grep -E "next 20|one launch|workgroups|XCD" >> "$OUT/steal.txt"
  IQGPU_LIB=$REPO/iq_tool_amd/lib/libiqgpu_st$v.so python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-host-leg | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('bench ms_per_step', d['ms_per_step'], 'kernel_ms', d['roofline']['kernel_ms'])" >> "$OUT/steal.txt" 2>&1
done
done
cat "$OUT/steal.txt"
