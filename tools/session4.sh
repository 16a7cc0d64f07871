#!/bin/bash
set -u
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/s4
mkdir -p "$OUT"
cd "$REPO"
for rep in 1 2; do
for m in 0 1 2 3 4; do
  echo "=== prio mode $m (rep $rep)" >> "$OUT/prio.txt"
  IQGPU_LIB=$REPO/iq_tool_amd/lib/libiqgpu_p$m.so python3 tools/clock.py 2>&1 | grep -E "next 20|one launch|workgroups" >> "$OUT/prio.txt"
done
done
cat "$OUT/prio.txt"
