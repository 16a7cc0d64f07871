#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd "$REPO"; mkdir -p gpurun_out/s23
for i in 1 2 3; do
  for v in head new; do
    L=iq_tool_amd/lib/libiqgpu.so; [ $v = head ] && L=iq_tool_amd/lib/libiqgpu_head.so
    IQGPU_LIB=$REPO/$L python3 bench.py --steps 20 --warmup 5 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v', d['ms_per_step'], d['roofline']['frac'])"
  done
done | tee gpurun_out/s23/ab.txt
timeout 1500 python3 -m pytest tests -x -q -m gpu > gpurun_out/s23/pytest.log 2>&1
tail -3 gpurun_out/s23/pytest.log
