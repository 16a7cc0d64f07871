#!/bin/bash
# A/B of two builds of the library on one box: iq_tool_amd/lib/libiqgpu_head.so against iq_tool_amd/lib/libiqgpu.so,
# bench.py default line three times each, alternating; then the GPU test suite on the new build.
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd "$REPO"; mkdir -p gpurun_out/ab
for i in 1 2 3; do
  for v in head new; do
    L=iq_tool_amd/lib/libiqgpu.so; [ $v = head ] && L=iq_tool_amd/lib/libiqgpu_head.so
    IQGPU_LIB=$REPO/$L python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-host-leg --no-secondary --no-extra 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v', d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['frac'])"
  done
done | tee gpurun_out/ab/ab.txt
timeout 1500 python3 -m pytest tests -x -q -m gpu > gpurun_out/ab/pytest.log 2>&1
tail -3 gpurun_out/ab/pytest.log
