#!/bin/bash
# same-box timing of several builds of the library: tools/abn.sh name1 name2 ... (iq_tool_amd/lib/libiqgpu_<name>.so; "new" = libiqgpu.so),
# bench.py default line three times each, alternating.  Timing only: no tests are run.
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd "$REPO"; mkdir -p gpurun_out/ab
for i in 1 2 3; do
  for v in "$@"; do
    L=iq_tool_amd/lib/libiqgpu_$v.so; [ $v = new ] && L=iq_tool_amd/lib/libiqgpu.so
    IQGPU_LIB=$REPO/$L python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-host-leg --no-secondary --no-extra $ABN_ARGS 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v', d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['frac'])"
  done
done | tee gpurun_out/ab/abn.txt
