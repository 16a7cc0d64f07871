#!/bin/bash
# same-box timing of library builds under an environment switch: tools/abenv.sh "ENV=1" name1 name2 ...
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd "$REPO"; mkdir -p gpurun_out/ab
E=$1; shift
for i in 1 2; do
  for v in "$@"; do
    L=iq_tool_amd/lib/libiqgpu_$v.so; [ $v = new ] && L=iq_tool_amd/lib/libiqgpu.so
    env $E IQGPU_LIB=$REPO/$L python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-host-leg --no-secondary --no-extra $ABN_ARGS 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v', d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['frac'])"
  done
done | tee gpurun_out/ab/abenv.txt
