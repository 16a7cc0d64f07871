#!/bin/bash
# same-box timing of library builds on the presets and configs[2] / [3]: tools/gpu/r5_presets_ab.sh name1 name2 ... ("new" = libiqgpu.so)
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd "$REPO"; mkdir -p gpurun_out/r5_presets_ab
for i in 1 2; do
  for v in "$@"; do
    L=iq_tool_amd/lib/libiqgpu_$v.so; [ $v = new ] && L=iq_tool_amd/lib/libiqgpu.so
    IQGPU_LIB=$REPO/$L python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-host-leg --no-extra 2>/dev/null | python3 -c "
import sys,json; d=json.loads(sys.stdin.read()); s=d['secondary']
print('$v', {k: s[k].get('ms_per_step') for k in ('config3','config4','preset')}, {k: (v.get('ms_per_step'), v.get('kernels', {}).get('filter')) for k, v in s['presets'].items()})"
  done
done | tee gpurun_out/r5_presets_ab/out.txt
