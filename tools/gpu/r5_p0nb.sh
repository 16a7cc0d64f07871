#!/bin/bash
# k_front_p0 build variants (libiqgpu_<v>.so: nb3 = -DIQGPU_P0_NB=3, nts = -DIQGPU_P0_NT_STORE) against the shipped one, the cu8 presets' stage times
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd "$REPO"; mkdir -p gpurun_out/r5_p0nb
for i in 1 2; do
  for v in new ${P0_VARIANT:-nts}; do
    L=iq_tool_amd/lib/libiqgpu_$v.so; [ $v = new ] && L=iq_tool_amd/lib/libiqgpu.so
    IQGPU_LIB=$REPO/$L python3 bench.py --only-presets --steps 10 --warmup 3 --no-cpu-baseline --no-host-leg --no-extra 2>/dev/null | python3 -c "
import sys,json; d=json.loads(sys.stdin.read()); p=d['secondary']['presets']
print('$v', {k: (v.get('ms_per_step'), v.get('kernels', {}).get('front')) for k, v in p.items() if k.startswith('cu8')})"
  done
done | tee gpurun_out/r5_p0nb/out.txt
