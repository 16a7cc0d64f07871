#!/bin/bash
# round 5: k_front_p0 (IQGPU_P0=1) against k_front_s1<S0> on the cu8 presets: parity of the opt-in kernel, then the presets legs both ways
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r5_p0
timeout -k 10 300 python3 -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "p0_kernel" > gpurun_out/r5_p0/tests.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -3 gpurun_out/r5_p0/tests.log; [ $rc = 0 ] || exit 1
for v in 0 1; do
  IQGPU_NO_P0=$((1-v)) python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-host-leg --no-extra --only-presets 2>/dev/null > gpurun_out/r5_p0/presets_p0_$v.json
  python3 -c "
import json
d=json.load(open('gpurun_out/r5_p0/presets_p0_$v.json'))
for k,v in d['secondary']['presets'].items():
    if k.startswith('cu8'): print('P0=$v', k, v['ms_per_step'], v['frac'], v['front_kernel'], v['kernels'])"
done
