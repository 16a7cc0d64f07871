#!/bin/bash
# k_fftconv16 with the output format chosen once per block: filter tests, presets, configs[2] / [3]
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd "$REPO"; mkdir -p gpurun_out/r5_fftepi
timeout -k 10 900 python3 -m pytest tests -x -q -m gpu -k "fft or filter or fir or config3 or config4 or agc_fused or random_chain or preset" > gpurun_out/r5_fftepi/tests.log 2>&1 || { tail -30 gpurun_out/r5_fftepi/tests.log; exit 1; }
tail -3 gpurun_out/r5_fftepi/tests.log
python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-host-leg --no-extra 2>/dev/null | python3 -c "
import sys,json; d=json.loads(sys.stdin.read()); s=d['secondary']
for k in ('config3','config4','preset'): print(k, s[k].get('ms_per_step'), s[k].get('kernels'))
for k, v in s['presets'].items(): print(k, v.get('ms_per_step'), v.get('frac'), v.get('kernels'))" | tee gpurun_out/r5_fftepi/out.txt
