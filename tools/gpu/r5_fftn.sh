#!/bin/bash
# the -usb / -lsb presets' 97-tap band-pass at other overlap-save sizes (IQGPU_FFT_LOG2N): stage times
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd "$REPO"; mkdir -p gpurun_out/r5_fftn
for lg in 10 11 12 13; do
  IQGPU_FFT_LOG2N=$lg python3 bench.py --only-presets --steps 10 --warmup 3 --no-cpu-baseline --no-host-leg --no-extra 2>/dev/null | python3 -c "
import sys,json; d=json.loads(sys.stdin.read()); p=d['secondary']['presets']
for k, v in p.items():
    if 'usb' in k: print($lg, k, v.get('ms_per_step'), v.get('frac'), v.get('kernels'))"
done | tee gpurun_out/r5_fftn/out.txt
