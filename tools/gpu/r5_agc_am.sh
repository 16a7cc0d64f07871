#!/bin/bash
# the AGC verdict with the hang-time test inside the classification's workgroups: AGC tests, then the cs16-am-nrsc5 preset's stage times
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd "$REPO"; mkdir -p gpurun_out/r5_agc_am
timeout -k 10 900 python3 -m pytest tests -x -q -m gpu -k "agc or preset or pack or convert or golden or p0 or format or mid_kernel or fat" > gpurun_out/r5_agc_am/tests.log 2>&1 || { tail -30 gpurun_out/r5_agc_am/tests.log; exit 1; }
tail -3 gpurun_out/r5_agc_am/tests.log
python3 bench.py --only-presets --steps 10 --warmup 3 --no-cpu-baseline --no-host-leg --no-extra 2>/dev/null | python3 -c "
import sys,json; d=json.loads(sys.stdin.read()); p=d['secondary']['presets']
for k, v in p.items(): print(k, v.get('ms_per_step'), v.get('frac'), v.get('kernels'))" | tee gpurun_out/r5_agc_am/presets.txt
