#!/bin/bash
# round 5: k_cascade2 (two tiles per trip) against k_cascade on BASELINE configs[3] -- parity tests, then same-box timing under IQGPU_NO_CASC2
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd "$REPO"; mkdir -p gpurun_out/r5_casc2
timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "two_tile_trips or cascade or config4 or resample_ratios or preset or random_chain" > gpurun_out/r5_casc2/tests.log 2>&1 || { tail -30 gpurun_out/r5_casc2/tests.log; exit 1; }
tail -3 gpurun_out/r5_casc2/tests.log
for i in 1 2 3; do
  for v in new old; do
    E=""; [ $v = old ] && E="IQGPU_NO_CASC2=1"
    env $E python3 bench.py --config 4 --steps 20 --warmup 5 --no-cpu-baseline --no-host-leg --no-secondary --no-extra 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v', d['ms_per_step'], d['roofline'].get('kernel'), d['roofline']['kernel_ms'], d['roofline']['frac'])"
  done
done | tee gpurun_out/r5_casc2/ab.txt
for i in 1 2; do
  for v in new old; do
    E=""; [ $v = old ] && E="IQGPU_NO_CASC2=1"
    env $E python3 bench.py --only-presets --steps 10 --warmup 3 --no-cpu-baseline --no-host-leg --no-extra 2>/dev/null | python3 -c "
import sys,json; d=json.loads(sys.stdin.read()); p=d['secondary']['presets']
print('$v', {k: (v.get('ms_per_step'), v.get('kernels')) for k, v in p.items() if 'am' in k})"
  done
done | tee gpurun_out/r5_casc2/ab_am.txt
