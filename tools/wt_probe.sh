#!/bin/bash
# run-weight tuning of k_front_mid (IQGPU_RUN_WEIGHTS=a,b,c: runs of the first / second / third wave of a SIMD): bench kernel time per triple,
# three rounds; `tools/wt_probe.sh timeline "a,b,c"` prints the per-slot end times of one launch from the -DIQGPU_CLOCKSTAMP build instead
cd ${GRAFT_REPO_ROOT:-/root/repo}
if [ "$1" = timeline ]; then
  IQGPU_RUN_WEIGHTS=$2 IQGPU_LIB=$PWD/iq_tool_amd/lib/libiqgpu_clock.so timeout -k 10 200 python3 tools/clock.py 2>&1
  exit 0
fi
for i in 1 2 3; do
for w in "$@"; do
  IQGPU_RUN_WEIGHTS=$w python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-host-leg --no-secondary --no-extra 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('w=$w', d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['frac'])"
done
done
