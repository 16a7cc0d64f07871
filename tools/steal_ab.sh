#!/bin/bash
# round 4: run stealing in k_front_mid -- same-box A/B through bench.py (kernel ms by HIP events, frac of 8 TB/s):
#   static runs (IQGPU_STEAL=0) against stealing under several settings; then the wave timeline of the -DIQGPU_CLOCKSTAMP build
#   (tools/clock.py) with and without stealing.   usage: tools/steal_ab.sh [rounds]  (settings in SETTINGS, one per line)
cd ${GRAFT_REPO_ROOT:-/root/repo}; mkdir -p gpurun_out/steal
one() {
  env "$@" python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-host-leg --no-secondary --no-extra 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$*', d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['frac'])"
}
SETTINGS=${SETTINGS:-"IQGPU_STEAL=0
IQGPU_STEAL=1
IQGPU_STEAL=1 IQGPU_STEAL_STRIDE=1
IQGPU_STEAL=1 IQGPU_STEAL_STRIDE=32
IQGPU_STEAL=1 IQGPU_STEAL_MIN=100000
IQGPU_STEAL=1 IQGPU_STEAL_LANES=16
IQGPU_STEAL=1 IQGPU_STEAL_MIN=12
IQGPU_STEAL=1 IQGPU_RUN_WEIGHTS=0,0,0"}
for i in $(seq 1 ${1:-2}); do
  echo "$SETTINGS" | while read -r line; do [ -n "$line" ] && one $line; done
done 2>&1 | tee gpurun_out/steal/ab.txt
if [ -f iq_tool_amd/lib/libiqgpu_clock.so ]; then
  for s in ${TIMELINES:-"IQGPU_STEAL=0" "IQGPU_STEAL=1"}; do
    echo "## $s"; env ${s//,/ } IQGPU_LIB=$PWD/iq_tool_amd/lib/libiqgpu_clock.so timeout -k 10 200 python3 tools/clock.py 2>&1
  done | tee gpurun_out/steal/timeline.txt
fi
