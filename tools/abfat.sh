#!/bin/bash
# same-box A/B of k_front_fat against k_front_s1 on the bench.py line: one library, IQGPU_NO_FAT=1 selects the 16-wave kernel
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd "$REPO"; mkdir -p gpurun_out/ab
for i in 1 2 3; do
  for v in s1 fat; do
    E=""; [ $v = s1 ] && E="IQGPU_NO_FAT=1"
    env $E python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-host-leg --no-secondary --no-extra $ABN_ARGS 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v', d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['frac'])"
  done
done | tee gpurun_out/ab/abfat.txt
