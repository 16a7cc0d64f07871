#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd "$REPO"; mkdir -p gpurun_out/s25
timeout 1500 python3 -m pytest tests -x -q -m gpu > gpurun_out/s25/pytest.log 2>&1; tail -5 gpurun_out/s25/pytest.log
for i in 1 2; do
    for cfg in 4 3; do
    python3 bench.py --steps 20 --warmup 5 --config $cfg --no-cpu-baseline --no-host-leg 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('cfg$cfg', d['ms_per_step'], d['roofline']['note'])"
    done
done | tee gpurun_out/s25/ab.txt
