#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd "$REPO"; mkdir -p gpurun_out/s25
timeout 1500 python3 -m pytest tests -x -q -m gpu > gpurun_out/s25/pytest.log 2>&1; tail -3 gpurun_out/s25/pytest.log
python3 -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
for cfg in 4 3 preset 2; do
    python3 bench.py --steps 20 --warmup 5 --config $cfg --no-cpu-baseline --no-host-leg 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('cfg$cfg', d['ms_per_step'], d['value'], d['roofline'].get('note'), d['roofline']['frac'])"
done | tee gpurun_out/s25/ab.txt
