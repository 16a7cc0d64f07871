#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd "$REPO"; mkdir -p gpurun_out/s22
timeout 1500 python3 -m pytest tests -x -q -m gpu > gpurun_out/s22/pytest.log 2>&1
tail -3 gpurun_out/s22/pytest.log
python3 -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
python3 bench.py --steps 20 --warmup 5 > gpurun_out/s22/bench.json 2> gpurun_out/s22/bench.err
cat gpurun_out/s22/bench.json
