#!/bin/bash
set -u
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/s12
mkdir -p "$OUT"
cd "$REPO"
python3 -c "import __graft_entry__ as g; g.smoke()" > "$OUT/smoke.txt" 2>&1
python3 bench.py --steps 20 --warmup 5 > "$OUT/bench.json" 2> "$OUT/bench.err"
IQGPU_BENCH_SHARE_GPU=1 python3 bench.py --gpus 2 --steps 10 --warmup 3 --settle-seconds 0.5 --no-host-leg > "$OUT/bench_2ranks_1gpu.json" 2> "$OUT/bench2.err"
IQGPU_BENCH_SHARE_GPU=1 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29555 bench.py --gpus 2 --steps 10 --warmup 3 --settle-seconds 0.5 > "$OUT/bench_torchrun2.json" 2> "$OUT/bench3.err"
timeout 1500 python3 -m pytest tests -x -q -m gpu > "$OUT/pytest.log" 2>&1
tail -4 "$OUT/pytest.log"
cat "$OUT/smoke.txt" | tail -2; cat "$OUT/bench.json" "$OUT/bench_2ranks_1gpu.json" "$OUT/bench_torchrun2.json"; tail -3 "$OUT/bench2.err" "$OUT/bench3.err"
