cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4g
timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "agc" > gpurun_out/r4g/tests.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -3 gpurun_out/r4g/tests.log; [ $rc = 0 ] || exit 1
for i in 1 2 3; do timeout -k 10 200 python3 bench.py --config preset --steps 200 --warmup 10 --settle-seconds 1 --no-cpu-baseline --no-host-leg --no-secondary --no-extra 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('preset', d['ms_per_step'], d['roofline']['kernel_ms'])"; done
