cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4f
timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "more_fixed_runs or block_samples or sixteen_wave or stealing or partition" > gpurun_out/r4f/tests.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -5 gpurun_out/r4f/tests.log; [ $rc = 0 ] || exit 1
timeout -k 10 300 python3 bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-host-leg --no-secondary > gpurun_out/r4f/bench.json 2> gpurun_out/r4f/bench.err; echo "bench rc=$?"
python3 -c "
import json; d=json.loads(open('gpurun_out/r4f/bench.json').read()); print(d['value'], d['ms_per_step'], d['roofline']['kernel_ms']); print(json.dumps(d['extra'].get('block_samples_262144'), indent=1))"
