cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/casc
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "cascade or config4 or full_size_secondary or random_chain or block_samples or partition" > gpurun_out/casc/tests.log 2>&1; echo "tests rc=$?"; tail -8 gpurun_out/casc/tests.log | cut -c1-300
for e in IQGPU_NO_REG01=1 IQGPU_NO_REG01=0 IQGPU_NO_REG01=1 IQGPU_NO_REG01=0; do
  env $e timeout -k 10 300 python3 bench.py --config 4 --steps 20 --warmup 5 --no-cpu-baseline --no-host-leg --no-secondary --no-extra 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$e', d['ms_per_step'], d['roofline']['note'])"
done
