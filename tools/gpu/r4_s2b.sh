cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/s2
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "two_stage_chain_in_one or cascade_instantiations or config3 or full_size_secondary or random_chain or dc_block or operator" > gpurun_out/s2/tests.log 2>&1; echo "tests rc=$?"; tail -5 gpurun_out/s2/tests.log
for e in IQGPU_NO_S2=1 IQGPU_NO_S2=0 IQGPU_NO_S2=1 IQGPU_NO_S2=0; do
  env $e timeout -k 10 300 python3 bench.py --config 3 --steps 20 --warmup 5 --no-cpu-baseline --no-host-leg --no-secondary --no-extra 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$e', d['ms_per_step'], d['roofline']['note'])"
done
