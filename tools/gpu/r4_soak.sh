cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/soak
export IQGPU_FUZZ_SEEDS=600
timeout -k 10 1100 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "random or schedules or fuzz" > gpurun_out/soak/soak.log 2>&1; echo "soak rc=$?"; tail -6 gpurun_out/soak/soak.log | cut -c1-300
