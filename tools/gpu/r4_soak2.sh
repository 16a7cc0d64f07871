cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/soak
export IQGPU_FUZZ_SEEDS=400
timeout -k 10 1100 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "two_stage_random" > gpurun_out/soak/soak2.log 2>&1; echo "soak rc=$?"; tail -12 gpurun_out/soak/soak2.log | cut -c1-400
