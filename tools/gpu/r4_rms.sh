cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/rms
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "agc" > gpurun_out/rms/tests.log 2>&1; echo "tests rc=$?"; tail -25 gpurun_out/rms/tests.log | cut -c1-300
