cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4m
timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "moved_inside or interp or ratios or random_chain" > gpurun_out/r4m/tests.log 2>&1; echo "tests rc=$?"; tail -5 gpurun_out/r4m/tests.log
