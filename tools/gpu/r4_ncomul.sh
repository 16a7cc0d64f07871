cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4n
timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "sixteen_wave or tap_placement or stealing or fixed_runs or block_samples or full_size or takes_long_calls or freq_shift or agc_fused" > gpurun_out/r4n/tests.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -3 gpurun_out/r4n/tests.log; [ $rc = 0 ] || exit 1
bash tools/abn.sh prev new
