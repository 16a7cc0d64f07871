cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4d
for i in 1 2 3; do timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "submit_collect or two_stage_chain or config3 or partition or block_samples" 2>&1 | tail -2; done
timeout -k 10 700 python -m pytest tests -x -q -m gpu > gpurun_out/r4d/gputests.log 2>&1; echo "gpu tests rc=$?"; tail -4 gpurun_out/r4d/gputests.log
