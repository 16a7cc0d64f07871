cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "next_to_the_thresholds or takes_long_calls" 2>&1 | tail -25 | cut -c1-400
