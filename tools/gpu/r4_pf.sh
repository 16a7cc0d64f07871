cd $GRAFT_REPO_ROOT
timeout -k 10 300 python3 -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "config4 or cascade or resample_ratios or random_chain" 2>&1 | tail -2
ABN_ARGS="--config 4" bash tools/abn.sh pf1 new pf3 pf4 neither2
