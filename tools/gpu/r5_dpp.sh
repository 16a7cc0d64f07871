#!/bin/bash
# round 5: k_front_mid with both windows by wave_shr moves (kDpp) against the LDS round trips (-DIQGPU_MID_NO_DPP) and against the
# polyphase window alone (-DIQGPU_MID_DPP_H_ONLY): parity, then same-box timing
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5_dpp
timeout -k 10 900 python3 -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "sixteen_wave or tap_placement or stealing or fixed_runs or eight_per_lane or full_size_output or takes_long_calls or agc_fused_path or agc_verdict" > gpurun_out/r5_dpp/tests.log 2>&1; rc=$?; echo "tests rc=$rc"; tail -5 gpurun_out/r5_dpp/tests.log; [ $rc = 0 ] || exit 1
bash tools/abn.sh nodpp dpph new | tee gpurun_out/r5_dpp/abn.txt
