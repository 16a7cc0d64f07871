set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/steal
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "run_stealing or fat_kernel_equals or mid_kernel_random or bytes_unchanged or agc_fused_path" > gpurun_out/steal/tests.log 2>&1; echo "tests rc=$?" | tee -a gpurun_out/steal/tests.log
tail -5 gpurun_out/steal/tests.log
timeout -k 10 600 tools/steal_ab.sh > gpurun_out/steal/ab_all.log 2>&1
cat gpurun_out/steal/ab.txt
cat gpurun_out/steal/timeline.txt | grep -v "^$" | head -60
