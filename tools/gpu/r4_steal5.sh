cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/steal
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "run_stealing" 2>&1 | tail -2
export SETTINGS="IQGPU_STEAL=0
IQGPU_STEAL=1
IQGPU_STEAL=1 IQGPU_STEAL_LANES=8 IQGPU_STEAL_ROUNDS=8
IQGPU_STEAL=1 IQGPU_STEAL_LANES=32 IQGPU_STEAL_ROUNDS=4
IQGPU_STEAL=1 IQGPU_STEAL_MIN=4 IQGPU_STEAL_ROUNDS=10
IQGPU_STEAL=1 IQGPU_RUN_WEIGHTS=0,0,0
IQGPU_STEAL=0 IQGPU_RUN_WEIGHTS=0,0,0"
export TIMELINES="IQGPU_STEAL=0 IQGPU_STEAL=1 IQGPU_STEAL=1,IQGPU_STEAL_MIN=4,IQGPU_STEAL_ROUNDS=10"
timeout -k 10 900 tools/steal_ab.sh 3 > gpurun_out/steal/ab_all.log 2>&1
cat gpurun_out/steal/ab.txt
grep -v "^$" gpurun_out/steal/timeline.txt | grep -v "cold"
