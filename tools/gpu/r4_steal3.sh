cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/steal
export SETTINGS="IQGPU_STEAL=0
IQGPU_STEAL=1
IQGPU_STEAL=1 IQGPU_STEAL_MIN=12
IQGPU_STEAL=1 IQGPU_STEAL_MIN=4
IQGPU_STEAL=1 IQGPU_STEAL_ROUNDS=8
IQGPU_STEAL=1 IQGPU_RUN_WEIGHTS=0,0,0
IQGPU_STEAL=0 IQGPU_RUN_WEIGHTS=0,0,0"
timeout -k 10 900 tools/steal_ab.sh 2 > gpurun_out/steal/ab_all.log 2>&1
cat gpurun_out/steal/ab.txt
grep -v "^$" gpurun_out/steal/timeline.txt | grep -v "cold"
