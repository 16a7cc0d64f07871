cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/steal
timeout -k 10 900 tools/steal_ab.sh 2 > gpurun_out/steal/ab_all.log 2>&1
cat gpurun_out/steal/ab.txt
grep -v "^$" gpurun_out/steal/timeline.txt | grep -v "XCD\|cold"
