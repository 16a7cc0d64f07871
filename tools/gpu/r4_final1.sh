cd $GRAFT_REPO_ROOT
export PMC_VARIANTS="mid"
timeout -k 10 1150 bash tools/profile_round.sh r04 > gpurun_out/prof_r04.log 2>&1; echo "rc=$?"; tail -6 gpurun_out/prof_r04.log | cut -c1-300
