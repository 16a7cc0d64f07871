cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4b
timeout -k 10 1150 python -m pytest tests -x -q -m gpu > gpurun_out/r4b/gputests.log 2>&1; echo "gpu tests rc=$?"; tail -6 gpurun_out/r4b/gputests.log
