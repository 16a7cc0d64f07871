#!/bin/bash
set -u
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd "$REPO"
bash tools/profile_round.sh r02 > gpurun_out/profile_r02.log 2>&1
tail -12 gpurun_out/profile_r02.log
