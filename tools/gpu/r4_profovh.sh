cd $GRAFT_REPO_ROOT
timeout -k 10 300 python3 tools/gpu/prof_overhead.py
