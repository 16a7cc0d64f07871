cd $GRAFT_REPO_ROOT
timeout -k 10 400 python3 tools/gpu/s2_parts.py
