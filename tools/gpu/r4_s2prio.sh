cd $GRAFT_REPO_ROOT
ABN_ARGS="--config 3" bash tools/abn.sh new s2prio
