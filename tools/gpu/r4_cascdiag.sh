cd $GRAFT_REPO_ROOT
# k_cascade (config 4) with parts of its tile left out (diagnostic builds, wrong results): what is the time made of?
ABN_ARGS="--config 4" bash tools/abn.sh new nofma noldsrd neither
