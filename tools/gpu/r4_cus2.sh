cd $GRAFT_REPO_ROOT
for cfg in "256 28" "128 27" "64 26" "32 25"; do
  set -- $cfg
  echo "## IQGPU_CUS=$1 frames 2^$2"
  IQGPU_CUS=$1 IQGPU_CLOCK_LOG2=$2 IQGPU_FORCE_FAT=1 IQGPU_LIB=$PWD/iq_tool_amd/lib/libiqgpu_clock.so timeout -k 10 120 python3 tools/clock.py 2>&1 | grep -E "next 20|one launch"
done
