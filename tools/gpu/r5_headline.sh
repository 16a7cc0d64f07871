#!/bin/bash
# round 5, headline kernel: same-box timing and LDS / VALU counters of the shipped k_front_mid against three diagnostic builds
# (timing only, wrong bytes): lean = nothing fetched a phase ahead; leanng2 = lean + slots 0 and 1 keep their taps (half of the tap
# gather gone); dpphw = the polyphase window by 26 wave_shr moves instead of the LDS round trip.
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r5_headline
bash tools/abn.sh new lean leanng2 dpphw > gpurun_out/r5_headline/abn.txt 2>&1
cat gpurun_out/r5_headline/abn.txt
cd /tmp && export TMPDIR=/tmp
BENCH="python3 $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-host-leg --no-secondary --no-extra --settle-seconds 0"
for v in new lean leanng2 dpphw; do
  L=$GRAFT_REPO_ROOT/iq_tool_amd/lib/libiqgpu_$v.so; [ $v = new ] && L=$GRAFT_REPO_ROOT/iq_tool_amd/lib/libiqgpu.so
  export IQGPU_LIB=$L
  i=0
  for set in "SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_INSTS_VALU" "SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU"; do
    i=$((i+1))
    rocprofv3 --pmc $set -d $GRAFT_REPO_ROOT/gpurun_out/r5_headline/pmc_${v}_$i -o pmc --output-format csv -- $BENCH > $GRAFT_REPO_ROOT/gpurun_out/r5_headline/pmc_${v}_$i.log 2>&1
    f=$(find $GRAFT_REPO_ROOT/gpurun_out/r5_headline/pmc_${v}_$i -name '*counter_collection.csv' | head -1)
    echo "## $v set $i"; [ -n "$f" ] && python3 $GRAFT_REPO_ROOT/tools/pmc_summary.py "$f" k_front
  done
done > $GRAFT_REPO_ROOT/gpurun_out/r5_headline/pmc.txt 2>&1
find $GRAFT_REPO_ROOT/gpurun_out/r5_headline -name '*.csv' -size +1M -delete
tail -40 $GRAFT_REPO_ROOT/gpurun_out/r5_headline/pmc.txt
