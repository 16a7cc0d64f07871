#!/bin/bash
# PMC counters of the headline kernel, k_front_fat and (IQGPU_NO_FAT=1) k_front_s1, same box: tools/pmc_front.sh <tag>
# one rocprofv3 --pmc pass per counter set (never combined with tracing); summaries in gpurun_out/pmc_<tag>/summary.txt
TAG=${1:-x}
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/pmc_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
BENCH="python3 $REPO/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-host-leg --no-secondary --no-extra --settle-seconds 0"
SETS=("SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES" "SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" "GRBM_GUI_ACTIVE SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY" "SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_BUSY_CYCLES")
for v in ${PMC_VARIANTS:-mid fat s1}; do
  unset IQGPU_NO_FAT IQGPU_FAT; [ $v = s1 ] && export IQGPU_NO_FAT=1; [ $v = fat ] && export IQGPU_FAT=1
  i=0
  for set in "${SETS[@]}"; do
    i=$((i+1))
    rocprofv3 --pmc $set -d "$OUT/${v}_$i" -o pmc --output-format csv -- $BENCH > "$OUT/${v}_$i.log" 2>&1
  done
done
unset IQGPU_NO_FAT IQGPU_FAT
cd "$REPO"
{
  for v in ${PMC_VARIANTS:-mid fat s1}; do
    echo "## variant $v"
    for p in "$OUT"/${v}_*/; do
      f=$(find "$p" -name '*counter_collection.csv' | head -1)
      [ -n "$f" ] && python3 tools/pmc_summary.py "$f" k_front
    done
  done
} > "$OUT/summary.txt"
find "$OUT" -name '*.csv' -size +1M -delete
cat "$OUT/summary.txt"
