#!/bin/bash
# counters of k_cascade2 on the cs16-am-nrsc5 preset and on configs[3] (separate --pmc passes, the program itself behind --)
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/r5_casc2_pmc
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
SETS=("SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES" "SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" "GRBM_GUI_ACTIVE SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY")
i=0
for set in "${SETS[@]}"; do
  i=$((i+1))
  rocprofv3 --pmc $set -d "$OUT/am_$i" -o pmc --output-format csv -- python3 $REPO/tools/gpu/r5_am.py fused > "$OUT/am_$i.log" 2>&1
  rocprofv3 --pmc $set -d "$OUT/c4_$i" -o pmc --output-format csv -- python3 $REPO/bench.py --config 4 --steps 5 --warmup 2 --no-cpu-baseline --no-host-leg --no-secondary --no-extra --settle-seconds 0 > "$OUT/c4_$i.log" 2>&1
  echo "set $i done"
done
cd "$REPO"
{
for n in am c4; do
  echo "## $n"
  for p in "$OUT"/${n}_*/; do
    f=$(find "$p" -name '*counter_collection.csv' < /dev/null | head -1)
    [ -n "$f" ] && python3 tools/pmc_summary.py "$f" k_cascade
  done
done
} > "$OUT/summary.txt"
find "$OUT" -name '*.csv' -size +1M -delete
cat "$OUT/summary.txt"
