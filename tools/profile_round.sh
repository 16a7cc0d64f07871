#!/bin/bash
# Collects the rocprofv3 evidence quoted in DESIGN.md / bench.py for one round (run on the GPU box):
#   tools/profile_round.sh r01   -> gpurun_out/prof_r01/{stats,pmc_*}/...
# --pmc passes run on their own (never combined with tracing), one counter set per pass.
set -u
TAG=${1:-r01}
OUT=$PWD/gpurun_out/prof_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
REPO=${GRAFT_REPO_ROOT:-/root/repo}
BENCH="python3 $REPO/bench.py --steps 5 --warmup 2 --no-cpu-baseline"
rocprofv3 --kernel-trace --stats -d "$OUT/stats" -o stats --output-format csv -- $BENCH > "$OUT/stats.log" 2>&1
for cfg in 3 4; do
  rocprofv3 --kernel-trace --stats -d "$OUT/stats_cfg$cfg" -o stats --output-format csv -- $BENCH --config $cfg > "$OUT/stats_cfg$cfg.log" 2>&1
done
for set in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES" "SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" "SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_WAIT_INST_ANY SQ_BUSY_CYCLES"; do
  name=$(echo $set | tr ' ' '_' | cut -c1-40)
  rocprofv3 --pmc $set -d "$OUT/pmc_$name" -o pmc --output-format csv -- $BENCH > "$OUT/pmc_$name.log" 2>&1
done
cd "$REPO"
{
  echo "# rocprofv3 --kernel-trace --stats of: $BENCH (config 2 = BASELINE configs[1])"
  f=$(find "$OUT/stats" -name '*kernel_stats.csv' | head -1); head -6 "$f" | cut -c1-200
  for cfg in 3 4; do echo; echo "# config $cfg"; f=$(find "$OUT/stats_cfg$cfg" -name '*kernel_stats.csv' | head -1); head -8 "$f" | cut -c1-200; done
} > "$OUT/kernel_stats_summary.txt"
{
  for d in "$OUT"/pmc_*/; do
    f=$(find "$d" -name '*counter_collection.csv' | head -1)
    [ -n "$f" ] && { echo "## $(basename $d)"; python3 tools/pmc_summary.py "$f" k_front; echo; }
  done
} > "$OUT/pmc_summary.txt"
ls "$OUT"
