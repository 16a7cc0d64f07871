#!/bin/bash
# Collects the rocprofv3 evidence quoted in DESIGN.md / bench.py for one round (run on the GPU box):
#   tools/profile_round.sh r02   -> gpurun_out/prof_r02/{stats,pmc*}/...  and the summaries copied to profiles/
# --pmc passes run on their own (never combined with tracing), one counter set per pass; the program itself
# (python3 bench.py) comes after `--`.
set -u
TAG=${1:-r02}
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p "$OUT" "$REPO/profiles"
cd /tmp && export TMPDIR=/tmp
BENCH="python3 $REPO/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-host-leg --no-secondary --settle-seconds 0"
TIMED="python3 $REPO/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-host-leg --no-secondary"     # default settle: the sustained clock
SETS=("FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES" "SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" "SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_WAIT_INST_ANY SQ_BUSY_CYCLES" "GRBM_GUI_ACTIVE SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAVES")
rocprofv3 --kernel-trace --stats -d "$OUT/stats" -o stats --output-format csv -- $TIMED > "$OUT/stats.log" 2>&1
for cfg in 3 4 preset; do
  rocprofv3 --kernel-trace --stats -d "$OUT/stats_cfg$cfg" -o stats --output-format csv -- $TIMED --config $cfg > "$OUT/stats_cfg$cfg.log" 2>&1
done
for cfg in 2 3 4; do
  for set in "${SETS[@]}"; do
    name=$(echo $set | tr ' ' '_' | cut -c1-40)
    rocprofv3 --pmc $set -d "$OUT/pmc${cfg}_$name" -o pmc --output-format csv -- $BENCH --config $cfg > "$OUT/pmc${cfg}_$name.log" 2>&1
  done
done
cd "$REPO"
f=$(find "$OUT/stats_cfgpreset" -name '*kernel_stats.csv' | head -1)
[ -n "$f" ] && cp "$f" "profiles/${TAG}_kernel_stats_preset.csv"
for cfg in 2 3 4; do
  sfx=""; [ $cfg != 2 ] && sfx="_config$cfg"
  d="$OUT/stats"; [ $cfg != 2 ] && d="$OUT/stats_cfg$cfg"
  f=$(find "$d" -name '*kernel_stats.csv' | head -1)
  [ -n "$f" ] && cp "$f" "profiles/${TAG}_kernel_stats$sfx.csv"
  {
    echo "# profiles/${TAG}_pmc_summary$sfx.txt -- rocprofv3 --pmc passes of \`$BENCH --config $cfg\`"
    echo "# (tools/profile_round.sh: one pass per counter set, never combined with tracing); per-dispatch averages."
    echo "# FETCH_SIZE / WRITE_SIZE in KiB as rocprofv3 reports them."
    for p in "$OUT"/pmc${cfg}_*/; do
      f=$(find "$p" -name '*counter_collection.csv' | head -1)
      [ -n "$f" ] && { echo "## $(basename $p)"; python3 tools/pmc_summary.py "$f" iqgpu; echo; }
    done
  } > "profiles/${TAG}_pmc_summary$sfx.txt"
done
ff=$(find "$OUT/pmc2_FETCH_SIZE" -name '*counter_collection.csv' | head -1)
fw=$(find "$OUT/pmc2_WRITE_SIZE" -name '*counter_collection.csv' | head -1)
[ -n "$ff" ] && [ -n "$fw" ] && python3 tools/traffic_from_pmc.py "$ff" "$fw" 28 > "$OUT/traffic.log" 2>&1
mkdir -p "$OUT/profiles" && cp profiles/${TAG}_* profiles/traffic.json "$OUT/profiles/" 2>/dev/null
find "$OUT" -name '*.csv' -size +2M -delete
ls "$OUT" "$OUT/profiles"
