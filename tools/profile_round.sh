#!/bin/bash
# Collects the rocprofv3 evidence quoted in DESIGN.md / bench.py for one round (run on the GPU box):
#   tools/profile_round.sh r03   -> gpurun_out/prof_r03/...  and the summaries copied to profiles/
# --pmc passes run on their own (never combined with tracing), one counter set per pass; the program itself
# (python3 bench.py) comes after `--`.  Environment switches are exported BEFORE rocprofv3 (no `env` hop behind `--`).
# Headline config: the three kernels for the preset shape -- k_front_mid (default), k_front_fat (IQGPU_FAT=1), k_front_s1
# (IQGPU_NO_FAT=1) -- then configs 3, 4 and the preset with its AGC.
set -u
TAG=${1:-r03}
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p "$OUT" "$REPO/profiles"
cd /tmp && export TMPDIR=/tmp
BENCH="python3 $REPO/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-host-leg --no-secondary --no-extra --settle-seconds 0"
TIMED="python3 $REPO/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-host-leg --no-secondary --no-extra"     # default settle: the sustained clock
SETS=("SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES" "SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" "GRBM_GUI_ACTIVE SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY" "SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_BUSY_CYCLES")
variant() { unset IQGPU_NO_FAT IQGPU_FAT; [ "$1" = s1 ] && export IQGPU_NO_FAT=1; [ "$1" = fat ] && export IQGPU_FAT=1; }
# ---- kernel-trace statistics
for v in mid fat s1; do
  variant $v
  rocprofv3 --kernel-trace --stats -d "$OUT/stats_$v" -o stats --output-format csv -- $TIMED > "$OUT/stats_$v.log" 2>&1
  echo "stats $v done"
done
variant mid
for cfg in 3 4 preset; do
  rocprofv3 --kernel-trace --stats -d "$OUT/stats_cfg$cfg" -o stats --output-format csv -- $TIMED --config $cfg > "$OUT/stats_cfg$cfg.log" 2>&1
  echo "stats cfg $cfg done"
done
# ---- the seven shipped presets (secondary.presets), and the cs16-fm-nrsc5 preset through submit / collect (verdict on the host)
rocprofv3 --kernel-trace --stats -d "$OUT/stats_presets" -o stats --output-format csv -- python3 $REPO/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-host-leg --no-extra --only-presets > "$OUT/stats_presets.log" 2>&1
echo "stats presets done"
rocprofv3 --kernel-trace --stats -d "$OUT/stats_preset_submit" -o stats --output-format csv -- python3 $REPO/tools/gpu/r5_preset_pipe.py > "$OUT/stats_preset_submit.log" 2>&1
echo "stats preset submit done"
# ---- counters of the headline kernel (PMC_VARIANTS: default all three)
for v in ${PMC_VARIANTS:-mid fat s1}; do
  variant $v
  i=0
  for set in "${SETS[@]}"; do
    i=$((i+1))
    rocprofv3 --pmc $set -d "$OUT/pmc_${v}_$i" -o pmc --output-format csv -- $BENCH > "$OUT/pmc_${v}_$i.log" 2>&1
  done
  echo "pmc $v done"
done
variant mid
# ---- the same counter sets for the kernels of configs 3 and 4
for cfg in 3 4; do
  i=0
  for set in "${SETS[@]}"; do
    i=$((i+1))
    rocprofv3 --pmc $set -d "$OUT/pmc_cfg${cfg}_$i" -o pmc --output-format csv -- $BENCH --config $cfg > "$OUT/pmc_cfg${cfg}_$i.log" 2>&1
  done
  echo "pmc cfg $cfg done"
done
# ---- HBM traffic: FETCH_SIZE and WRITE_SIZE in separate passes, every config
for cfg in 2 3 4 preset; do
  for c in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $c -d "$OUT/hbm_${cfg}_$c" -o pmc --output-format csv -- $BENCH --config $cfg > "$OUT/hbm_${cfg}_$c.log" 2>&1
  done
  echo "hbm $cfg done"
done
cd "$REPO"
for v in mid fat s1; do
  f=$(find "$OUT/stats_$v" -name '*kernel_stats.csv' | head -1)
  sfx="_$v"; [ $v = mid ] && sfx=""
  [ -n "$f" ] && cp "$f" "profiles/${TAG}_kernel_stats$sfx.csv"
done
for cfg in 3 4 preset; do
  f=$(find "$OUT/stats_cfg$cfg" -name '*kernel_stats.csv' | head -1)
  n=config$cfg; [ $cfg = preset ] && n=preset
  [ -n "$f" ] && cp "$f" "profiles/${TAG}_kernel_stats_$n.csv"
done
for n in presets preset_submit; do
  f=$(find "$OUT/stats_$n" -name '*kernel_stats.csv' | head -1)
  [ -n "$f" ] && cp "$f" "profiles/${TAG}_kernel_stats_$n.csv"
done
cp "$OUT/stats_preset_submit.log" "profiles/${TAG}_preset_submit.log" 2>/dev/null
{
  echo "# profiles/${TAG}_pmc_summary.txt -- rocprofv3 --pmc passes of \`$BENCH\` (BASELINE configs[1], 2^28 frames per launch),"
  echo "# one pass per counter set, never combined with tracing (tools/profile_round.sh); per-dispatch averages."
  echo "# Three kernels for the same chain: mid = k_front_mid (default), fat = k_front_fat (IQGPU_FAT=1), s1 = k_front_s1 (IQGPU_NO_FAT=1)."
  for v in ${PMC_VARIANTS:-mid fat s1}; do
    echo "## variant $v"
    for p in "$OUT"/pmc_${v}_*/; do
      f=$(find "$p" -name '*counter_collection.csv' | head -1)
      [ -n "$f" ] && python3 tools/pmc_summary.py "$f" k_front
    done
  done
  for cfg in 3 4; do
    echo "## config $cfg (every iqgpu kernel of the step)"
    for p in "$OUT"/pmc_cfg${cfg}_*/; do
      f=$(find "$p" -name '*counter_collection.csv' | head -1)
      [ -n "$f" ] && python3 tools/pmc_summary.py "$f" iqgpu
    done
  done
  echo "## HBM traffic per config (KiB as rocprofv3 reports them; FETCH_SIZE is doubled per the gfx950 note)"
  for cfg in 2 3 4 preset; do
    for c in FETCH_SIZE WRITE_SIZE; do
      f=$(find "$OUT/hbm_${cfg}_$c" -name '*counter_collection.csv' | head -1)
      [ -n "$f" ] && { echo "### config $cfg $c"; python3 tools/pmc_summary.py "$f" iqgpu; }
    done
  done
} > "profiles/${TAG}_pmc_summary.txt"
python3 tools/traffic_from_pmc.py "$OUT" > "$OUT/traffic.log" 2>&1
mkdir -p "$OUT/profiles" && cp profiles/${TAG}_* profiles/traffic.json "$OUT/profiles/" 2>/dev/null
find "$OUT" -name '*.csv' -size +1M -delete
ls "$OUT/profiles"; cat "$OUT/traffic.log"
