#!/bin/bash
# round 5: kernel-trace statistics of the seven shipped presets (bench.py --only-presets) and of the cs16-fm-nrsc5 preset through
# submit / collect (two kernels per batch: the verdict is read on the host) -- the two passes tools/profile_round.sh r05 gained late
set -u
REPO=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$REPO/gpurun_out/prof_r05b; mkdir -p "$OUT/profiles"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d "$OUT/stats_presets" -o stats --output-format csv -- python3 $REPO/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-host-leg --no-extra --only-presets > "$OUT/stats_presets.log" 2>&1
echo "stats presets done"
rocprofv3 --kernel-trace --stats -d "$OUT/stats_preset_submit" -o stats --output-format csv -- python3 $REPO/tools/gpu/r5_preset_pipe.py > "$OUT/stats_preset_submit.log" 2>&1
echo "stats preset submit done"
for n in presets preset_submit; do
  f=$(find "$OUT/stats_$n" -name '*kernel_stats.csv' | head -1)
  [ -n "$f" ] && cp "$f" "$OUT/profiles/r05_kernel_stats_$n.csv"
done
grep "preset through" "$OUT/stats_preset_submit.log" > "$OUT/profiles/r05_preset_submit.log"
find "$OUT" -name '*.csv' -size +1M -delete
cut -c1-140 "$OUT/profiles/r05_kernel_stats_preset_submit.csv" < /dev/null | head -8; cat "$OUT/profiles/r05_preset_submit.log"
