#!/bin/bash
# round 5: what k_front_p0 waits for -- counters of the cu8-nrsc5 preset leg (bench.py --only-presets), one pass per set
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5_p0; mkdir -p $O

BENCH="python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-host-leg --no-extra --only-presets --preset-settle 0 --secondary-steps 3"
i=0
for set in "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES" "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" "SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM_RD SQ_BUSY_CYCLES SQ_INSTS_SALU" "TA_BUSY_avr TA_FLAT_READ_WAVEFRONTS_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum"; do
  i=$((i+1))
  rocprofv3 --pmc $set -d $O/pmc_$i -o pmc --output-format csv -- $BENCH > $O/pmc_$i.log 2>&1
  f=$(find $O/pmc_$i -name '*counter_collection.csv' | head -1)
  echo "## set $i: $set"; [ -n "$f" ] && python3 $R/tools/pmc_summary.py "$f" k_front_p0 | head -12; [ -n "$f" ] || tail -3 $O/pmc_$i.log
done > $O/pmc.txt 2>&1
find $O -name '*.csv' -size +1M -delete
cat $O/pmc.txt
