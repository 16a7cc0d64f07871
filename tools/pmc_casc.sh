#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/pmc_casc; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for v in base noraw nokt; do
  unset IQGPU_NO_RAW0 IQGPU_NO_KT
  [ $v = noraw ] && export IQGPU_NO_RAW0=1
  [ $v = nokt ] && export IQGPU_NO_KT=1
  rocprofv3 --pmc SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_INSTS_VALU -d $OUT/$v -o pmc --output-format csv -- python3 $REPO/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-host-leg --no-secondary --no-extra --settle-seconds 0 --config 4 > $OUT/$v.log 2>&1
  f=$(find $OUT/$v -name '*counter_collection.csv' | head -1)
  echo "== $v"; python3 $REPO/tools/pmc_summary.py $f k_cascade
done
