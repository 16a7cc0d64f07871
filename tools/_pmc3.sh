OUT=$PWD/gpurun_out/pmc3; mkdir -p $OUT
REPO=$PWD
cd /tmp && export TMPDIR=/tmp
for set in "SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_INST_LDS" "SQ_INSTS_SALU SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_INSTS_VMEM_RD SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY"; do
  name=$(echo $set | tr ' ' '_' | cut -c1-30)
  rocprofv3 --pmc $set -d $OUT/$name -o pmc --output-format csv -- python3 $REPO/bench.py --config 3 --steps 3 --warmup 1 --no-cpu-baseline > $OUT/$name.log 2>&1
  python3 $REPO/tools/pmc_summary.py $(find $OUT/$name -name '*counter_collection.csv' | head -1) k_fftconv
done
