cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4t
timeout -k 10 500 python3 -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "fft or fir or filter or config3 or config4 or random_chain" > gpurun_out/r4t/tests.log 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/r4t/tests.log
for cfg in 3 4; do
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/r4t/stats$cfg -o stats --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-host-leg --no-secondary --no-extra --config $cfg > $GRAFT_REPO_ROOT/gpurun_out/r4t/bench$cfg.log 2>&1
cd $GRAFT_REPO_ROOT
f=$(find gpurun_out/r4t/stats$cfg -name '*kernel_stats.csv' | head -1); python3 -c "
import csv,sys
for r in list(csv.DictReader(open('$f')))[:5]: print('%-50s %8s %10.1f us'%(r['Name'][:50], r['Calls'], float(r['AverageNs'])/1e3))"
grep -o '"ms_per_step": [0-9.]*' gpurun_out/r4t/bench$cfg.log | head -1
find gpurun_out/r4t/stats$cfg -name '*.csv' -size +1M -delete
done
