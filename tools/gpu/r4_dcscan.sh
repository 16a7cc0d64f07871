cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4s
timeout -k 10 400 python3 -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "dc or config3 or two_stage" > gpurun_out/r4s/tests.log 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/r4s/tests.log
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/r4s/stats -o stats --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-host-leg --no-secondary --no-extra --config 3 > $GRAFT_REPO_ROOT/gpurun_out/r4s/bench3.log 2>&1
cd $GRAFT_REPO_ROOT
f=$(find gpurun_out/r4s/stats -name '*kernel_stats.csv' | head -1); python3 -c "
import csv,sys
for r in list(csv.DictReader(open('$f')))[:6]: print('%-50s %8s %10.1f us'%(r['Name'][:50], r['Calls'], float(r['AverageNs'])/1e3))"
tail -c 600 gpurun_out/r4s/bench3.log
find gpurun_out/r4s/stats -name '*.csv' -size +1M -delete
