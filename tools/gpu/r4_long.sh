cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4l
timeout -k 10 600 python3 bench.py --steps 50000 --warmup 5 --no-cpu-baseline --no-host-leg --no-secondary --no-extra > gpurun_out/r4l/bench_long.json 2> gpurun_out/r4l/bench_long.err; echo "rc=$?"
python3 -c "
import json; d=json.load(open('gpurun_out/r4l/bench_long.json')); print(d['value'], d['ms_per_step'], d['steps'], d['roofline']['kernel_ms'], d['roofline']['frac'])"
