cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4a
# 1. power / clocks while the headline step runs back to back (is the chip power-limited under this kernel?)
( for i in $(seq 1 40); do rocm-smi --showpower --showclocks --showuse 2>/dev/null | grep -E "Power|sclk|mclk|fclk|GPU use" | tr '\n' ' '; echo; sleep 0.5; done > gpurun_out/r4a/smi.txt ) &
SMI=$!
timeout -k 10 300 python3 bench.py --steps 2000 --warmup 5 --no-cpu-baseline --no-host-leg --no-secondary --no-extra > gpurun_out/r4a/bench_long.json 2> gpurun_out/r4a/bench_long.err
wait $SMI
head -c 600 gpurun_out/r4a/bench_long.json; echo
sed -n '1p;10p;20p;30p;40p' gpurun_out/r4a/smi.txt
rocm-smi --showmaxpower --showperflevel 2>/dev/null | grep -v "^=\|^$" | head -8
# 2. the default line
timeout -k 10 900 python3 bench.py > gpurun_out/r4a/bench.json 2> gpurun_out/r4a/bench.err; echo "bench rc=$?"
python3 -c "
import json; d=json.load(open('gpurun_out/r4a/bench.json'))
print('value', d['value'], 'ms', d['ms_per_step'], 'roofline', d['roofline'])
print('host', d['host_end_to_end'])
print('extra', json.dumps(d.get('extra'), indent=1)[:2500])
print('secondary', {k:(v.get('ms_per_step'), v.get('error')) for k,v in (d.get('secondary') or {}).items()})
print('cpu', d.get('cpu_baseline'))
"
# 3. the GPU tests
timeout -k 10 1100 python -m pytest tests -x -q -m gpu > gpurun_out/r4a/gputests.log 2>&1; echo "gpu tests rc=$?"; tail -4 gpurun_out/r4a/gputests.log
