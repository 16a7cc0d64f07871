cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4c
timeout -k 10 500 python3 bench.py > gpurun_out/r4c/bench.json 2> gpurun_out/r4c/bench.err; echo "bench rc=$?"
python3 -c "
import json; d=json.load(open('gpurun_out/r4c/bench.json'))
print('value', d['value'], 'ms', d['ms_per_step'], 'roofline', {k:d['roofline'][k] for k in ('frac','kernel_ms','traffic','kernel')})
print('host', d['host_end_to_end']['value'], d['extra']['host_end_to_end_stub']['value'], d['extra']['host_end_to_end_stub']['kernel'])
print('blk', d['extra']['block_samples_262144'])
print('secondary', {k:(v.get('ms_per_step'), v.get('traffic_over_algorithmic'), v.get('error')) for k,v in (d.get('secondary') or {}).items()})
print('devices', d['config']['devices'])
print('cpu', d['cpu_baseline']['value'], d['cpu_baseline']['cores'])
"
timeout -k 10 60 python3 -c "import __graft_entry__ as g; g.smoke()"
timeout -k 10 600 python -m pytest tests -x -q -m gpu > gpurun_out/r4c/gputests.log 2>&1; echo "gpu tests rc=$?"; tail -4 gpurun_out/r4c/gputests.log
