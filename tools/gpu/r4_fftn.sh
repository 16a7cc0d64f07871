cd $GRAFT_REPO_ROOT
# transform size of the overlap-save filter, configs 3 (1025 taps) and 4 (4097 taps): ms per step
for pair in "3 11" "3 12" "3 13" "3 14" "4 13" "4 14"; do
  set -- $pair
  IQGPU_FFT_LOG2N=$2 timeout -k 10 200 python3 bench.py --config $1 --steps 100 --warmup 10 --settle-seconds 1 --no-cpu-baseline --no-host-leg --no-secondary --no-extra 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('config $1 log2n $2', d['ms_per_step'])"
done
