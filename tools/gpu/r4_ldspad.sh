cd $GRAFT_REPO_ROOT
# k_fftconv16<12> (config 3) with extra bytes of LDS per workgroup (diagnostic build): 0 -> 5 workgroups per CU, 256 -> 4, 9000 -> 3
for pad in 0 256 9000 0 256; do
  IQGPU_LIB=$GRAFT_REPO_ROOT/iq_tool_amd/lib/libiqgpu_ldspad.so IQGPU_FFT_THREADS=$pad timeout -k 10 200 python3 bench.py --config 3 --steps 100 --warmup 10 --settle-seconds 1 --no-cpu-baseline --no-host-leg --no-secondary --no-extra 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('pad $pad', d['ms_per_step'], d['roofline'].get('note','')[:160])"
done
