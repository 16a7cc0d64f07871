cd $GRAFT_REPO_ROOT
# the same work PER CU on fewer CUs: does a CU run faster when fewer of them work?  (k_front_mid, 2^20 frames per CU)
for cfg in "256 28" "192 0" "128 27" "64 26" "32 25"; do
  set -- $cfg; cus=$1; lg=$2
  if [ "$lg" = 0 ]; then continue; fi
  for i in 1 2; do
  IQGPU_CUS=$cus IQGPU_FORCE_FAT=1 timeout -k 10 200 python3 bench.py --log2-frames $lg --steps 40 --warmup 5 --settle-seconds 2 --no-cpu-baseline --no-host-leg --no-secondary --no-extra 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('CUs $cus frames 2^$lg', d['roofline']['kernel'].split()[0], 'kernel ms', d['roofline']['kernel_ms'], 'GS/s per CU', round(d['value']/1e3/$cus*(d['ms_per_step']/d['roofline']['kernel_ms']),3))"
  done
done
