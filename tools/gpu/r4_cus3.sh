cd $GRAFT_REPO_ROOT
# the WHOLE 2^28-frame launch planned for fewer CUs: is there a CU count below 256 at which the higher clock outweighs the idle CUs?
for cus in 256 248 240 224 192 160; do
  IQGPU_CUS=$cus timeout -k 10 200 python3 bench.py --steps 40 --warmup 5 --settle-seconds 2 --no-cpu-baseline --no-host-leg --no-secondary --no-extra 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('CUs $cus', 'kernel ms', d['roofline']['kernel_ms'], 'frac', d['roofline']['frac'])"
done
