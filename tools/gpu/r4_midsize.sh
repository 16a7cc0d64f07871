cd $GRAFT_REPO_ROOT
# k_front_mid against k_front_s1 on SHORT calls (the size rule keeps calls below 8 tiles per wave on k_front_s1): kernel ms at 2^21 .. 2^25 frames
for lg in 21 22 23 24 25; do
  for e in "IQGPU_NO_FAT=1" "IQGPU_FORCE_FAT=1"; do
    env $e timeout -k 10 200 python3 bench.py --log2-frames $lg --steps 200 --warmup 20 --settle-seconds 1 --no-cpu-baseline --no-host-leg --no-secondary --no-extra 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('2^$lg', '$e', d['roofline']['kernel'].split()[0], d['ms_per_step'], d['roofline']['kernel_ms'])"
  done
done
