#!/bin/bash
# same-box A/B of two builds of libiqgpu: tools/ab.sh <libA> <libB> [bench args]
A=$1; B=$2; shift 2
for rep in 1 2 3; do
  for L in "$A" "$B"; do
    IQGPU_LIB=$L python bench.py --steps 10 --no-cpu-baseline "$@" 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$L', d['ms_per_step'], d['roofline']['kernel_ms'])"
  done
done
