#!/bin/bash
# same-box timing of the multi-stage (k_cascade) shapes for two or more builds: tools/abc.sh head new
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd "$REPO"; mkdir -p gpurun_out/ab
for v in "$@" "$@"; do
  L=iq_tool_amd/lib/libiqgpu_$v.so; [ $v = new ] && L=iq_tool_amd/lib/libiqgpu.so
  export IQGPU_LIB=$REPO/$L
  echo "== $v"
  python3 tools/bench_chain.py --in-format cu8 --out-format cu8 --in-rate 61.44e6 --out-rate 1488375 --log2-frames 28 2>/dev/null
  python3 tools/bench_chain.py --in-format cu8 --out-format cu8 --in-rate 61.44e6 --out-rate 1488375 --shift 1e6 --log2-frames 28 2>/dev/null
  python3 tools/bench_chain.py --in-format cs16 --out-format cs16 --in-rate 20e6 --out-rate 744187.5 --log2-frames 28 2>/dev/null
  python3 tools/bench_chain.py --in-format cs16 --out-format cs16 --in-rate 10e6 --out-rate 744187.5 --shift 1e5 --dc-block --log2-frames 28 2>/dev/null
  python3 tools/bench_chain.py --in-format cs16 --out-format cs16 --in-rate 10e6 --out-rate 2.4e6 --log2-frames 28 2>/dev/null
  python3 tools/bench_chain.py --in-format cf32 --out-format cf32 --in-rate 20e6 --out-rate 744187.5 --log2-frames 27 2>/dev/null
done | tee gpurun_out/ab/abc.txt
