#!/bin/bash
# k_front_p0 on the step classes below 1.6 (late round 5): parity, then a 2.048 MS/s cu8 capture to 1.488375 MS/s against k_front_s1<S0> (IQGPU_NO_P0=1)
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd "$REPO"; mkdir -p gpurun_out/r5_p0cls
timeout -k 10 700 python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "p0_kernel" > gpurun_out/r5_p0cls/tests.log 2>&1 || { tail -30 gpurun_out/r5_p0cls/tests.log; exit 1; }
tail -2 gpurun_out/r5_p0cls/tests.log
B="python3 tools/bench_chain.py --log2-frames 28 --steps 40 --in-format cu8 --out-format cu8"
for i in 1 2; do
  for e in "" "IQGPU_NO_P0=1"; do
    for args in "--in-rate 2.4e6 --out-rate 1488375" "--in-rate 2.048e6 --out-rate 1488375" "--in-rate 1.8e6 --out-rate 1488375 --agc" "--in-rate 2.4e6 --out-rate 2.0e6"; do
      echo "[${e:-p0}] $(env $e $B $args 2>&1 | grep -v amdgpu.ids)"
    done
  done
done | tee gpurun_out/r5_p0cls/out.txt
