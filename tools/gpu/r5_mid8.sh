#!/bin/bash
# k_front_mid on 8-bit frames against k_front_s1 (IQGPU_NO_MID_8BIT=1): 2^28 frames, device-resident, both ways on one box
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd "$REPO"; mkdir -p gpurun_out/r5_mid8
B="python3 tools/bench_chain.py --log2-frames 28 --steps 40"
for i in 1 2; do
  for e in "" "IQGPU_NO_MID_8BIT=1"; do
    for args in "--in-format cs16 --out-format cs16 --shift 200e3" "--in-format cu8 --out-format cu8 --out-rate 744187.5 --shift 200e3" "--in-format cu8 --out-format cu8 --out-rate 744187.5" "--in-format cs8 --out-format cs8 --out-rate 744187.5" "--in-format cu8 --out-format cs16 --out-rate 744187.5" "--in-format cs16 --out-format cu8 --shift 200e3" "--in-format cu8 --out-format cu8 --out-rate 744187.5 --agc" "--in-format cs16 --out-format cs16 --shift 200e3 --gain 0.5" "--in-format sc16q11 --out-format cs16 --shift 200e3"; do
      echo "[${e:-mid}] $(env $e $B $args 2>&1 | grep -v amdgpu.ids)"
    done
  done
done | tee gpurun_out/r5_mid8/out.txt
