#!/bin/bash
# how far the shapes NEXT to the headline chain fall from it (2^28 frames, device-resident): other input formats, a gain, a dc blocker, other output formats
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd "$REPO"; mkdir -p gpurun_out/r5_near
B="python3 tools/bench_chain.py --log2-frames 28 --steps 20"
{
$B --in-format cs16 --out-format cs16 --shift 200e3
$B --in-format cs16 --out-format cs16
$B --in-format cu8 --out-format cu8 --out-rate 744187.5 --shift 200e3
$B --in-format cu8 --out-format cu8 --out-rate 744187.5
$B --in-format cs8 --out-format cs8 --out-rate 744187.5
$B --in-format cu8 --out-format cs16 --out-rate 744187.5
$B --in-format cs16 --out-format cf32 --shift 200e3
$B --in-format cs16 --out-format cu8 --shift 200e3
$B --in-format cs16 --out-format cs16 --shift 200e3 --dc-block
$B --in-format cs16 --out-format cs16 --out-rate 600e3 --shift 200e3
$B --in-format cs16 --out-format cs16 --out-rate 1.0e6 --shift 200e3
$B --in-format cf32 --out-format cf32 --shift 200e3
} 2>&1 | grep -v "^$" | tee gpurun_out/r5_near/out.txt
# (late round 5: the dc-blocker chain with its switch set compiled in -- IQGPU_NO_FAST=1 keeps the run-time form)
for e in "" "IQGPU_NO_FAST=1"; do
  echo "[$e] $(env $e python3 tools/bench_chain.py --log2-frames 28 --steps 20 --in-format cs16 --out-format cs16 --shift 200e3 --dc-block 2>&1 | grep -v amdgpu.ids)" | tee -a gpurun_out/r5_near/out.txt
  echo "[$e] $(env $e python3 tools/bench_chain.py --log2-frames 28 --steps 20 --in-format cs16 --out-format cs16 --dc-block 2>&1 | grep -v amdgpu.ids)" | tee -a gpurun_out/r5_near/out.txt
done
