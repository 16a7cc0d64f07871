#!/bin/bash
# round 6: where k_p0fft16's time goes -- the shipped kernel against its two diagnostic builds (window fill alone, transforms alone)
# and against the two-kernel path, on one box.  cu8-nrsc5-usb preset, 2^28 frames per step.
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd "$REPO"; O=gpurun_out/r6; mkdir -p $O
run() {  # name, lib, env...
  local name=$1 lib=$2; shift 2
  env "$@" IQGPU_LIB=$REPO/iq_tool_amd/lib/$lib timeout -k 10 200 python3 bench.py --only-presets --presets cu8-nrsc5-usb --no-cpu-baseline --no-host-leg --no-extra 2>/dev/null | python3 -c "
import sys, json
j = json.loads(sys.stdin.read()); v = j['secondary']['presets']['cu8-nrsc5-usb']
print('$name', v['ms_per_step'], v['front_kernel'], v['kernels'])"
}
for i in 1 2; do
  run two_kernels libiqgpu.so A=1
  run two_kernels_fused_geometry libiqgpu.so IQGPU_FFT_GEOMETRY=keep
  run fused libiqgpu.so IQGPU_FUSE_FILTER=1
  [ -f iq_tool_amd/lib/libiqgpu_p0nofft.so ] && run fill_only libiqgpu_p0nofft.so IQGPU_FUSE_FILTER=1
  [ -f iq_tool_amd/lib/libiqgpu_p0nofill.so ] && run transforms_only libiqgpu_p0nofill.so IQGPU_FUSE_FILTER=1
done 2>&1 | tee $O/fuse_parts.txt
