#!/bin/bash
# round 6: k_front_p0 at 12 waves per CU (three per SIMD; -DIQGPU_P0_WAVES=12, one or two frame buffers) against the shipped 8
cd ${GRAFT_REPO_ROOT:-/root/repo}
for i in 1 2 3; do for v in new w16n1; do L=iq_tool_amd/lib/libiqgpu.so; [ $v != new ] && L=iq_tool_amd/lib/libiqgpu_$v.so
[ -f $L ] || continue
IQGPU_LIB=$PWD/$L python3 bench.py --only-presets --presets cu8-nrsc5,cu8-nrsc5-usb --secondary-steps 40 --preset-settle 0.8 --no-cpu-baseline --no-host-leg --no-extra 2>/dev/null | python3 -c "
import sys, json
j = json.loads(sys.stdin.read())
print('$v', ' | '.join('%s %.4f front %.4f' % (k, v['ms_per_step'], v['kernels']['front']) for k, v in j['secondary']['presets'].items()))"
done; done
for v in w16n1; do L=iq_tool_amd/lib/libiqgpu_$v.so; [ -f $L ] || continue
IQGPU_LIB=$PWD/$L timeout -k 10 400 python3 -m pytest tests -m gpu -x -q --timeout 300 -k "p0_kernel_equals or s0_chain" 2>&1 | tail -2; done
