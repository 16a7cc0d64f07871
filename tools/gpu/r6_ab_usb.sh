cd $GRAFT_REPO_ROOT
for i in 1 2 3 4; do for v in new cf32; do L=iq_tool_amd/lib/libiqgpu.so; [ $v = cf32 ] && L=iq_tool_amd/lib/libiqgpu_cf32.so
IQGPU_LIB=$PWD/$L python3 bench.py --only-presets --presets cs16-fm-nrsc5-usb --secondary-steps 60 --preset-settle 1.0 --no-cpu-baseline --no-host-leg --no-extra 2>/dev/null | python3 -c "
import sys, json
j = json.loads(sys.stdin.read()); v = j['secondary']['presets']['cs16-fm-nrsc5-usb']; print('$v', v['ms_per_step'], v['kernels'])"
done; done
