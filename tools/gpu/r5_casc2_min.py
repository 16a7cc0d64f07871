#!/usr/bin/env python3
"""round 5: from which call length do k_cascade2's two-tile trips pay?  BASELINE configs[3]'s resampler (cu8 61.44 -> 1.488 MS/s, K = 4,
no filter) and the cs16-am-nrsc5 one at 2^21 .. 2^27 frames per call, device-resident: k_cascade (IQGPU_NO_CASC2=1) against
k_cascade2 forced from two tiles per run on (IQGPU_CASC2_MIN_RUN=2).  One child process per setting (the switches are read once)."""
import os, sys, time, subprocess
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
if len(sys.argv) > 1:
    import iq_tool_amd
    from iq_tool_amd import synth
    from iq_tool_amd.chain import DeviceBuffer
    for fmt, rate, target in (("cu8", 61.44e6, 1488375.0), ("cs16", 2.4e6, 46511.71875), ("cs8", 20e6, 400e3)):
        bpf = 2 if fmt in ("cu8", "cs8") else 4
        raw = np.tile(synth.raw_stream(1 << 21, rate, 3, fmt), 1 << 6)
        d_in = DeviceBuffer(raw.nbytes); d_in.upload(raw)
        for lg in (range(21, 28) if fmt != "cs8" else (24, 27)):
            n = 1 << lg
            ch = iq_tool_amd.Chain(in_format=fmt, out_format=fmt, input_rate_hz=rate, target_rate_hz=target)
            d_out = DeviceBuffer(ch.out_bytes * (ch.max_out_frames(n) + 64))
            reps = max(4, (1 << 29) >> lg)
            for _ in range(reps):
                ch.process_device(d_in.ptr, n, d_out.ptr, d_out.nbytes)
            ch.synchronize()
            ch.set_profiling(True); ch.profile()
            for _ in range(reps):
                ch.process_device(d_in.ptr, n, d_out.ptr, d_out.nbytes)
            p = ch.profile()
            print(sys.argv[1], fmt, "2^%d" % lg, ch.front_kernel(), "cascade us per call %.2f" % (p["cascade"]["ms"] / reps * 1e3), flush=True)
            ch.close()
else:
    for name, env in (("one-tile", {"IQGPU_NO_CASC2": "1"}), ("two-tile", {"IQGPU_CASC2_MIN_RUN": "2"})):
        subprocess.run([sys.executable, __file__, name], env=dict(os.environ, **env))
