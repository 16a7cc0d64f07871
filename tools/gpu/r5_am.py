#!/usr/bin/env python3
"""round 5: the cs16-am-nrsc5 preset (S = 5: k_cascade + last stage + digital AGC): per-kernel device time, fused AGC against IQGPU_AGC_NOFUSE"""
import os, sys, time, subprocess, json
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
if len(sys.argv) > 1:
    import iq_tool_amd
    from iq_tool_amd import synth
    from iq_tool_amd.chain import DeviceBuffer
    frames = 1 << 28
    kw = dict(in_format="cs16", out_format="cs16", input_rate_hz=2.4e6, target_rate_hz=46511.71875, agc=True)
    ch = iq_tool_amd.Chain(**kw)
    raw = np.tile(synth.raw_stream(1 << 22, 2.4e6, 3, "cs16"), frames >> 22)
    d_in, d_out = DeviceBuffer(raw.nbytes), DeviceBuffer(4 * ch.max_out_frames(frames))
    d_in.upload(raw)
    for _ in range(3):
        ch.process_device(d_in.ptr, frames, d_out.ptr, d_out.nbytes)
    ch.synchronize()
    ch.set_profiling(True); ch.profile()
    t0 = time.perf_counter()
    for _ in range(5):
        ch.process_device(d_in.ptr, frames, d_out.ptr, d_out.nbytes)
    ch.synchronize()
    dt = (time.perf_counter() - t0) / 5
    p = ch.profile()
    print(sys.argv[1], "ms per step %.4f" % (dt * 1e3), {k: round(v["ms"] / 5, 4) for k, v in p.items() if v["launches"]}, ch.agc_state())
else:
    for env in ({}, {"IQGPU_AGC_NOFUSE": "1"}):
        subprocess.run([sys.executable, __file__, "nofuse" if env else "fused"], env=dict(os.environ, **env))
