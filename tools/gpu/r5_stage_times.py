#!/usr/bin/env python3
"""per-stage device times of an arbitrary chain (2^28 frames, device-resident): tools/gpu/r5_stage_times.py cu8 cu8 2.4e6 1488375 [dc] [agc] [shift=200e3]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import iq_tool_amd
from iq_tool_amd import synth
from iq_tool_amd.chain import DeviceBuffer
fi, fo, ri, ro = sys.argv[1], sys.argv[2], float(sys.argv[3]), float(sys.argv[4])
opts = sys.argv[5:]
kw = dict(in_format=fi, out_format=fo, input_rate_hz=ri, target_rate_hz=ro, dc_block="dc" in opts, agc="agc" in opts)
for o in opts:
    if o.startswith("shift="): kw["shift_hz"] = float(o[6:])
frames = 1 << 28
raw = np.tile(synth.raw_stream(1 << 22, ri, 3, fi), frames >> 22)
d_in = DeviceBuffer(raw.nbytes); d_in.upload(raw)
ch = iq_tool_amd.Chain(**kw)
d_out = DeviceBuffer(ch.out_bytes * (ch.max_out_frames(frames) + 64))
t_end = time.perf_counter() + 1.0
while time.perf_counter() < t_end:
    ch.process_device(d_in.ptr, frames, d_out.ptr, d_out.nbytes)
ch.synchronize(); ch.set_profiling(True); ch.profile()
for _ in range(10):
    ch.process_device(d_in.ptr, frames, d_out.ptr, d_out.nbytes)
p = ch.profile()
print(" ".join(sys.argv[1:]), ch.front_kernel(), {k: round(v["ms"] / 10, 4) for k, v in p.items() if v["launches"]})
