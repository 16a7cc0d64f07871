#!/usr/bin/env python3
"""round 5: the cs16-am-nrsc5 cascade takes 0.237 or 0.279 ms from one bench.py process to the next on the same box.  Placement or clock?
Eight rounds in ONE process: fresh input / output buffers and a fresh chain every round (odd rounds keep the previous buffers)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import iq_tool_amd
from iq_tool_amd import synth
from iq_tool_amd.chain import DeviceBuffer
frames = 1 << 28
raw = np.tile(synth.raw_stream(1 << 22, 2.4e6, 3, "cs16"), frames >> 22)
kw = dict(in_format="cs16", out_format="cs16", input_rate_hz=2.4e6, target_rate_hz=46511.71875, agc=True)
d_in = d_out = None
keep = []
for rnd in range(8):
    if rnd % 2 == 0:
        if d_in is not None: keep.append((d_in, d_out))           # (held: the next buffers land elsewhere)
        d_in = DeviceBuffer(raw.nbytes); d_in.upload(raw)
    ch = iq_tool_amd.Chain(**kw)
    if rnd % 2 == 0:
        d_out = DeviceBuffer(4 * ch.max_out_frames(frames))
    t_end = time.perf_counter() + 0.5
    while time.perf_counter() < t_end:
        ch.process_device(d_in.ptr, frames, d_out.ptr, d_out.nbytes)
    ch.synchronize()
    ch.set_profiling(True); ch.profile()
    for _ in range(20):
        ch.process_device(d_in.ptr, frames, d_out.ptr, d_out.nbytes)
    p = ch.profile()
    print("round", rnd, "in @ %#x" % d_in.ptr, {k: round(v["ms"] / 20, 4) for k, v in p.items() if v["launches"]}, flush=True)
    ch.close()
    if len(keep) > 2: keep.pop(0)
