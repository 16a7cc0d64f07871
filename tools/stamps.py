#!/usr/bin/env python3
"""Per-phase cycle shares of k_front_s1 from a diagnostic build (-DIQGPU_STAMPS).
   run on the GPU box:  IQGPU_LIB=iq_tool_amd/lib/libiqgpu_stamps.so python tools/stamps.py"""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import iq_tool_amd
from iq_tool_amd import synth
from iq_tool_amd.chain import DeviceBuffer

frames = 1 << 26
raw = np.tile(synth.raw_stream(1 << 20, 2.4e6, 1, "cs16"), frames >> 20)
ch = iq_tool_amd.Chain(in_format="cs16", out_format="cs16", input_rate_hz=2.4e6, target_rate_hz=744187.5, shift_hz=200e3)
d_in = DeviceBuffer(raw.nbytes)
d_in.upload(raw)
d_out = DeviceBuffer(ch.max_out_frames(frames) * 4)
buf = np.zeros(65536, np.uint8)
ch.process_device(d_in.ptr, frames, d_out.ptr, d_out.nbytes)
ch._lib.iqgpu_chain_debug_read_scratch(ch._h, buf.ctypes.data_as(C.c_void_p))
ch.process_device(d_in.ptr, frames, d_out.ptr, d_out.nbytes)
ch._lib.iqgpu_chain_debug_read_scratch(ch._h, buf.ctypes.data_as(C.c_void_p))
acc = buf[32768:32768 + 64].view(np.uint64).astype(np.float64)
names = ["wait loads + unpack", "stores + cmul + X write", "(unused)", "half-band + window + tap gathers", "polyphase fma", "pack + k", "slide", "-"]
tiles = frames / 512 * (1 + 1 / 32.0)
tot = acc.sum()
for n, v in zip(names, acc):
    print("%-24s %10.0f cycles/tile  %5.1f %%" % (n, v / tiles, 100 * v / max(tot, 1)))
print("total %.0f cycles per wave-tile" % (tot / tiles))
