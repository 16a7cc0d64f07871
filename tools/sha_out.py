import hashlib, sys, os
sys.path.insert(0, os.getcwd())
import numpy as np
import iq_tool_amd
from iq_tool_amd import synth
from iq_tool_amd.chain import DeviceBuffer
frames = 1 << 28
raw = np.tile(synth.raw_stream(1 << 22, 2.4e6, 1, "cs16"), frames >> 22)
ch = iq_tool_amd.Chain(in_format="cs16", out_format="cs16", input_rate_hz=2.4e6, target_rate_hz=744187.5, shift_hz=200e3)
d_in = DeviceBuffer(raw.nbytes); d_in.upload(raw)
d_out = DeviceBuffer(ch.max_out_frames(frames) * 4)
got = ch.process_device(d_in.ptr, frames, d_out.ptr, d_out.nbytes)
ch.synchronize()
out = d_out.download(got * 4)
print(os.getcwd().split("/")[-1], got, hashlib.sha256(out.tobytes()).hexdigest())
