#!/usr/bin/env python3
"""round 5: what a call costs with the RMS output AGC (`local` is the reference's default --output-agc, `dx`): device time per
process_device call of the NRSC-5 chain at the binding's batch sizes -- the latency floor warm + chunk dependent samples sets."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import iq_tool_amd
from iq_tool_amd import synth
from iq_tool_amd.chain import DeviceBuffer

for profile in ("local", "dx"):
    for frames in (16384, 262144, 1048576, 1 << 24):
        kw = dict(in_format="cs16", out_format="cs16", input_rate_hz=2.4e6, target_rate_hz=744187.5, shift_hz=200e3, agc=True, agc_profile=profile)
        ch = iq_tool_amd.Chain(**kw)
        raw = np.tile(synth.raw_stream(min(frames, 1 << 20), 2.4e6, 3, "cs16"), max(1, frames >> 20))
        d_in, d_out = DeviceBuffer(raw.nbytes), DeviceBuffer(4 * ch.max_out_frames(frames))
        d_in.upload(raw)
        reps = 6 if profile == "dx" else 30
        for _ in range(2):
            ch.process_device(d_in.ptr, frames, d_out.ptr, d_out.nbytes)
        ch.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            ch.process_device(d_in.ptr, frames, d_out.ptr, d_out.nbytes)
        ch.synchronize()
        dt = (time.perf_counter() - t0) / reps
        print("%-5s %9d frames per call: %8.3f ms per call, %8.1f MS/s" % (profile, frames, dt * 1e3, frames / dt / 1e6), flush=True)
        ch.close()
