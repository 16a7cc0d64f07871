#!/usr/bin/env python3
"""round 6, VERDICT r5 item 3 (ii): would the DC prefix pass run faster if its span were still in the 256 MB Infinity Cache?
The same chain (BASELINE configs[2]'s front: cs16 10 MS/s -> 2.4 MS/s, dc block + iq correction) on resident inputs of 2^22 .. 2^27
frames (16 MB .. 512 MB), the call repeated back to back on the SAME buffer, so that from the second call on k_dc_prefix reads what
the call before has just read (and k_front_s2 behind it): per-launch times of k_dc_prefix and of the front kernel from the chain's
own HIP-event profile.  If the Infinity Cache served the prefix pass, its bytes per second would rise as the input shrinks below
256 MB."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import iq_tool_amd                                                        # noqa: E402
from iq_tool_amd import synth                                             # noqa: E402
from iq_tool_amd.chain import DeviceBuffer                                # noqa: E402

kw = dict(in_format="cs16", out_format="cs16", input_rate_hz=10e6, target_rate_hz=2.4e6, dc_block=True, iq_correct=True, iq_mag=0.01, iq_phase=0.02)
seg = synth.raw_stream(1 << 22, 10e6, 3, "cs16")
out = {}
for lg in (22, 23, 24, 25, 26, 27):
    n = 1 << lg
    ch = iq_tool_amd.Chain(**kw)
    d_in = DeviceBuffer(4 * n)
    host = np.tile(seg, n // (1 << 22))
    d_in.upload(host)
    d_out = DeviceBuffer(4 * ch.max_out_frames(n))
    for _ in range(5):
        ch.process_device(d_in.ptr, n, d_out.ptr, d_out.nbytes)
    ch.synchronize()
    ch.set_profiling(True)
    reps = 20
    for _ in range(reps):
        ch.process_device(d_in.ptr, n, d_out.ptr, d_out.nbytes)
    ch.synchronize()
    p = ch.profile()
    pre = p["dc_prefix"]["ms"] / max(p["dc_prefix"]["launches"], 1)
    fr = p["front"]["ms"] / max(p["front"]["launches"], 1)
    out[lg] = dict(input_MB=4 * n / 2**20, dc_prefix_us=round(pre * 1e3, 2), dc_prefix_TBps=round(4 * n / (pre * 1e-3) / 1e12, 2) if pre else None,
                   front_us=round(fr * 1e3, 2), front_kernel=ch.front_kernel())
    print(lg, out[lg], flush=True)
    d_in.free(); d_out.free(); ch.close()
os.makedirs(os.path.join(ROOT, "gpurun_out", "r6"), exist_ok=True)
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "r6", "mall_dc.json"), "w"), indent=1)
