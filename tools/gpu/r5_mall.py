#!/usr/bin/env python3
"""round 5: does the cf32 hand-off between the front kernel and the user filter stay in the 256 MB Infinity Cache when a long
call is cut into sub-calls?  The -usb presets, 2^28 frames resident in HBM: one call against 2^(28-k) calls of 2^k frames
(any split of the stream gives the same bytes), wall time over the synchronised loop and the per-stage device times."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import iq_tool_amd
from iq_tool_amd import synth
from iq_tool_amd.chain import DeviceBuffer

frames = 1 << 28
for fmt, target in (("cu8", 1488375.0), ("cs16", 744187.5)):
    kw = dict(in_format=fmt, out_format=fmt, input_rate_hz=2.4e6, target_rate_hz=target, agc=True, agc_profile="digital",
              filters=(("passband", 158.5e3, 113e3),))
    bpf = 2 if fmt == "cu8" else 4
    raw = np.tile(synth.raw_stream(1 << 22, 2.4e6, 3, fmt), frames >> 22)
    d_in = DeviceBuffer(raw.nbytes); d_in.upload(raw)
    ref = None
    for k in (28, 26, 25, 24, 23, 22):
        ch = iq_tool_amd.Chain(**kw)
        d_out = DeviceBuffer(ch.out_bytes * (ch.max_out_frames(frames) + 4096))
        n_sub = frames >> k
        def step():
            off = 0
            for i in range(n_sub):
                got = ch.process_device(d_in.ptr + i * (bpf << k), 1 << k, d_out.ptr + off * ch.out_bytes, d_out.nbytes - off * ch.out_bytes)
                off += got
            return off
        for _ in range(4):
            step()
        ch.synchronize()
        ch.set_profiling(True); ch.profile()
        t0 = time.perf_counter()
        for _ in range(5):
            n = step()
        ch.synchronize()
        dt = (time.perf_counter() - t0) / 5
        p = ch.profile()
        print(fmt, "2^%d x %d" % (k, n_sub), "ms per 2^28 frames %.4f" % (dt * 1e3), {q: round(v["ms"] / 5, 4) for q, v in p.items() if v["launches"]}, "out", n, flush=True)
        ch.close()
