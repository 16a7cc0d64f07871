#!/usr/bin/env python3
"""Experiment: the front kernel reading its batch straight from pinned host memory (and writing to it)
instead of H2D copy -> kernel -> D2H copy.  us per batch with calls queued back to back on one stream."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import iq_tool_amd
from iq_tool_amd import synth
from iq_tool_amd.chain import DeviceBuffer, PinnedBuffer

KW = dict(in_format="cs16", out_format="cs16", input_rate_hz=2.4e6, target_rate_hz=744187.5, shift_hz=200e3)
for lf in (14, 16, 18, 20, 22):
    n = 1 << lf
    raw = np.tile(synth.raw_stream(min(n, 1 << 20), 2.4e6, 1, "cs16"), max(1, n >> 20))
    ref = iq_tool_amd.Chain(**KW).process(raw)
    for mode in ("dev->dev", "host->dev", "host->host"):
        ch = iq_tool_amd.Chain(**KW)
        cap = ch.max_out_frames(n) * 4
        hin = [PinnedBuffer(n * 4) for _ in range(4)]
        hout = [PinnedBuffer(cap) for _ in range(4)]
        din, dout = DeviceBuffer(n * 4), DeviceBuffer(cap)
        din.upload(raw)
        for b in hin:
            b.array[:] = raw.view(np.uint8)
        k = 400

        def run(cnt):
            for i in range(cnt):
                src = din.ptr if mode == "dev->dev" else hin[i % 4].ptr
                dst = hout[i % 4].ptr if mode == "host->host" else dout.ptr
                got = ch.process_device(src, n, dst, cap)
            ch.synchronize()
            return got
        ch.reset(); got = run(1)
        if mode == "host->host":
            ok = np.array_equal(hout[0].array[:got * 4].view(np.int16), ref)
        else:
            ok = np.array_equal(dout.download(got * 4, np.int16), ref)
        run(20)
        t0 = time.perf_counter()
        run(k)
        dt = (time.perf_counter() - t0) / k
        print("%-10s 2^%d frames: %8.1f us per batch  %6.2f GS/s  %5.1f GB/s in  first call %s"
              % (mode, lf, dt * 1e6, n / dt / 1e9, n * 4 / dt / 1e9, "ok" if ok else "MISMATCH"), flush=True)
