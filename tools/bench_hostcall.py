#!/usr/bin/env python3
"""Cost of the host-buffer entry points (through ctypes):
  * iqgpu_chain_process with pageable caller buffers (one synchronous call per batch),
  * iqgpu_chain_submit / _collect with pinned buffers (iqgpu_chain_pipeline_depth() batches in flight),
at 2^14 (one reference chunk), 2^18 (the INTEGRATION.md stub's 16-chunk batch), 2^22 and 2^24 frames per batch."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import iq_tool_amd
from iq_tool_amd import synth
from iq_tool_amd.chain import PinnedBuffer

KW = dict(in_format="cs16", out_format="cs16", input_rate_hz=2.4e6, target_rate_hz=744187.5, shift_hz=200e3)
for lf in (14, 18, 22, 24):
    n = 1 << lf
    raw = synth.raw_stream(min(n, 1 << 22), 2.4e6, 1, "cs16")
    raw = np.tile(raw, max(1, n >> 22))
    if lf <= 22:
        ch = iq_tool_amd.Chain(**KW)
        for _ in range(5):
            ch.process(raw)
        k = 200 if lf < 22 else 30
        t0 = time.perf_counter()
        for _ in range(k):
            ch.process(raw)
        dt = (time.perf_counter() - t0) / k
        print("iqgpu_chain_process (pageable, synchronous), %8d frames per call: %8.1f us per call, %6.2f GS/s" % (n, dt * 1e6, n / dt / 1e9))
    ch = iq_tool_amd.Chain(**KW)
    depth = ch._lib.iqgpu_chain_pipeline_depth()
    cap = ch.max_out_frames(n) * 4
    slots = [(PinnedBuffer(n * 4), PinnedBuffer(cap)) for _ in range(depth)]
    for ib, _ in slots:
        ib.array[:] = raw.view(np.uint8)
    total = max(depth * 4, min(4000, (1 << 31) >> lf))

    def run(count):
        flight = []
        for i in range(count):
            if len(flight) == depth:
                ch.collect(flight.pop(0))
            ib, ob = slots[i % depth]
            flight.append(ch.submit(ib.ptr, n, ob.ptr, cap)[1])
        for t in flight:
            ch.collect(t)

    run(depth * 2)
    t0 = time.perf_counter()
    run(total)
    dt = (time.perf_counter() - t0) / total
    print("iqgpu_chain_submit/_collect (pinned, %d in flight), %8d frames per batch: %8.1f us per batch, %6.2f GS/s sustained"
          % (depth, n, dt * 1e6, n / dt / 1e9))
