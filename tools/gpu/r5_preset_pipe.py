#!/usr/bin/env python3
"""round 5: the cs16-fm-nrsc5 preset (BASELINE configs[1] without the shift + digital AGC) through iqgpu_chain_submit / _collect
from pinned host buffers, for a kernel trace: past the lock a batch is TWO kernels (front with the fused AGC + k_agc_classify) --
the verdict is read on the host and the four fallback launches of the device-resident path are gone.
usage: r5_preset_pipe.py [batch_frames] [batches]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import iq_tool_amd
from iq_tool_amd import synth
from iq_tool_amd.chain import PinnedBuffer

batch = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 24
n_batches = int(sys.argv[2]) if len(sys.argv) > 2 else 48
kw = dict(in_format="cs16", out_format="cs16", input_rate_hz=2.4e6, target_rate_hz=744187.5, agc=True, agc_profile="digital")
ch = iq_tool_amd.Chain(**kw)
depth = ch._lib.iqgpu_chain_pipeline_depth()
cap = ch.max_out_frames(batch) * ch.out_bytes
seg = synth.raw_stream(1 << 20, 2.4e6, 1, "cs16").view(np.uint8)
slots = []
for _ in range(depth):
    ib, ob = PinnedBuffer(batch * 4), PinnedBuffer(cap)
    ib.array[:] = np.tile(seg, -(-ib.nbytes // seg.size))[:ib.nbytes]
    slots.append((ib, ob))
flight, t0, done = [], None, 0
for i in range(n_batches):
    if i == 8:
        for t in flight: ch.collect(t)
        flight = []; t0 = time.perf_counter()
    if len(flight) == depth: ch.collect(flight.pop(0))
    ib, ob = slots[i % depth]
    got, t = ch.submit(ib.ptr, batch, ob.ptr, cap)
    flight.append(t)
for t in flight: ch.collect(t)
dt = time.perf_counter() - t0
print("preset through submit / collect: %d frames per batch, %.1f us per batch, %.1f MS/s, AGC %s" % (batch, dt / (n_batches - 8) * 1e6, (n_batches - 8) * batch / dt / 1e6, ch.agc_state()))
