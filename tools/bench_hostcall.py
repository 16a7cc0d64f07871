import sys, time, numpy as np
sys.path.insert(0, ".")
import iq_tool_amd
from iq_tool_amd import synth
for lf in (14, 18, 22):
    n = 1 << lf
    raw = synth.raw_stream(n, 2.4e6, 1, "cs16")
    ch = iq_tool_amd.Chain(in_format="cs16", out_format="cs16", input_rate_hz=2.4e6, target_rate_hz=744187.5, shift_hz=200e3)
    for _ in range(5): ch.process(raw)
    t0 = time.perf_counter(); k = 200 if lf < 22 else 30
    for _ in range(k): ch.process(raw)
    dt = (time.perf_counter() - t0) / k
    print("iqgpu_chain_process (pageable host buffers), %d frames per call: %.1f us per call, %.2f GS/s" % (n, dt * 1e6, n / dt / 1e9))
