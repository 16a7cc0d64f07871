#!/bin/bash
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd "$REPO"; mkdir -p gpurun_out/s26
timeout 900 python3 -m pytest tests -x -q -m gpu -k "agc" 2>&1 | tail -15
python3 - <<'PY'
import time, numpy as np, iq_tool_amd
from iq_tool_amd import synth
from iq_tool_amd.chain import DeviceBuffer
for prof in ("digital", "local", "dx"):
    frames = 1 << 26
    raw = np.tile(synth.raw_stream(1 << 22, 2.4e6, 1, "cs16"), frames >> 22)
    ch = iq_tool_amd.Chain(in_format="cs16", out_format="cs16", input_rate_hz=2.4e6, target_rate_hz=744187.5, shift_hz=200e3, agc=True, agc_profile=prof)
    d_in = DeviceBuffer(raw.nbytes); d_in.upload(raw)
    d_out = DeviceBuffer(ch.max_out_frames(frames) * 4)
    for _ in range(2): ch.process_device(d_in.ptr, frames, d_out.ptr, d_out.nbytes)
    ch.synchronize(); t0 = time.perf_counter()
    for _ in range(3): got = ch.process_device(d_in.ptr, frames, d_out.ptr, d_out.nbytes)
    ch.synchronize(); dt = (time.perf_counter() - t0) / 3
    print("%-8s 2^26 frames (%d outputs): %.2f ms per call, %.1f GS/s in" % (prof, got, dt * 1e3, frames / dt / 1e9), ch.agc_state())
PY
