import ctypes as C, os, sys, time
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import iq_tool_amd
from iq_tool_amd import synth
from iq_tool_amd.chain import DeviceBuffer
frames = 1 << 28
raw = np.tile(synth.raw_stream(1 << 22, 2.4e6, 1, "cs16"), frames >> 22)
ch = iq_tool_amd.Chain(in_format="cs16", out_format="cs16", input_rate_hz=2.4e6, target_rate_hz=744187.5, shift_hz=200e3)
d_in = DeviceBuffer(raw.nbytes); d_in.upload(raw)
d_out = DeviceBuffer(ch.max_out_frames(frames) * 4)
buf = np.zeros(65536, np.uint8)
def read():
    ch._lib.iqgpu_chain_debug_read_scratch(ch._h, buf.ctypes.data_as(C.c_void_p))
t0 = time.time()
while time.time() - t0 < 2.0:
    for _ in range(50): ch.process_device(d_in.ptr, frames, d_out.ptr, d_out.nbytes)
    read()
for it in range(8):
    for _ in range(it * 3): ch.process_device(d_in.ptr, frames, d_out.ptr, d_out.nbytes)
    read()
    ch.process_device(d_in.ptr, frames, d_out.ptr, d_out.nbytes)
    read()
    w = buf[:32768].view(np.uint32).astype(np.int64)
    start, end = w[:4096], w[4096:8192]
    hw = buf[32768 + 256:32768 + 256 + 16384].view(np.uint32)
    ok = end != 0
    t0_ = start[ok].min()
    e_us = (end[ok] - t0_) / 100.0
    xcc = (hw[ok] >> 16) & 15
    idx = np.nonzero(ok)[0]
    rr = ((idx // 12) % 8 == xcc).mean()
    wg8 = (idx // 12) % 8
    if it == 0:
        print("xcc of workgroups 0..15:", [int(np.bincount(xcc[idx // 12 == g]).argmax()) for g in range(16)])
    print("launch %d: by wg %% 8, end p50 " % it + " ".join("%.0f" % np.median(e_us[wg8 == x]) for x in range(8)) + "   last-slot waves p50 " + " ".join("%.0f" % np.median(e_us[(wg8 == x) & (idx % 12 >= 8)]) for x in range(8)))
    print("launch %d: XCD end p50 " % it + " ".join("%.0f" % np.median(e_us[xcc == x]) for x in range(8)) + "  max %.0f  (wg %% 8 == xcc: %.2f)" % (e_us.max(), rr))
