#!/usr/bin/env python3
"""In-kernel shader clock and per-wave timeline of k_front_s1 from a diagnostic build (-DIQGPU_CLOCKSTAMP):
cycles (s_memtime) / 100 MHz ticks (s_memrealtime) per wave run, after >= 2 s of back-to-back launches, and
for one launch when every wave started and ended (per XCD / CU).
   run on the GPU box:  IQGPU_LIB=iq_tool_amd/lib/libiqgpu_clock.so python tools/clock.py"""
import ctypes as C
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import iq_tool_amd
from iq_tool_amd import synth
from iq_tool_amd.chain import DeviceBuffer

frames = 1 << int(os.environ.get("IQGPU_CLOCK_LOG2", "28"))       # (with IQGPU_CUS=n: scale the work with the CUs, e.g. 128 CUs, 2^27 frames)
raw = np.tile(synth.raw_stream(1 << 22, 2.4e6, 1, "cs16"), frames >> 22)
ch = iq_tool_amd.Chain(in_format="cs16", out_format="cs16", input_rate_hz=2.4e6, target_rate_hz=744187.5, shift_hz=200e3)
d_in = DeviceBuffer(raw.nbytes)
d_in.upload(raw)
d_out = DeviceBuffer(ch.max_out_frames(frames) * 4)
buf = np.zeros(65536, np.uint8)


def read():
    ch._lib.iqgpu_chain_debug_read_scratch(ch._h, buf.ctypes.data_as(C.c_void_p))
    return buf[32768 + 128:32768 + 128 + 24].view(np.uint64).astype(np.float64)


for label, n in (("cold (first 3 launches)", 3), ("after 2 s of launches", 0), ("next 20 launches", 20)):
    if n == 0:
        t0 = time.time()
        while time.time() - t0 < 2.0:
            for _ in range(50):
                ch.process_device(d_in.ptr, frames, d_out.ptr, d_out.nbytes)
            read()
        continue
    read()
    t0 = time.perf_counter()
    for _ in range(n):
        ch.process_device(d_in.ptr, frames, d_out.ptr, d_out.nbytes)
    cyc, rt, waves = read()
    dt = time.perf_counter() - t0
    print("%-26s %.3f ms/launch  wave run %.0f cycles = %.1f us  clock %.3f GHz  (%d wave runs)"
          % (label, dt / n * 1e3, cyc / waves, rt / waves / 100.0, cyc / rt * 0.1, waves))

# one launch: the timeline of its waves
read()
ch.process_device(d_in.ptr, frames, d_out.ptr, d_out.nbytes)
read()
w = buf[:32768].view(np.uint32).astype(np.int64)
start, end = w[:4096], w[4096:8192]
hw = buf[32768 + 256:32768 + 256 + 16384].view(np.uint32)
ok = end != 0
t0 = start[ok].min()
s_us, e_us = (start[ok] - t0) / 100.0, (end[ok] - t0) / 100.0
edge = (hw[ok] >> 31) != 0
xcc = (hw[ok] >> 16) & 15
print("one launch, %d waves (%d edge): starts %.1f .. %.1f us, ends %.1f .. %.1f us (p5 %.1f p50 %.1f p95 %.1f), run length p5 %.1f p50 %.1f p95 %.1f max %.1f us"
      % (ok.sum(), edge.sum(), s_us.min(), s_us.max(), e_us.min(), e_us.max(), *np.percentile(e_us, [5, 50, 95]),
         *np.percentile(e_us - s_us, [5, 50, 95]), (e_us - s_us).max()))
if edge.any():
    print("edge waves: end %.1f .. %.1f us" % (e_us[edge].min(), e_us[edge].max()))
for x in range(8):
    m = (xcc == x) & ~edge
    if m.any():
        print("  XCD %d: %4d waves, start p50 %.1f, end p50 %.1f max %.1f, run p50 %.1f max %.1f us"
              % (x, m.sum(), np.median(s_us[m]), np.median(e_us[m]), e_us[m].max(), np.median((e_us - s_us)[m]), (e_us - s_us)[m].max()))
# per wave slot of a workgroup (WPW = IQGPU_CLOCK_WPW waves per workgroup: 12 for k_front_mid, 16 for k_front_s1): is the spread systematic?
WPW = int(os.environ.get("IQGPU_CLOCK_WPW", "12"))
idx_all = np.nonzero(ok)[0]
slot = idx_all % WPW
print("per wave slot (end time, us): " + "  ".join("%d: p50 %.0f p95 %.0f" % (k, *np.percentile(e_us[(slot == k) & ~edge], [50, 95])) for k in range(WPW) if ((slot == k) & ~edge).any()))
wg = idx_all // WPW
within = []
for g in np.unique(wg):
    m = (wg == g) & ~edge
    if m.sum() >= 2:
        within.append((e_us[m].min(), np.median(e_us[m]), e_us[m].max()))
within = np.array(within)
print("within a workgroup: first wave ends p50 %.1f, median wave p50 %.1f, last wave p50 %.1f us; last - first p50 %.1f p95 %.1f us"
      % (np.median(within[:, 0]), np.median(within[:, 1]), np.median(within[:, 2]), *np.percentile(within[:, 2] - within[:, 0], [50, 95])))
# per workgroup (16 consecutive waves share a CU): spread of workgroup end times
idx = np.nonzero(ok)[0]
wg_end = {}
for i, e in zip(idx, e_us):
    wg_end.setdefault(i // WPW, []).append(e)
ends = np.array([max(v) for v in wg_end.values()])
print("workgroups: %d, end time min %.1f p50 %.1f p95 %.1f max %.1f us" % (len(ends), ends.min(), *np.percentile(ends, [50, 95]), ends.max()))
# run stealing (k_front_mid): runs a wave took from others, and how close the last wave ends to the median
st = buf[49408:49408 + 4 * 4032].view(np.uint32)[ok[:4032]]
if st.size:
    print("steals per wave: total %d, mean %.2f, max %d, waves with none %d; last wave end / median wave end = %.3f"
          % (st.sum(), st.mean(), st.max(), (st == 0).sum(), e_us.max() / np.median(e_us[~edge])))
