#!/usr/bin/env python3
"""Device-resident throughput of an arbitrary chain (same method as bench.py, no roofline / baseline):
   tools/bench_chain.py --in-format cu8 --out-format cu8 --in-rate 2.4e6 --out-rate 1488375 [--log2-frames 28] [chain options]"""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--in-format", default="cs16"); ap.add_argument("--out-format", default="cs16")
    ap.add_argument("--in-rate", type=float, default=2.4e6); ap.add_argument("--out-rate", type=float, default=744187.5)
    ap.add_argument("--shift", type=float, default=0.0); ap.add_argument("--dc-block", action="store_true")
    ap.add_argument("--agc", action="store_true"); ap.add_argument("--log2-frames", type=int, default=26)
    ap.add_argument("--steps", type=int, default=10); ap.add_argument("--gain", type=float, default=1.0)
    a = ap.parse_args()
    import torch
    import iq_tool_amd
    from iq_tool_amd import synth
    frames = 1 << a.log2_frames
    seg = synth.raw_stream(1 << 20, a.in_rate, 1, a.in_format)
    d_in = torch.from_numpy(seg).cuda().repeat(frames >> 20).contiguous()
    ch = iq_tool_amd.Chain(in_format=a.in_format, out_format=a.out_format, input_rate_hz=a.in_rate, target_rate_hz=a.out_rate,
                           shift_hz=a.shift, dc_block=a.dc_block, agc=a.agc, gain=a.gain, block_samples=0)
    ch.set_stream(torch.cuda.current_stream().cuda_stream)
    d_out = torch.empty(ch.max_out_frames(frames) * ch.out_bytes, dtype=torch.uint8, device="cuda")
    n_out = 0
    for _ in range(3):
        n_out = ch.process_device(d_in.data_ptr(), frames, d_out.data_ptr(), d_out.numel())
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(a.steps):
        ch.process_device(d_in.data_ptr(), frames, d_out.data_ptr(), d_out.numel())
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / a.steps
    gb = (frames * ch.in_bytes + n_out * ch.out_bytes) / 1e9
    print("%s %.0f -> %s %.1f: %.3f ms per 2^%d frames, %.1f GS/s in, %.1f GS/s out, %.0f GB/s" %
          (a.in_format, a.in_rate, a.out_format, a.out_rate, dt * 1e3, a.log2_frames, frames / dt / 1e9, n_out / dt / 1e9, gb / dt))


if __name__ == "__main__":
    main()
