#!/usr/bin/env python3
"""Does a long call of a multi-kernel chain run faster as MALL-sized sub-calls (the cf32 intermediate of sub-call k and the raw
input behind the dc prefix pass are then read back from the 256 MiB Infinity Cache instead of HBM)?  BASELINE configs[2] / [3],
one call against 2, 4, 8, 16 sub-calls of equal size; same bytes either way (the chain is split-invariant).
   tools/split_probe.py [3|4]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402


def main():
    cfg = int(sys.argv[1]) if len(sys.argv) > 1 else 3
    import torch
    import iq_tool_amd
    from iq_tool_amd import synth
    o = bench.OTHER[cfg]
    frames = 1 << o["log2_frames"]
    seg = synth.raw_stream(1 << 22, o["rate"], 1, o["fmt"])
    d_in = torch.from_numpy(seg).cuda().repeat(frames >> 22).contiguous()
    for parts in (1, 2, 4, 8, 16):
        ch = iq_tool_amd.Chain(device=0, block_samples=0, **o["chain"])
        ch.set_stream(torch.cuda.current_stream().cuda_stream)
        d_out = torch.empty(ch.max_out_frames(frames) * ch.out_bytes + 64, dtype=torch.uint8, device="cuda")
        sub = frames // parts

        def step():
            pos = 0
            for p in range(parts):
                n = ch.process_device(d_in.data_ptr() + p * sub * o["bps"], sub, d_out.data_ptr() + pos, d_out.numel() - pos)
                pos += n * ch.out_bytes
        t0 = time.perf_counter()
        while time.perf_counter() - t0 < 1.0:
            for _ in range(20):
                step()
            torch.cuda.synchronize()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(20):
            step()
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 20
        print("config %d: %2d sub-calls of 2^%.1f frames: %.4f ms per 2^%d frames" % (cfg, parts, __import__("math").log2(sub), dt * 1e3, o["log2_frames"]), flush=True)
        ch.close()


if __name__ == "__main__":
    main()
