"""ms per step of the headline chain with the per-kernel HIP events on and off (what do the events of bench.py's timed region cost?)"""
import sys, time, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import numpy as np, torch
import iq_tool_amd, bench
from iq_tool_amd import synth
dev = torch.device("cuda:0")
frames = 1 << 28
seg = synth.raw_stream(1 << 22, 2.4e6, 1, "cs16")
d_in = torch.from_numpy(seg).to(dev).repeat(frames >> 22).contiguous()
for name, kw in (("headline", bench.CHAIN), ("preset", dict(bench.CHAIN, agc=True))):
    chain = iq_tool_amd.Chain(device=0, block_samples=0, **kw)
    chain.set_stream(torch.cuda.current_stream(dev).cuda_stream)
    d_out = torch.empty(chain.max_out_frames(frames) * chain.out_bytes, dtype=torch.uint8, device=dev)
    def run(n):
        torch.cuda.synchronize(dev); t = time.perf_counter()
        for _ in range(n): chain.process_device(d_in.data_ptr(), frames, d_out.data_ptr(), d_out.numel())
        torch.cuda.synchronize(dev); return (time.perf_counter() - t) / n * 1e3
    run(3000)
    for rep in range(3):
        chain.set_profiling(False); off = run(500)
        chain.set_profiling(True); chain.profile(); on = run(500); chain.profile()
        print(name, "events off %.4f ms  on %.4f ms  (+%.1f us)" % (off, on, (on - off) * 1e3), flush=True)
