"""config 3's chain with the dc blocker / iq correction / filter switched off one at a time: ms per 2^27-frame call (device-resident)"""
import sys, time, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import numpy as np, torch
import iq_tool_amd, bench
from iq_tool_amd import synth
dev = torch.device("cuda:0")
frames = 1 << 27
seg = synth.raw_stream(1 << 22, 10e6, 1, "cs16")
d_in = torch.from_numpy(seg).to(dev).repeat(frames >> 22).contiguous()
base = dict(bench.OTHER[3]["chain"])
variants = [("config 3", {}), ("no dc", dict(dc_block=False)), ("no iq", dict(iq_correct=False)), ("no dc, no iq", dict(dc_block=False, iq_correct=False)),
            ("no filter", dict(filters=(), filter_taps=0)), ("front only, no dc / iq", dict(dc_block=False, iq_correct=False, filters=(), filter_taps=0))]
for name, chg in variants:
    kw = dict(base); kw.update(chg)
    chain = iq_tool_amd.Chain(device=0, block_samples=0, **kw)
    chain.set_stream(torch.cuda.current_stream(dev).cuda_stream)
    d_out = torch.empty(chain.max_out_frames(frames) * chain.out_bytes, dtype=torch.uint8, device=dev)
    def run(n):
        torch.cuda.synchronize(dev); t = time.perf_counter()
        for _ in range(n): chain.process_device(d_in.data_ptr(), frames, d_out.data_ptr(), d_out.numel())
        torch.cuda.synchronize(dev); return (time.perf_counter() - t) / n * 1e3
    run(2000)
    chain.set_profiling(True); chain.profile(); ms = run(300); prof = chain.profile()
    print("%-26s %.4f ms  %s  %s" % (name, ms, chain.front_kernel(), ", ".join("%s %.3f" % (k, v["ms"] / 300) for k, v in prof.items() if v["launches"])), flush=True)
