"""debug helper: k_front_s2 against k_cascade + k_front_s1 on one aligned call: where do the outputs differ?"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import iq_tool_amd as gpu
from iq_tool_amd import synth

n = 1 << 21
raw = synth.raw_stream(n, 10e6, 41, "cs16")
for name, extra in (("plain", {}), ("dc", dict(dc_block=True)), ("iq", dict(iq_correct=True, iq_mag=0.01, iq_phase=-0.005)),
                    ("shift", dict(shift_hz=250e3)), ("dc+iq", dict(dc_block=True, iq_correct=True, iq_mag=0.01, iq_phase=-0.005))):
    kw = dict(in_format="cs16", out_format="cs16", input_rate_hz=10e6, target_rate_hz=2.4e6, **extra)
    os.environ.pop("IQGPU_NO_S2", None)
    ch = gpu.Chain(**kw); a = ch.process(raw); ka = ch.front_kernel()
    a2 = ch.process(raw[:2 * (1 << 20)])
    os.environ["IQGPU_NO_S2"] = "1"
    ch = gpu.Chain(**kw); b = ch.process(raw); kb = ch.front_kernel()
    b2 = ch.process(raw[:2 * (1 << 20)])
    d = np.abs(a.astype(np.int64) - b.astype(np.int64))
    bad = np.flatnonzero(d > 1)
    d2 = np.abs(a2.astype(np.int64) - b2.astype(np.int64))
    print(name, ka, kb, "size", a.size, b.size, "max", d.max(), "n>1", bad.size, "first", bad[:6] // 2, "last", bad[-3:] // 2,
          "| second call max", d2.max(), "n>1", int((d2 > 1).sum()))
    if bad.size:
        # runs of bad output frames -> which last-stage tile (about 246 outputs per 512 intermediate samples)
        fr = np.unique(bad // 2)
        gaps = np.flatnonzero(np.diff(fr) > 1)
        starts = np.concatenate([[fr[0]], fr[gaps + 1]]); ends = np.concatenate([fr[gaps], [fr[-1]]])
        print("   bad frame runs:", [(int(s), int(e)) for s, e in zip(starts[:8], ends[:8])], "of", starts.size)

print("---- the test's call sequence, per call")
n = (1 << 22) + 4 * 777
raw = synth.raw_stream(n, 10e6, 41, "cs16")
cuts = [0, 1 << 21, (1 << 21) + 40_000, (1 << 21) + 40_000 + 131_073, (1 << 21) + 40_000 + 131_073 + 262_147, n]
cuts[-2] = cuts[-2] + (-cuts[-2]) % 4
for name, extra in (("plain", {}), ("dc", dict(dc_block=True))):
    kw = dict(in_format="cs16", out_format="cs16", input_rate_hz=10e6, target_rate_hz=2.4e6, **extra)
    outs = {}
    for mode in ("fused", "two"):
        if mode == "two":
            os.environ["IQGPU_NO_S2"] = "1"
        else:
            os.environ.pop("IQGPU_NO_S2", None)
        ch = gpu.Chain(**kw)
        outs[mode] = []
        for a, b in zip(cuts[:-1], cuts[1:]):
            o = ch.process(raw[2 * a:2 * b]); outs[mode].append((o, ch.front_kernel()))
    for i, ((fa, ka), (fb, kb)) in enumerate(zip(outs["fused"], outs["two"])):
        d = np.abs(fa.astype(np.int64) - fb.astype(np.int64)) if fa.size == fb.size else np.array([-1])
        bad = np.flatnonzero(d > 1) // 2
        print(name, "call", i, cuts[i + 1] - cuts[i], ka, kb, "sizes", fa.size, fb.size, "max", d.max(), "bad frames", bad.size, bad[:4], bad[-2:])
