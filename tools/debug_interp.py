import os, sys, subprocess, json
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import iq_tool_amd as gpu
from iq_tool_amd import synth
kw = dict(in_format="cs16", out_format="cs16", input_rate_hz=600e3, target_rate_hz=2.4e6)
n = 150001
raw = synth.raw_stream(n, 600e3, 6, "cs16")
one = gpu.Chain(**kw).process(raw)
ch = gpu.Chain(**kw)
parts = []
for p in range(0, n, 65536):
    parts.append(ch.process(raw[2 * p:2 * min(n, p + 65536)]))
chunked = np.concatenate(parts)
print("sizes", one.size, chunked.size)
d = np.abs(one.astype(np.int64) - chunked.astype(np.int64))
bad = np.nonzero(d > 1)[0]
print("python chunked vs one call: max", d.max(), "bad", bad.size, bad[:10], bad[-5:] if bad.size else "")
fin, fout = "/tmp/in.raw", "/tmp/out.raw"
raw.tofile(fin)
exe = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "iq_tool_amd", "lib", "iqgpu_run")
for cf in ("65536", "262144", "16384"):
    r = subprocess.run([exe, "-i", fin, "-o", fout, "--raw-file-input-rate", "600000.0", "--raw-file-input-sample-format", "cs16", "--output-rate", "2400000.0",
                        "--output-sample-format", "cs16", "--chunk-frames", cf], capture_output=True, text=True)
    got = np.fromfile(fout, np.int16)
    d = np.abs(one.astype(np.int64) - got.astype(np.int64)) if got.size == one.size else None
    if d is None:
        print("harness chunk", cf, "size mismatch", got.size, one.size, r.stderr[-200:])
    else:
        bad = np.nonzero(d > 1)[0]
        print("harness chunk", cf, "max", d.max(), "bad", bad.size, bad[:10], bad[-5:] if bad.size else "")
