#!/usr/bin/env python3
"""round 5: k_cascade2 against k_cascade on random geometry -- formats, stage counts, block sizes (runs of 2 .. 40 tiles), ragged call
splits that leave the stream at any phase of a decimation group.  Bytes must be equal.  One child per switch setting per batch of cases
(the switches are read at chain create)."""
import os, sys, json, subprocess, hashlib
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))

def cases(seed, n):
    rng = np.random.default_rng(seed)
    out = []
    shapes = [("cu8", 20e6, 1488375.0), ("cu8", 20e6, 744187.5), ("cu8", 61.44e6, 1488375.0), ("cs16", 2.4e6, 46511.71875),
              ("cs16", 20e6, 1488375.0), ("sc16q11", 20e6, 744187.5), ("cs16", 10e6, 700e3), ("cs8", 20e6, 744187.5), ("cs8", 20e6, 400e3)]
    for _ in range(n):
        fmt, ri, ro = shapes[int(rng.integers(len(shapes)))]
        block = int(rng.integers(2, 41)) * 8192
        total = int(rng.integers(1 << 20, 5 << 20))
        cuts, pos = [], 0
        while pos < total:
            k = int(rng.choice([rng.integers(1, 64), rng.integers(1 << 16, 1 << 21), 8 * rng.integers(1 << 13, 1 << 18)]))
            k = min(k, total - pos); cuts.append(k); pos += k
        out.append(dict(fmt=fmt, ri=ri, ro=ro, block=block, cuts=cuts, outf=str(rng.choice(["cs16", "cu8", "cf32"])), seed=int(rng.integers(1 << 30))))
    return out

if len(sys.argv) > 2:
    import iq_tool_amd
    from iq_tool_amd import synth
    res = []
    for c in cases(int(sys.argv[2]), int(sys.argv[3])):
        n = sum(c["cuts"])
        raw = synth.raw_stream(n, c["ri"], c["seed"], c["fmt"])
        per = raw.size // n
        ch = iq_tool_amd.Chain(in_format=c["fmt"], out_format=c["outf"], input_rate_hz=c["ri"], target_rate_hz=c["ro"], block_samples=c["block"])
        h, pos, names = hashlib.sha256(), 0, set()
        for k in c["cuts"]:
            h.update(ch.process(raw[per * pos:per * (pos + k)]).tobytes()); pos += k
            names.add(ch.front_kernel())
        res.append((h.hexdigest(), sorted(names)))
        ch.close()
    print(json.dumps(res))
else:
    seed, n = (int(sys.argv[1]) if len(sys.argv) > 1 else 1), 40
    outs = {}
    for name, env in (("one", {"IQGPU_NO_CASC2": "1"}), ("two", {})):
        r = subprocess.run([sys.executable, __file__, name, str(seed), str(n)], env=dict(os.environ, **env), capture_output=True, text=True)
        if r.returncode != 0:
            print(r.stderr[-2000:]); sys.exit(1)
        outs[name] = json.loads(r.stdout.strip().splitlines()[-1])
    bad = [i for i, (a, b) in enumerate(zip(outs["one"], outs["two"])) if a[0] != b[0]]
    used = sum(1 for _, nm in outs["two"] if any("k_cascade2" in x for x in nm))
    print("seed", seed, "cases", n, "with k_cascade2:", used, "mismatches:", bad)
    if bad:
        cs = cases(seed, n)
        for i in bad: print(cs[i], outs["one"][i][1], outs["two"][i][1])
        sys.exit(1)
