#!/usr/bin/env python3
"""Static instruction mix of the streaming loop of k_front_s1<4, fast> (all blocks of the loop that
holds the register-prefetch global_load_dwordx4 pair).  usage: tools/loop_stats.py [-v] [--dump]"""
import collections
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "iq_tool_amd", "csrc", "front_wave.hip")


def main():
    verbose = "-v" in sys.argv
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, "k.s")
        subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-slp-vectorize",
                        "-S", "--cuda-device-only", "-o", out, SRC], check=True, stderr=subprocess.DEVNULL)
        lines = open(out).read().split("\n")
    name = "_ZN5iqgpu10k_front_s1ILi4ELb1ELb0ELb0ELi0EEEvNS_9FrontArgsE:"
    i0 = next(i for i, l in enumerate(lines) if l.startswith(name))
    i1 = next(i for i in range(i0, len(lines)) if lines[i].startswith(".Lfunc_end"))
    loops = collections.defaultdict(list)      # header label -> instructions
    cur = None
    for l in lines[i0:i1]:
        m = re.match(r"\.L(BB\d+_\d+):\s*;?\s*(.*)", l)
        if m:
            lab, cmt = m.group(1), m.group(2)
            h = re.search(r"Header=(BB\d+_\d+)", cmt)
            cur = h.group(1) if h else (lab if "Loop Header" in cmt else None)
            continue
        m = re.match(r";\s*%bb\.\d+:\s*;?\s*(.*)", l)
        if m:
            h = re.search(r"Header=(BB\d+_\d+)", m.group(1))
            cur = h.group(1) if h else None
            continue
        t = l.strip()
        if cur and t and not t.startswith(";") and not t.startswith("."):
            loops[cur].append(t)
    key = max((k for k, v in loops.items() if sum(i.startswith("global_load_dwordx4") for i in v) >= 2),
              key=lambda k: len(loops[k]))
    ins = loops[key]
    if "--dump" in sys.argv:
        print("\n".join(ins))
        return
    cls, ops = collections.Counter(), collections.Counter()
    for i in ins:
        op = i.split()[0]
        ops[op] += 1
        if op.startswith("v_"):
            cls["VALU"] += 1
        elif op.startswith("ds_"):
            cls["LDS"] += 1
        elif op.startswith(("global_", "flat_", "buffer_")):
            cls["VMEM"] += 1
        elif op.startswith("s_waitcnt"):
            cls["waitcnt"] += 1
        elif op.startswith("s_nop"):
            cls["nop"] += 1
        elif op.startswith(("s_cbranch", "s_branch")):
            cls["branch"] += 1
        else:
            cls["SALU"] += 1
    print("loop %s: %d instructions (static, all paths)" % (key, len(ins)))
    print("  " + ", ".join("%s %d" % kv for kv in sorted(cls.items(), key=lambda kv: -kv[1])))
    for op, n in ops.most_common(60 if verbose else 20):
        print("  %4d %s" % (n, op))


if __name__ == "__main__":
    main()
