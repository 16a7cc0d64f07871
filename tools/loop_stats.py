#!/usr/bin/env python3
"""Static instruction mix of the steady-state tile loop of the three kernels for the preset shape -- k_front_s1<4, fast>
(front_wave.hip), k_front_fat<false, 4, 6> (front_fat.hip), k_front_mid<6, false, 4, 0, false> (front_mid.hip): the innermost
loop of each kernel that holds the tile's packed FMAs.  Per loop: instructions by class and by opcode, and the same per 512 input
frames (a tile is 512 / 1024 / 768 frames).  usage: tools/loop_stats.py [-v]  (writes nothing; profiles/r03_isa_loop_stats.txt is
its output)"""
import collections
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KERNELS = [("front_wave.hip", "_ZN5iqgpu10k_front_s1ILi4ELb1ELb0ELb0ELi0EEEvNS_9FrontArgsE:", 512, 130),
           ("front_fat.hip", "_ZN5iqgpu11k_front_fatILb0ELi4ELi6EEEvNS_9FrontArgsE:", 1024, 230),
           ("front_mid.hip", "_ZN5iqgpu11k_front_midILi6ELb0ELi4ELi0ELb0ELb0ELb0ELi11ELi0EEEvNS_9FrontArgsE:", 768, 180)]


def loops_of(path, name):
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, "k.s")
        subprocess.run(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-slp-vectorize",
                        "-S", "--cuda-device-only", "-o", out, path], check=True, stderr=subprocess.DEVNULL)
        lines = open(out).read().split("\n")
    i0 = next(i for i, l in enumerate(lines) if l.startswith(name))
    i1 = next(i for i in range(i0, len(lines)) if lines[i].startswith(".Lfunc_end"))
    vg = re.search(r"\.amdhsa_kernel %s\n(.*?)\.end_amdhsa_kernel" % re.escape(name[:-1]), "\n".join(lines), re.S)
    vgpr = re.search(r"\.amdhsa_next_free_vgpr (\d+)", vg.group(1)).group(1) if vg else "?"
    loops, cur = collections.defaultdict(list), None
    for l in lines[i0:i1]:
        m = re.match(r"\.L(BB\d+_\d+):\s*;?\s*(.*)", l)
        if m:
            h = re.search(r"Header=(BB\d+_\d+)", m.group(2))
            cur = h.group(1) if h else (m.group(1) if "Loop Header" in m.group(2) else None)
            continue
        m = re.match(r";\s*%bb\.\d+:\s*;?\s*(.*)", l)
        if m:
            h = re.search(r"Header=(BB\d+_\d+)", m.group(1))
            cur = h.group(1) if h else None
            continue
        t = l.strip()
        if cur and t and not t.startswith(";") and not t.startswith("."):
            loops[cur].append(t.split()[0])
    return loops, vgpr


def main():
    verbose = "-v" in sys.argv
    for src, name, frames, min_fma in KERNELS:
        loops, vgpr = loops_of(os.path.join(ROOT, "iq_tool_amd", "csrc", src), name)
        cand = [(k, v) for k, v in loops.items() if v.count("v_pk_fma_f32") >= min_fma]
        key, ins = min(cand, key=lambda kv: len(kv[1]))        # the innermost loop that holds a whole tile's FMAs
        cls = collections.Counter()
        for op in ins:
            cls["VALU" if op.startswith("v_") else "LDS" if op.startswith("ds_") else "VMEM" if op.startswith(("global_", "scratch_", "buffer_"))
                else "waitcnt" if op.startswith("s_waitcnt") else "nop" if op.startswith("s_nop") else "SALU"] += 1
        ops = collections.Counter(ins)
        pk = sum(v for k, v in ops.items() if k.startswith("v_pk_"))
        print("%s  loop %s: %d instructions per %d-frame tile (static, all paths), %s VGPRs" % (name[:-1], key, len(ins), frames, vgpr))
        print("   " + ", ".join("%s %d" % kv for kv in sorted(cls.items(), key=lambda kv: -kv[1])))
        print("   per 512 frames: VALU %.0f (packed %.0f), LDS %.1f, all %.0f" % (cls["VALU"] * 512.0 / frames, pk * 512.0 / frames, cls["LDS"] * 512.0 / frames, len(ins) * 512.0 / frames))
        for op, n in ops.most_common(60 if verbose else 14):
            print("   %4d %s" % (n, op))
        print()


if __name__ == "__main__":
    main()
