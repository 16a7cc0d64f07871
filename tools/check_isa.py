#!/usr/bin/env python3
"""Static check of the hand-written asm in k_front_s1 (iq_tool_amd/csrc/front_wave.hip).

The polyphase tap gather issues ds_read_b64 from inline asm.  hipcc does not know that the destination
registers of asm loads are still in flight, so every such block must wait for its own reads: this
script compiles front_wave.hip to gfx950 ISA and verifies for every instantiation that
  (1) each asm block that reads LDS holds the 14 reads of one slot pair and ENDS with
      s_waitcnt lgkmcnt(0) (nothing of the compiler's can then see a register that is still filling),
  (2) there are two such blocks per polyphase section,
  (3) no asm global_load / buffer_load is left in the file (loads are compiler-managed),
  (4) neither k_front_s1 nor k_cascade (cascade_wave.hip) holds a ds_read2_b32: that is what hipcc's load vectoriser makes of
      a window load it has trimmed to the dwords in use -- two 4-byte accesses per lane at a 16-byte lane stride, which
      conflict (SQ_LDS_BANK_CONFLICT 94 -> 12 cycles per tile when the cascade's window loads were made whole).
  (10) no instantiation of k_front_p0 (front_p0.hip) or k_cascade2 (cascade2.hip) uses scratch or more registers than its occupancy allows.
The sources are compiled side by side (front_mid.hip alone: 2 min 40 s).  Exit code 0 = ok.  Run by __graft_entry__.build() and
tests/test_host_logic.py."""
import os
import re
import subprocess
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "..", "iq_tool_amd", "csrc", "front_wave.hip")
SRC_CASC = os.path.join(HERE, "..", "iq_tool_amd", "csrc", "cascade_wave.hip")


def source_flags(src):
    """the per-source flags of the real build (iq_tool_amd/build.py SOURCE_FLAGS): the ISA checked is the ISA shipped"""
    sys.path.insert(0, os.path.join(HERE, ".."))
    from iq_tool_amd import build as b
    return list(b.SOURCE_FLAGS.get(os.path.basename(src), []))


def compile_isa(src=None):
    src = src or SRC
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, "fw.s")
        cmd = ["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-slp-vectorize", *source_flags(src),
               "--cuda-device-only", "-S", src, "-o", out]
        subprocess.run(cmd, check=True, stderr=subprocess.DEVNULL)
        return open(out).read().split("\n")


def asm_blocks(lines):
    """[(start, end, [instructions])] for every ;;#ASMSTART .. ;;#ASMEND region"""
    out, i = [], 0
    while i < len(lines):
        if lines[i].strip().startswith(";;#ASMSTART"):
            j = i + 1
            body = []
            while not lines[j].strip().startswith(";;#ASMEND"):
                if lines[j].strip():
                    body.append(lines[j].strip())
                j += 1
            out.append((i, j, body))
            i = j
        i += 1
    return out


def check(lines):
    errors = []
    funcs, cur = {}, None
    for idx, l in enumerate(lines):
        m = re.match(r"(_ZN5iqgpu10k_front_s1ILi(\d)ELb([01])ELb([01])ELb([01])ELi(\d)EEEvNS_9FrontArgsE):", l)
        if m:
            cur = (m.group(2) + (", fast" if m.group(3) == "1" else "") + (", s0" if m.group(4) == "1" else "") +
                   (", agc" if m.group(5) == "1" else "") + ({"0": "", "1": ", nonco", "2": ", plain cu8", "3": ", plain cs16->cu8", "4": ", mid stage", "5": ", dc + mixer", "6": ", dc", "7": ", s0 cu8 + mixer", "8": ", s0 cu8 + dc + mixer", "9": ", s0 cu8 + dc"}[m.group(6)]))
            funcs[cur] = []
        elif cur is not None:
            funcs[cur].append(l)
            if l.startswith(".Lfunc_end"):      # not s_endpgm: a kernel may hold early exits before its body
                cur = None
    if not funcs:
        errors.append("no k_front_s1 instantiation found")
    n_gathers = 0
    for bps, fl in sorted(funcs.items()):
        blocks = asm_blocks(fl)
        for (_, _, body) in blocks:
            for ins in body:
                if re.match(r"(global|buffer|flat)_load", ins):
                    errors.append("k_front_s1<%s>: asm VMEM load found: %s" % (bps, ins))
        gather = [b for b in blocks if b[2] and all(x.startswith(("ds_read_b64", "s_waitcnt", "s_mov_b64", "s_and_b64")) for x in b[2])
                  and any(x.startswith("ds_read_b64") for x in b[2])]
        if len(gather) % 2 != 0 or not gather:
            errors.append("k_front_s1<%s>: %d tap-gather asm blocks (expected an even, non-zero number)" % (bps, len(gather)))
            continue
        for g in gather:
            n_gathers += 1
            if sum(1 for x in g[2] if x.startswith("ds_read_b64")) != 14:
                errors.append("k_front_s1<%s>: a gather block does not hold 14 reads" % bps)
            ex = [x for x in g[2] if x.startswith(("s_mov_b64", "s_and_b64"))]   # EXEC = hit mask per slot, restored before the wait
            if ex and not (re.match(r"s_mov_b64 s\[\d+:\d+\], exec", ex[0]) and re.match(r"s_mov_b64 exec, s\[\d+:\d+\]", ex[-1])
                           and g[2][-2] == ex[-1]):
                errors.append("k_front_s1<%s>: a gather block does not save / restore EXEC around its reads" % bps)
            if not g[2][-1].startswith("s_waitcnt lgkmcnt(0)"):
                errors.append("k_front_s1<%s>: a gather block does not end with s_waitcnt lgkmcnt(0)" % bps)
            if any(x.startswith("s_waitcnt") for x in g[2][:-1]):
                errors.append("k_front_s1<%s>: unexpected wait inside a gather block" % bps)
    return errors, n_gathers


def check_no_read2_b32(lines, what):
    errors, cur = [], None
    for l in lines:
        m = re.match(r"(_ZN5iqgpu\d+k_(?:front_s1|cascade2?)I\w+):", l)
        if m:
            cur = m.group(1)
        elif l.startswith(".Lfunc_end"):
            cur = None
        elif cur and re.match(r"\s*ds_read2(st64)?_b32", l):
            errors.append("%s: %s holds a ds_read2_b32 (a trimmed, re-chunked window load?)" % (what, cur))
            cur = None
    return errors


def check_no_scratch(lines, src, kernel, max_vgpr):
    """every instantiation of `kernel` in `src`: no scratch, at most max_vgpr registers (late round 5: k_front_p0's tail for a
    build-switched buffer count spilled up to 60 registers in the AGC variants for a round of commits -- no parity test sees that)"""
    errors, n = [], 0
    text = "\n".join(lines)
    for m in re.finditer(r"\.amdhsa_kernel (_ZN5iqgpu\d+%sI\w+)\n(.*?)\.end_amdhsa_kernel" % kernel, text, re.S):
        n += 1
        v = int(re.search(r"\.amdhsa_next_free_vgpr (\d+)", m.group(2)).group(1))
        sc = int(re.search(r"\.amdhsa_private_segment_fixed_size (\d+)", m.group(2)).group(1))
        if sc != 0:
            errors.append("%s: %s spills %d bytes to scratch" % (src, m.group(1), sc))
        if v > max_vgpr:
            errors.append("%s: %s needs %d VGPRs (budget %d)" % (src, m.group(1), v, max_vgpr))
    if n == 0:
        errors.append("%s: no instantiation of %s" % (src, kernel))
    return errors


def check_scratch_at_most(lines, src, kernel, max_vgpr, max_scratch):
    """every instantiation of `kernel` in `src`: at most max_vgpr registers and max_scratch bytes of scratch"""
    errors = []
    text = "\n".join(lines)
    n = 0
    for m in re.finditer(r"\.amdhsa_kernel (\S*%s\S*)\n(.*?)\.end_amdhsa_kernel" % re.escape(kernel), text, re.S):
        n += 1
        v = int(re.search(r"\.amdhsa_next_free_vgpr (\d+)", m.group(2)).group(1))
        sc = int(re.search(r"\.amdhsa_private_segment_fixed_size (\d+)", m.group(2)).group(1))
        if sc > max_scratch:
            errors.append("%s: %s spills %d bytes to scratch (bound %d)" % (src, m.group(1), sc, max_scratch))
        if v > max_vgpr:
            errors.append("%s: %s needs %d VGPRs (budget %d)" % (src, m.group(1), v, max_vgpr))
    if n == 0:
        errors.append("%s: no instantiation of %s" % (src, kernel))
    return errors


def check_fat_mid(src, kernel, max_vgpr, lines=None):
    """k_front_fat / k_front_mid are plain C++ whose speed hangs on what hipcc makes of it: (5) no scratch (a spill inside the
    tile loop is a memory round trip per tile), (6) the VGPR count that the occupancy they are built for allows (8 waves per CU:
    256, 12 waves: 168), (7) the tap reads stay single ds_read_b64 -- fused into ds_read2_b64 / ds_read2st64_b64 they run at
    half rate (the tap planes are 2056 bytes apart so that they cannot be fused: a change of that layout shows up here), (8) no
    ds_read2_b32 (a trimmed, re-chunked window load); k_front_mid only: (9) the tile claim of the run-stealing scheme is a returning
    global_atomic_add_x2 issued by one lane whose result is NOT waited for on the spot (the atomic optimizer's wave reduction --
    s_bcnt1 + an immediate s_waitcnt vmcnt(0) + readfirstlane -- would put a memory round trip in front of every tile), and the
    tile loop that holds it is a scalar loop (a divergent run bound turns it into v_cmp / EXEC-mask control: s_andn2_b64 exec)."""
    lines = lines if lines is not None else compile_isa(src)
    errors, cur, n = [], None, 0
    if kernel == "k_front_mid":
        errors += check_claims(lines)
    for l in lines:
        m = re.match(r"(_ZN5iqgpu\d+%sI\w+):" % kernel, l)
        if m:
            cur = m.group(1); n += 1
        elif l.startswith(".Lfunc_end"):
            cur = None
        elif cur:
            t = l.strip()
            if re.match(r"ds_read2(st64)?_b(32|64)", t):
                errors.append("%s: %s holds a %s" % (kernel, cur, t.split()[0])); cur = None
            elif t.startswith("scratch_"):
                # (the multi-run instantiations of k_front_mid -- last template argument true: run stealing, fixed-length runs --
                #  were exempt until the NCO phase left the tile loop's vector multiplies: they hold no scratch either now)
                errors.append("%s: %s spills to scratch" % (kernel, cur))
                cur = None
    for m in re.finditer(r"\.amdhsa_kernel (_ZN5iqgpu\d+%sI\w+)\n(.*?)\.end_amdhsa_kernel" % kernel, "\n".join(lines), re.S):
        v = re.search(r"\.amdhsa_next_free_vgpr (\d+)", m.group(2))
        if v and int(v.group(1)) > max_vgpr:
            errors.append("%s: %s needs %s VGPRs (more than %d: a wave per SIMD lost)" % (kernel, m.group(1), v.group(1), max_vgpr))
    if n == 0:
        errors.append("no %s instantiation found" % kernel)
    return errors


def check_claims(lines):
    errors, cur, body = [], None, []
    funcs = {}
    for l in lines:
        m = re.match(r"(_ZN5iqgpu11k_front_midI\w+):", l)
        if m:
            cur = m.group(1); funcs[cur] = []
        elif l.startswith(".Lfunc_end"):
            cur = None
        elif cur:
            t = l.strip()
            if t and not t.startswith((";", ".")) or t.startswith(".LBB"):
                funcs[cur].append(t)
    for name, ins in funcs.items():
        if not re.search(r"ELb1ELb[01](ELi\d+ELi\d+)?EEEvNS_9FrontArgsE$", name):       # STEAL (the template's last bool but one; CF32OUT, INF, OUT8 follow it) = false: no claims at all
            if any(t.startswith("global_atomic_add_x2") and t.endswith("sc0") for t in ins):
                errors.append("k_front_mid: %s (no stealing) holds a returning atomic add" % name)
            continue
        claims = [i for i, t in enumerate(ins) if t.startswith("global_atomic_add_x2") and t.endswith("sc0") and "off" not in t.split()[1:4]]
        if len(claims) < 2:
            errors.append("k_front_mid: %s holds %d returning tile claims (expected one in front of the first tile and one in the loop)" % (name, len(claims)))
        for i in claims:
            nxt = [t for t in ins[i + 1:i + 8] if not t.startswith(".LBB")]
            if any(t.startswith("s_waitcnt vmcnt(0)") for t in nxt[:4]) and any(t.startswith("v_readfirstlane") for t in nxt[:6]):
                errors.append("k_front_mid: %s waits for its tile claim on the spot (atomic optimizer on?)" % name)
        if any(t.startswith("s_bcnt1_i32_b64") for t in ins):
            errors.append("k_front_mid: %s holds a wave-reduced atomic (s_bcnt1_i32_b64): build with -amdgpu-atomic-optimizer-strategy=None" % name)
        if claims:
            # the loop around the last claim: from the nearest label above to the first backward branch below
            i = claims[-1]
            seg = ins[i:i + 900]
            if any(t.startswith("s_andn2_b64 exec, exec") for t in seg[:700]) and not any(t.startswith("s_cbranch_vccnz") or t.startswith("s_cbranch_scc") for t in seg[:700]):
                errors.append("k_front_mid: %s runs its tile loop under EXEC-mask control (a divergent run bound)" % name)
    return errors


# the instantiations the bench configs run, with the register budget of their launch bounds: a spill in one of them halves its speed
# without failing a single parity test (round 3: a 64-bit division inlined into every kernel's prologue cost the last-stage
# instantiation 968 bytes of scratch and config 3 a quarter of its throughput before a profile showed it)
HOT = {
    "front_wave.hip": [("_ZN5iqgpu10k_front_s1ILi4ELb1ELb0ELb0ELi0EEEvNS_9FrontArgsE", 128),      # headline fallback (16 waves)
                       ("_ZN5iqgpu10k_front_s1ILi4ELb1ELb0ELb0ELi1EEEvNS_9FrontArgsE", 128),      # ... without a mixer
                       ("_ZN5iqgpu10k_front_s1ILi8ELb0ELb0ELb0ELi4EEEvNS_9FrontArgsE", 128),      # last stage behind k_cascade (configs 3, 4)
                       ("_ZN5iqgpu10k_front_s1ILi2ELb0ELb1ELb0ELi2EEEvNS_9FrontArgsE", 128),      # cu8-nrsc5 preset shapes
                       ("_ZN5iqgpu10k_front_s1ILi4ELb0ELb1ELb0ELi3EEEvNS_9FrontArgsE", 128)],
    "cascade_wave.hip": [("_ZN5iqgpu9k_cascadeILi4ELb0ELi1EEEvNS_9FrontArgsE", 168),               # S = 2 calls off a group boundary (12 waves)
                         ("_ZN5iqgpu9k_cascadeILi2ELb1ELi4EEEvNS_9FrontArgsE", 128)],              # config 4 (16 waves)
    "front_s2.hip": [("_ZN5iqgpu10k_front_s2ILi4ELi5ELi1EEEvNS_6S2ArgsE", 168),                    # config 3 (12 waves; its switch set compiled in)
                     ("_ZN5iqgpu10k_front_s2ILi4ELi5ELi0EEEvNS_6S2ArgsE", 168),                    # ... and with run-time switches
                     ("_ZN5iqgpu10k_front_s2ILi2ELi5ELi0EEEvNS_6S2ArgsE", 168)],
}


def check_hot(lines_by_src):
    errors = []
    for src, kernels in HOT.items():
        text = "\n".join(lines_by_src[src])
        for name, max_vgpr in kernels:
            m = re.search(r"\.amdhsa_kernel %s\n(.*?)\.end_amdhsa_kernel" % re.escape(name), text, re.S)
            if not m:
                errors.append("%s: no instantiation %s" % (src, name)); continue
            v = int(re.search(r"\.amdhsa_next_free_vgpr (\d+)", m.group(1)).group(1))
            sc = int(re.search(r"\.amdhsa_private_segment_fixed_size (\d+)", m.group(1)).group(1))
            if sc != 0:
                errors.append("%s: %s spills %d bytes to scratch" % (src, name, sc))
            if v > max_vgpr:
                errors.append("%s: %s needs %d VGPRs (budget %d)" % (src, name, v, max_vgpr))
    return errors


def main():
    from concurrent.futures import ThreadPoolExecutor
    csrc = os.path.join(HERE, "..", "iq_tool_amd", "csrc")
    names = ["front_wave.hip", "cascade_wave.hip", "front_s2.hip", "front_mid.hip", "front_fat.hip", "front_p0.hip", "cascade2.hip", "p0fft_cu8.hip"]
    with ThreadPoolExecutor(max_workers=min(len(names), max(2, (os.cpu_count() or 4) - 1))) as ex:     # (front_mid.hip alone takes 2 min 40 s)
        isa = dict(zip(names, ex.map(lambda nm: compile_isa(os.path.join(csrc, nm)), names)))
    lines = isa["front_wave.hip"]
    errors, n = check(lines)
    errors += check_hot({k: isa[k] for k in ("front_wave.hip", "cascade_wave.hip", "front_s2.hip")})
    errors += check_no_read2_b32(lines, "front_wave.hip")
    errors += check_no_read2_b32(isa["cascade_wave.hip"], "cascade_wave.hip")
    errors += check_no_read2_b32(isa["cascade2.hip"], "cascade2.hip")
    errors += check_fat_mid(os.path.join(csrc, "front_mid.hip"), "k_front_mid", 168, isa["front_mid.hip"])
    errors += check_fat_mid(os.path.join(csrc, "front_fat.hip"), "k_front_fat", 256, isa["front_fat.hip"])
    errors += check_no_scratch(isa["front_p0.hip"], "front_p0.hip", "k_front_p0", 168)       # (round 6: three waves per SIMD)
    errors += check_no_scratch(isa["cascade2.hip"], "cascade2.hip", "k_cascade2", 168)
    # k_p0fft16 (opt-in, round 6): two waves per SIMD; its transforms spill a few dozen registers under that cap (profiles/r06_fused_filter.md)
    # -- bounded here so that a change of the window fill does not silently double it
    errors += check_scratch_at_most(isa["p0fft_cu8.hip"], "p0fft_cu8.hip", "k_p0fft16", 256, 256)
    for e in errors:
        print("FAIL", e)
    print("check_isa: %d tap gathers checked: %s" % (n, "ok" if not errors and n > 0 else "FAILED"))
    return 0 if (not errors and n > 0) else 1


if __name__ == "__main__":
    sys.exit(main())
