"""CPU tests of the host side: the C ABI library loads and exports everything include/iqgpu.h
declares, struct layouts match the ctypes mirror, the create-time design path equals the oracle's,
errors carry the reference's codes, the product never touches oracle/, the hand-written asm passes
its static check, and the N > 1 shard/timing logic of bench.py works over gloo (world_size 2).
No compute entry point is called here (there is no GPU)."""
import ctypes as C
import json
import os
import re
import subprocess
import sys
import textwrap

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "iqgpu.h")


@pytest.fixture(scope="module")
def lib():
    import iq_tool_amd
    return iq_tool_amd.load()


def declared_symbols():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(iqgpu_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol(lib):
    from iq_tool_amd import _lib
    names = declared_symbols()
    assert len(names) >= 38
    out = subprocess.run(["nm", "-D", "--defined-only", _lib.LIB_PATH], capture_output=True, text=True, check=True).stdout
    exported = set(re.findall(r" T (iqgpu_[a-z0-9_]+)", out))
    missing = [n for n in names if n not in exported]
    assert not missing, missing
    bound = {n for n, _, _ in _lib.SYMBOLS}
    assert set(names) == bound, (set(names) ^ bound)
    hdr = open(os.path.join(ROOT, "include", "iqgpu.h")).read()
    assert lib.iqgpu_abi_version() == int(re.search(r"#define\s+IQGPU_ABI_VERSION\s+(\d+)", hdr).group(1))


def test_struct_layouts_match_ctypes(tmp_path):
    from iq_tool_amd import _lib
    prog = textwrap.dedent("""
        #include <stdio.h>
        #include <stddef.h>
        #include "iqgpu.h"
        int main(void) {
            printf("%zu %zu %zu %zu %zu %zu %zu %zu %zu %zu\\n", sizeof(iqgpu_chain_desc), offsetof(iqgpu_chain_desc, shift_hz),
                   offsetof(iqgpu_chain_desc, filters), offsetof(iqgpu_chain_desc, block_samples),
                   sizeof(iqgpu_chain_info), offsetof(iqgpu_chain_info, arb_step), sizeof(iqgpu_profile),
                   offsetof(iqgpu_chain_desc, agc_chunk_frames), sizeof(iqgpu_agc_state), offsetof(iqgpu_agc_state, samples_seen));
            return 0;
        }""")
    src = tmp_path / "layout.c"
    src.write_text(prog)
    exe = tmp_path / "layout"
    subprocess.run(["gcc", "-std=c99", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)], check=True)
    got = [int(x) for x in subprocess.run([str(exe)], capture_output=True, text=True, check=True).stdout.split()]
    D, I, P, A = _lib.ChainDesc, _lib.ChainInfo, _lib.Profile, _lib.AgcState
    want = [C.sizeof(D), D.shift_hz.offset, D.filters.offset, D.block_samples.offset, C.sizeof(I), I.arb_step.offset, C.sizeof(P),
            D.agc_chunk_frames.offset, C.sizeof(A), A.samples_seen.offset]
    assert got == want


def probe(lib, **kw):
    from iq_tool_amd import _lib
    from iq_tool_amd.chain import make_desc
    d = make_desc(**kw)
    info = _lib.ChainInfo()
    ft = np.zeros(2 * 20000, np.float32)
    hb = np.zeros(4096, np.float32)
    arb = np.zeros(3584, np.float32)
    rc = lib.iqgpu_design_probe(C.byref(d), C.byref(info), ft.ctypes.data_as(C.c_void_p), 20000,
                                hb.ctypes.data_as(C.c_void_p), 4096, arb.ctypes.data_as(C.c_void_p), 3584)
    return rc, info, ft.view(np.complex64), hb, arb


@pytest.mark.parametrize("rates", [(2.4e6, 744187.5), (10e6, 2.4e6), (61.44e6, 1488375.0), (2.4e6, 2.0e6), (2.4e6, 2.4e6), (8e6, 8e3)])
def test_resampler_design_equals_oracle(lib, oracle, rates):
    rc, info, _, hb, arb = probe(lib, input_rate_hz=rates[0], target_rate_hz=rates[1], shift_hz=200e3)
    assert rc == 0
    r = np.float32(rates[1] / rates[0])
    m = oracle.MsResamp(r)
    assert info.ratio == r and info.interp == 0
    assert info.num_halfband_stages == m.S and info.arb_step == m.step and info.rate_arb == m.rate_arb
    assert [info.stage_m[k] for k in range(m.S)] == [m.stage_m(k) for k in range(m.S)]
    o = 0
    for k in range(m.S):
        t = m.stage_taps(k)
        assert np.array_equal(hb[o:o + t.size], t)
        o += t.size
    assert np.array_equal(arb, m.arb_proto())
    nco = oracle.Nco(np.float32(2 * np.pi * 200e3 / rates[0]))
    assert info.nco_dtheta == nco.dtheta_u32


def test_filter_design_equals_oracle(lib, oracle):
    cases = [
        (dict(input_rate_hz=10e6, target_rate_hz=2.4e6, filters=(("passband", 158.5e3, 113e3),), filter_taps=1024),
         oracle.make_filter_cfg((("passband", 158.5e3, 113e3),), filter_taps=1025), 10e6, 2.4e6, False),
        (dict(input_rate_hz=61.44e6, target_rate_hz=1488375.0, filters=(("lowpass", 300e3, 0.0),), filter_taps=4097, filter_impl="fir"),
         oracle.make_filter_cfg((("lowpass", 300e3, 0.0),), filter_taps=4097, impl="fir"), 61.44e6, 1488375.0, False),
        (dict(input_rate_hz=2.4e6, no_resample=True, filters=(("highpass", 100e3, 0.0), ("stopband", 400e3, 50e3), ("lowpass", 900e3, 0.0))),
         oracle.make_filter_cfg((("highpass", 100e3, 0.0), ("stopband", 400e3, 50e3), ("lowpass", 900e3, 0.0))), 2.4e6, 2.4e6, True),
        (dict(input_rate_hz=2.4e6, no_resample=True, filters=(("passband", -300e3, 100e3),), transition_width_hz=20e3, attenuation_db=70.0, filter_impl="fft", fft_size=2048),
         oracle.make_filter_cfg((("passband", -300e3, 100e3),), transition_width_hz=20e3, attenuation_db=70.0, impl="fft", fft_size=2048), 2.4e6, 2.4e6, True),
    ]
    for kw, cfg, fin, fout, nores in cases:
        rc, info, ft, _, _ = probe(lib, **kw)
        assert rc == 0, lib.iqgpu_last_error()
        f = oracle.Filter(cfg, fin, fout, no_resample=nores)
        assert (bool(info.filter_post_resample), info.filter_impl, info.filter_ntaps, info.filter_block) == (f.post, f.impl, f.ntaps, f.block)
        assert np.array_equal(ft[:f.ntaps].view(np.float32), f.taps().view(np.float32))


def test_dc_alpha_and_history(lib):
    rc, info, *_ = probe(lib, input_rate_hz=10e6, target_rate_hz=2.4e6, dc_block=True)
    assert rc == 0
    assert info.dc_alpha == np.float32(2.0 * np.pi * 10.0 / 10e6)
    assert info.history_samples >= 2 * (2 * 13 + 40) + 20          # two stages: 2*(2*13+4*10)+4*5


def test_error_codes_follow_the_reference_fatal_paths(lib):
    def code(**kw):
        rc, *_ = probe(lib, **kw)
        return rc, lib.iqgpu_last_error().decode()
    assert code(input_rate_hz=2.4e6, target_rate_hz=100.0)[0] == -4            # ratio < 1e-3  (src/setup.c:109)
    assert code(input_rate_hz=2.4e6, target_rate_hz=2.4e9 * 2)[0] == -4
    assert code(in_format=3)[0] == -5                                          # real scalar format: unhandled
    assert code(out_format=0)[0] == -5
    rc, msg = code(shift_hz=0.0, shift_after_resample=True)
    assert rc == -6 and "shift-after-resample" in msg
    assert code(shift_hz=2.4e6 * 5.1)[0] == -6                                 # SHIFT_FACTOR_LIMIT
    rc, msg = code(filters=(("lowpass", 600e3, 0.0),))
    assert rc == -7 and "Nyquist" in msg
    assert code(no_resample=True, filters=(("passband", 50e3, 20e3),), filter_taps=1025, fft_size=1024)[0] == -7
    assert code(input_rate_hz=2.4e6, target_rate_hz=4.8e6)[0] == 0             # interpolation
    assert code(input_rate_hz=2.4e6, target_rate_hz=2.4e6, filters=(("lowpass", 100e3, 0.0),))[0] == 0   # filter, then r = 1
    assert code(block_samples=1000)[0] == -1


def test_create_without_device_fails_loudly(lib):
    import iq_tool_amd
    if lib.iqgpu_device_count() > 0:
        pytest.skip("a GPU is visible")
    with pytest.raises(iq_tool_amd.IqgpuError) as e:
        iq_tool_amd.Chain(shift_hz=200e3)
    assert e.value.code == -2
    from iq_tool_amd import ops
    with pytest.raises(iq_tool_amd.IqgpuError):
        ops.convert_block_to_cf32(np.zeros(8, np.int16), "cs16")
    assert ops.get_bytes_per_sample("cs16") == 4        # pure table, no device needed


def test_missing_library_raises(monkeypatch, tmp_path):
    env = dict(os.environ, IQGPU_LIB=str(tmp_path / "nope.so"), PYTHONPATH=ROOT)
    r = subprocess.run([sys.executable, "-c", "import iq_tool_amd; iq_tool_amd.load()"], env=env, capture_output=True, text=True)
    assert r.returncode != 0 and "no CPU fallback" in r.stderr


def test_product_never_touches_the_oracle():
    pkg = os.path.join(ROOT, "iq_tool_amd")
    bad = []
    for dp, _, fns in os.walk(pkg):
        for fn in fns:
            if fn.endswith((".py", ".cpp", ".hpp", ".hip", ".h", ".c")):
                txt = open(os.path.join(dp, fn), errors="replace").read()
                if re.search(r"pyoracle|iq_oracle|liboracle|from oracle|import oracle|oracle/", txt):
                    bad.append(os.path.join(dp, fn))
    assert not bad, bad
    hdr = open(HEADER).read()
    assert "oracle" not in hdr


def test_hand_written_asm_static_check():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "check_isa.py")], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr


def _gather_model(step, nl, fold):
    """numpy restatement of tap_gather_cycles (front_mid.hip): cycles per 8-byte tap read of a half-wave = the fullest of the 32
    bank pairs, counted in distinct entries"""
    ns = 4 if nl == 6 else 5
    lo = [(step * j) >> 24 for j in range(5)]
    total = reads = 0
    for t in range(48):
        base = 64 * nl * t * 7
        lanes = np.arange(64, dtype=np.int64)
        g = (base + nl * lanes) << 24
        k = (g + step - 1) // step
        P = (k * step - g).astype(np.int64)
        for j in range(ns):
            Pj = (P + j * step) & 0xFFFFFFFF
            x = Pj >> 16
            if fold:
                x = x ^ ((Pj >> 21) & 7)
            e = (x + (Pj >> 24) - 257 * lo[j]) & 0xFFFFFFFF
            for h in (slice(0, 32), slice(32, 64)):
                total += int(np.bincount(np.unique(e[h]) & 31, minlength=32).max())
            reads += 1
    return total / reads


@pytest.mark.parametrize("step_over_2_24,nl", [(27053208 / 2 ** 24, 6), (27053208 / 2 ** 24, 8), (1.625, 6), (1.75, 6), (1.5, 6), (1.724, 6),
                                               (1.9967, 6), (1.579, 6), (1.8, 8), (1.923, 8)])
def test_tap_placement_chooser_follows_its_bank_model(lib, step_over_2_24, nl):
    """front_tap_fold (front_mid.hip): the arms of the polyphase table sit in the tap planes in order unless the folded placement
    saves more than 1.5 modelled LDS cycles per tap read for the chain's step -- checked against a numpy restatement of the model,
    and on the two cases the design notes quote (NRSC-5: in order at 6 per lane, folded at 8)."""
    from iq_tool_amd import _lib
    so = C.CDLL(_lib.LIB_PATH)
    f = getattr(so, "_ZN5iqgpu14front_tap_foldEji")
    f.argtypes = [C.c_uint32, C.c_int]; f.restype = C.c_int
    step = int(round(step_over_2_24 * 2 ** 24))
    plain, folded = _gather_model(step, nl, False), _gather_model(step, nl, True)
    want = 1 if folded + 1.5 < plain else 0
    assert f(step, nl) == want, (plain, folded)
    if step == 27053208:
        assert want == (0 if nl == 6 else 1)
    if step_over_2_24 == 1.625:
        assert want == 1 and plain > 12 and folded < 5


def test_next_out_frames_closed_form_matches_oracle_counts(oracle):
    """the count law the host uses: K(Q) = ceil(Q 2^24 / step), Q = floor(N / 2^S)"""
    r = np.float32(744187.5 / 2.4e6)
    m = oracle.MsResamp(r)
    rng = np.random.default_rng(3)
    x = np.zeros(1, np.complex64)
    tot_in = tot_out = 0
    for n in rng.integers(0, 5000, 40):
        got = m.execute(np.zeros(int(n), np.complex64)).size
        tot_in += int(n)
        tot_out += got
        q = tot_in >> m.S
        assert tot_out == -(-(q << 24) // m.step)


# --------------------------------------------------------------------------------------------
# N > 1: independent shards, barrier + max-over-ranks timing (gloo, world_size 2)
# --------------------------------------------------------------------------------------------
WORKER = """
import os, sys, time
sys.path.insert(0, %r)
import torch, torch.distributed as dist
import bench
rank = int(os.environ["RANK"]); world = int(os.environ["WORLD_SIZE"])
dist.init_process_group(backend="gloo", rank=rank, world_size=world)
plan = bench.shard_plan(world, rank, 1 << 20)
assert plan["first_frame"] == rank << 20 and plan["frames"] == 1 << 20 and plan["seed"] == 10 + rank
calls = []
def step():
    calls.append(1)
    time.sleep(0.05 if rank == 0 else 0.15)      # rank 1 is the slow shard
dt = bench.timed_region(dist, lambda: None, step, 3)
assert len(calls) == 3
assert 0.44 <= dt < 2.0, dt                       # the MAX over ranks: 3 x 0.15 s on both ranks
print("rank", rank, "ok", round(dt, 3))
dist.destroy_process_group()
"""


def test_sharding_and_max_reduce_over_gloo(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER % ROOT)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29517", WORLD_SIZE="2")
    procs = [subprocess.Popen([sys.executable, str(script)], env=dict(env, RANK=str(r), LOCAL_RANK=str(r)),
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(2)]
    outs = [p.communicate(timeout=120)[0] for p in procs]
    for p, o in zip(procs, outs):
        assert p.returncode == 0, o
    assert all("ok" in o for o in outs)


def _run_bench_stub(extra_env, argv):
    env = dict(os.environ, IQGPU_BENCH_STUB="1", **extra_env)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT"):
        if k not in extra_env:
            env.pop(k, None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + argv, env=env, stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE, text=True, timeout=180)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, p.stdout                    # rank 0 prints ONE line, nothing else reaches stdout
    return json.loads(lines[0])


def test_bench_gpus_n_spawns_n_ranks_without_a_launcher():
    """`python bench.py --gpus 2` with no WORLD_SIZE starts two ranks itself (gloo barrier + MAX), and the one
    line says n_gpus 2; the step is a stub here (no GPU in this container), the launcher is the real one."""
    d = _run_bench_stub({}, ["--gpus", "2", "--steps", "4", "--warmup", "1", "--log2-frames", "12"])
    assert d["n_gpus"] == 2 and d["steps"] == 4
    # the slow rank (rank 1 sleeps twice as long) sets the time: MAX over ranks
    assert d["ms_per_step"] >= 3.9
    d1 = _run_bench_stub({}, ["--gpus", "1", "--steps", "2", "--log2-frames", "12"])
    assert d1["n_gpus"] == 1


def test_bench_spawn_fails_fast_when_a_rank_dies():
    """a rank that exits before the rendezvous (no GPU for its local rank, an import error) must end the whole
    run within seconds with its exit code -- not leave rank 0 in the gloo rendezvous until its timeout"""
    import time
    env = dict(os.environ, IQGPU_BENCH_STUB="1", IQGPU_BENCH_STUB_FAIL_RANK="1")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k, None)
    t0 = time.monotonic()
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--log2-frames", "12"],
                       env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=120)
    assert p.returncode == 3, (p.returncode, p.stderr[-1000:])
    assert time.monotonic() - t0 < 60
    assert "a rank failed" in p.stderr
    assert not [l for l in p.stdout.splitlines() if l.startswith("{")]      # no result line from a broken run


def test_bench_under_a_launcher_reads_ranks_from_the_environment():
    """the driver's form: python -m torch.distributed.run --nproc-per-node 2 bench.py --gpus 2"""
    env = dict(os.environ, IQGPU_BENCH_STUB="1")
    p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                        "--master-addr", "127.0.0.1", "--master-port", "29533", os.path.join(ROOT, "bench.py"),
                        "--gpus", "2", "--steps", "2", "--log2-frames", "12"], env=env, stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE, text=True, timeout=240)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1 and json.loads(lines[0])["n_gpus"] == 2
    # a mismatch between the launcher's world size and --gpus is an error, not a silent 1-GPU run
    q = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4"],
                       env=dict(env, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT="29534"),
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=60)
    assert q.returncode != 0 and "WORLD_SIZE" in q.stderr


@pytest.mark.parametrize("kw", [
    dict(in_format="cs16", out_format="cs16", input_rate_hz=2.4e6, target_rate_hz=744187.5, shift_hz=200e3),
    dict(in_format="cs16", out_format="cs16", input_rate_hz=10e6, target_rate_hz=2.4e6, filters=(("passband", 158.5e3, 113e3),), filter_taps=1024),
    dict(in_format="cs16", out_format="cs16", input_rate_hz=600e3, target_rate_hz=2.4e6),
    dict(in_format="cs16", out_format="cf32", input_rate_hz=1.0e6, target_rate_hz=2.5e6, filters=(("lowpass", 200e3, 0.0),), filter_taps=257, filter_impl="fft"),
    dict(in_format="cu8", out_format="cu8", input_rate_hz=2.0e6, target_rate_hz=2.0e6, filters=(("lowpass", 300e3, 0.0),), filter_taps=129, filter_impl="fft"),
    dict(in_format="cu8", out_format="cu8", input_rate_hz=61.44e6, target_rate_hz=1488375.0, filters=(("lowpass", 300e3, 0.0),), filter_taps=4097, filter_impl="fir"),
    dict(in_format="cs16", out_format="cs16", input_rate_hz=2.4e6, target_rate_hz=2.4e6, no_resample=True),
])
def test_design_out_frames_is_the_oracles_count(kw):
    """iqgpu_design_out_frames (what places shard outputs) against the oracle actually running the stream"""
    import iq_tool_amd
    from iq_tool_amd import synth
    from iq_tool_amd.chain import make_desc
    from oracle import pyoracle
    pyoracle.build()
    lib = iq_tool_amd.load()
    d = make_desc(**kw)
    for n in (0, 1, 1023, 4096, 50_001, 131_072 + 5):
        got = C.c_size_t(0)
        assert lib.iqgpu_design_out_frames(C.byref(d), n, C.byref(got)) == 0
        raw = synth.raw_stream(max(n, 1), kw["input_rate_hz"], 1, kw["in_format"])
        bpf = 2 if kw["in_format"] in ("cu8", "cs8") else 4
        out = pyoracle.Chain(**kw).process(raw.view(np.uint8)[:n * bpf])
        obpf = {"cs16": 2, "cu8": 2, "cf32": 2}[kw["out_format"]]
        assert got.value == out.size // obpf, (n, got.value, out.size // obpf)


def test_shard_plan_at_the_real_sizes_of_configs4():
    """BASELINE configs[4] as written: 8 independent 10 GB raw cs16 shards = 2.5 G frames each (beyond 2^31), the NRSC-5
    chain.  The harness places shard s at the sum of iqgpu_design_out_frames of the shards before it (iqgpu_run.c): the
    closed form of the resampler law in Python integers must give the same counts and offsets -- no 32-bit step anywhere
    (src/output_raw_file.c:146-184 writes what it is handed; the counts are resampler.c's)."""
    import iq_tool_amd
    from iq_tool_amd.chain import make_desc
    lib = iq_tool_amd.load()
    kw = dict(in_format="cs16", out_format="cs16", input_rate_hz=2.4e6, target_rate_hz=744187.5, shift_hz=200e3)
    d = make_desc(**kw)
    info = iq_tool_amd._lib.ChainInfo()
    assert lib.iqgpu_design_probe(C.byref(d), C.byref(info), None, 0, None, 0, None, 0) == 0
    S, step = int(info.num_halfband_stages), int(info.arb_step)
    assert S == 1 and step == 27053208                              # SPEC B.6: NRSC-5
    total = 8 * 2_500_000_000 + 12_345                              # a ragged tail on the last shard
    per = total // 8
    offsets, off = [], 0
    for s in range(8):
        frames = total - s * per if s == 7 else per
        got = C.c_size_t(0)
        assert lib.iqgpu_design_out_frames(C.byref(d), frames, C.byref(got)) == 0
        groups = frames >> S                                        # a fresh shard: rem = 0, phi = 0
        want = -(-(groups << 24) // step)                           # ceil(G 2^24 / step), exact in Python ints
        assert got.value == want, (s, frames, got.value, want)
        assert frames > 2**31 and want > 2**29
        offsets.append(off); off += want * 4
    assert offsets[1] == 4 * (-(-((per >> 1) << 24) // step)) and off > 2**34      # byte offsets beyond 16 GiB
    # the whole job as ONE stream emits at most a frame per shard boundary more or fewer (each shard restarts the phase)
    one = C.c_size_t(0)
    assert lib.iqgpu_design_out_frames(C.byref(d), total, C.byref(one)) == 0
    assert abs(one.value - off // 4) <= 8


@pytest.mark.parametrize("kw", [
    dict(input_rate_hz=10e6, target_rate_hz=2.4e6, filters=(("passband", 158.5e3, 113e3),), filter_taps=1024),
    dict(input_rate_hz=61.44e6, target_rate_hz=1488375.0, filters=(("lowpass", 300e3, 0.0),), filter_taps=4097, filter_impl="fir"),
    dict(input_rate_hz=2.4e6, no_resample=True, filters=(("highpass", 100e3, 0.0), ("stopband", 400e3, 50e3), ("lowpass", 900e3, 0.0))),
    dict(input_rate_hz=2.4e6, no_resample=True, filters=(("passband", -300e3, 100e3),), transition_width_hz=20e3, attenuation_db=70.0, filter_impl="fft", fft_size=2048),
    dict(input_rate_hz=2.4e6, target_rate_hz=744187.5, filters=(("lowpass", 120e3, 0.0),)),                      # auto length, DC normalisation, post placement
    dict(input_rate_hz=1.0e6, target_rate_hz=2.5e6, filters=(("lowpass", 200e3, 0.0),), filter_taps=256, filter_impl="fft"),   # even -> odd bump, pre placement
    dict(input_rate_hz=2.4e6, target_rate_hz=1.2e6, filters=(("passband", 0.0, 300e3),)),                        # pass-band centred on 0: real taps, peak normalisation
    dict(input_rate_hz=2.4e6, target_rate_hz=1.2e6, filters=(("stopband", 100e3, 20e3), ("passband", 250e3, 80e3)), attenuation_db=45.0),
    dict(input_rate_hz=4.8e3, target_rate_hz=1.2e3, filters=(("lowpass", 3.0, 0.0),), transition_width_hz=0.2),   # transition floor 1 Hz, long filter
])
def test_filter_design_against_numpy_restatement(lib, kw):
    """iqgpu_design_probe (design.cpp) against tests/np_design.py: placement, odd bump, auto length, chain convolution,
    peak / DC normalisation, implementation choice, block size -- a formulation independent of the C oracle"""
    import np_design
    rc, info, ft, _, _ = probe(lib, **kw)
    assert rc == 0, lib.iqgpu_last_error()
    d = np_design.design(kw["filters"], kw["input_rate_hz"], kw.get("target_rate_hz", kw["input_rate_hz"]), no_resample=kw.get("no_resample", False),
                         filter_taps=kw.get("filter_taps", 0), transition_width_hz=kw.get("transition_width_hz", 0.0),
                         attenuation_db=kw.get("attenuation_db", 0.0), impl=kw.get("filter_impl", "auto"), fft_size=kw.get("fft_size", 0))
    assert bool(info.filter_post_resample) == d["post"]
    assert info.filter_ntaps == d["taps"].size
    assert info.filter_impl == d["impl"] and info.filter_block == d["block"]
    got = ft[:info.filter_ntaps].astype(np.complex128)
    scale = np.abs(d["taps"]).max()
    assert np.abs(got - d["taps"]).max() <= 3e-6 * scale, np.abs(got - d["taps"]).max() / scale


def test_filter_design_fatal_paths_against_numpy_restatement(lib):
    import np_design
    bad = [dict(input_rate_hz=2.4e6, target_rate_hz=744187.5, filters=(("lowpass", 400e3, 0.0),)),               # beyond output Nyquist
           dict(input_rate_hz=2.4e6, target_rate_hz=744187.5, filters=(("passband", 300e3, 200e3),)),
           dict(input_rate_hz=2.4e6, no_resample=True, filters=(("lowpass", 100e3, 0.0),), filter_taps=1001, filter_impl="fft", fft_size=512)]
    for kw in bad:
        rc, *_ = probe(lib, **kw)
        assert rc == -7, kw
        with pytest.raises(ValueError):
            np_design.design(kw["filters"], kw["input_rate_hz"], kw.get("target_rate_hz", kw["input_rate_hz"]), no_resample=kw.get("no_resample", False),
                             filter_taps=kw.get("filter_taps", 0), impl=kw.get("filter_impl", "auto"), fft_size=kw.get("fft_size", 0))


def test_diagnostic_switches_go_through_one_entry_point_not_the_environment(monkeypatch):
    """VERDICT r5 item 7 / ABI v6: the library reads no switch from the environment.  (a) no string of the shared object starts with
    IQGPU_ (what `strings libiqgpu.so | grep -c '^IQGPU_'` counts); (b) iqgpu_debug_set refuses unknown names, sets, lists, clears;
    (c) a switch exported in the environment changes nothing until somebody passes it to iqgpu_debug_set -- which the ctypes mirror
    does, explicitly, for the tests (iq_tool_amd._lib.apply_debug_env); (d) the design path sees the switch (fft_log2n moves the
    overlap-save transform size is device-side; here: the table itself)."""
    import re
    from iq_tool_amd import _lib
    blob = open(_lib.LIB_PATH, "rb").read()
    if os.environ.get("IQGPU_SANITIZED") != "1":          # (tools/sanitize_host.py's build carries debug info: the header's enum names)
        assert not re.findall(rb"(?:^|[^\x20-\x7e])(IQGPU_[A-Z0-9_]{2,})", blob), "an IQGPU_* string is back in libiqgpu.so"
    lib = _lib.load()
    assert lib.iqgpu_debug_set(None, None) == 0 and _lib.debug_switches() == {}
    assert lib.iqgpu_debug_set(b"no_such_switch", b"1") != 0 and b"unknown switch" in lib.iqgpu_last_error()
    assert lib.iqgpu_debug_set(b"no_fast", b"1") == 0 and lib.iqgpu_debug_set(b"fft_log2n", b"12") == 0
    assert _lib.debug_switches() == {"no_fast": "1", "fft_log2n": "12"}
    assert lib.iqgpu_debug_set(b"no_fast", None) == 0 and _lib.debug_switches() == {"fft_log2n": "12"}
    small = C.create_string_buffer(4)
    assert lib.iqgpu_debug_list(small, 4) != 0                      # (too small a buffer is an error, not a truncation)
    assert lib.iqgpu_debug_set(None, None) == 0
    # the environment alone does nothing ...
    monkeypatch.setenv("IQGPU_NO_FAST", "1")
    monkeypatch.setenv("IQGPU_BENCH_STUB", "1")                     # (not a library switch: never forwarded)
    assert _lib.debug_switches() == {}
    # ... the mirror forwards it, and clears what is no longer exported
    _lib.apply_debug_env()
    assert _lib.debug_switches() == {"no_fast": "1"}
    monkeypatch.delenv("IQGPU_NO_FAST")
    _lib.apply_debug_env()
    assert _lib.debug_switches() == {}
    # every name the mirror knows is a name the library knows
    for name in _lib.DEBUG_NAMES:
        assert lib.iqgpu_debug_set(name.encode(), b"1") == 0, name
    assert lib.iqgpu_debug_set(None, None) == 0


# --------------------------------------------------------------------------------------------
# NUMA placement of the process / thread that feeds a GPU (iqgpu_device_numa_node, iqgpu_bind_thread_to_device: topology.cpp)
# against a stand-in sysfs tree shaped like the 8-GPU, two-socket boxes of this pool (IQGPU_SYSFS_ROOT)
# --------------------------------------------------------------------------------------------
def _fake_sysfs(root, gpus):
    """gpus: [(bus, numa_node, cpulist)]; KFD nodes 0, 1 are the two CPU sockets, the GPUs follow in this order"""
    def put(path, text):
        os.makedirs(os.path.dirname(path), exist_ok=True)
        with open(path, "w") as fh:
            fh.write(text)
    nodes = os.path.join(root, "class", "kfd", "kfd", "topology", "nodes")
    for n in range(2):
        put(os.path.join(nodes, str(n), "properties"), "cpu_cores_count 64\nsimd_count 0\nlocation_id 0\ndomain 0\n")
    for i, (bus, node, cpus) in enumerate(gpus):
        put(os.path.join(nodes, str(2 + i), "properties"), "cpu_cores_count 0\nsimd_count 1024\nlocation_id %d\ndomain 0\n" % (bus << 8))
        dev = os.path.join(root, "bus", "pci", "devices", "0000:%02x:00.0" % bus)
        put(os.path.join(dev, "numa_node"), "%d\n" % node)
        put(os.path.join(dev, "local_cpulist"), cpus + "\n")


def test_numa_binding_follows_the_device_through_a_stand_in_sysfs(tmp_path):
    allowed = sorted(os.sched_getaffinity(0))
    if len(allowed) < 2:
        pytest.skip("needs two CPUs in the affinity mask")
    half = len(allowed) // 2
    lo, hi = allowed[:half], allowed[half:]
    fmt = lambda cs: ",".join(str(c) for c in cs)       # noqa: E731
    # four GPUs: two next to the first half of this machine's CPUs ("node 0"), two next to the second ("node 1"); the last entry's
    # cpulist is a range that also names CPUs this process may not use (the binding intersects with the current mask)
    gpus = [(0x0d, 0, fmt(lo)), (0x26, 0, fmt(lo)), (0x8e, 1, fmt(hi)), (0xa7, 1, "%d-%d" % (hi[0], hi[-1] + 400))]
    _fake_sysfs(str(tmp_path), gpus)
    prog = textwrap.dedent("""
        import json, os, sys
        sys.path.insert(0, %r)
        import iq_tool_amd
        out = {}
        for o in (0, 3, 1, 7):
            node, bdf, err = iq_tool_amd.bind_thread_to_device(o)
            out[str(o)] = dict(node=node, bdf=bdf, err=err, cpus=sorted(os.sched_getaffinity(0)))
        print(json.dumps(out))
    """ % ROOT)
    def run(**env_extra):
        env = dict(os.environ, IQGPU_SYSFS_ROOT=str(tmp_path), **env_extra)
        for k in ("ROCR_VISIBLE_DEVICES", "HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
            if k not in env_extra:
                env.pop(k, None)
        p = subprocess.run([sys.executable, "-c", prog], env=env, capture_output=True, text=True)
        assert p.returncode == 0, p.stderr
        return json.loads(p.stdout.strip().splitlines()[-1])
    r = run()
    assert r["0"] == dict(node=0, bdf="0000:0d:00.0", err=None, cpus=lo)
    # ordinal 3 next: node 1's CPUs are no longer in the mask the first call left -> refused, mask untouched, the error says why
    assert r["3"]["node"] == 1 and r["3"]["bdf"] == "0000:a7:00.0" and "affinity mask" in r["3"]["err"] and r["3"]["cpus"] == lo
    assert r["1"] == dict(node=0, bdf="0000:26:00.0", err=None, cpus=lo)
    assert r["7"]["node"] == -1 and "out of range" in r["7"]["err"]
    # device selection the way the runtime applies it: ROCR first, HIP on top of it; ordinal 0 is then the GPU on bus a7
    r = run(ROCR_VISIBLE_DEVICES="1,2,3", HIP_VISIBLE_DEVICES="2,0")
    assert r["0"] == dict(node=1, bdf="0000:a7:00.0", err=None, cpus=hi)
    assert r["1"]["bdf"] == "0000:26:00.0" and r["3"]["node"] == -1
    # UUID lists are not interpreted: nothing is bound and the caller is told
    r = run(ROCR_VISIBLE_DEVICES="GPU-deadbeef")
    assert r["0"]["node"] == -1 and "not a list of indices" in r["0"]["err"] and r["0"]["cpus"] == allowed


def _two_socket_node(tmp_path):
    """stand-in sysfs of an 8-GPU, two-socket box: four GPUs next to the first half of this machine's CPUs, four next to the second"""
    allowed = sorted(os.sched_getaffinity(0))
    half = len(allowed) // 2
    lo, hi = allowed[:half], allowed[half:]
    fmt = lambda cs: ",".join(str(c) for c in cs)       # noqa: E731
    buses = [0x0d, 0x26, 0x43, 0x5b, 0x8a, 0xa7, 0xc4, 0xdc]
    _fake_sysfs(str(tmp_path), [(b, 0 if i < 4 else 1, fmt(lo if i < 4 else hi)) for i, b in enumerate(buses)])
    return buses, lo, hi


def test_bench_dry_placement_eight_ranks_two_sockets(tmp_path):
    """`bench.py --gpus 8 --dry-placement` does everything the first 8-GPU run does except GPU work -- eight fresh
    rank processes, each bound to the NUMA node of ITS GPU from (stand-in) sysfs before anything else, the gloo rendezvous, the
    device table gathered on rank 0 with eight distinct PCI addresses, buffer sizes, barrier + MAX on an empty step.  World size 8
    on the CPU; and the same line must FAIL (exit 4) when two ranks would land on one GPU."""
    allowed = sorted(os.sched_getaffinity(0))
    if len(allowed) < 2:
        pytest.skip("needs two CPUs in the affinity mask")
    buses, lo, hi = _two_socket_node(tmp_path)
    env = dict(os.environ, IQGPU_SYSFS_ROOT=str(tmp_path), IQGPU_BENCH_RDZV_S="120")
    for k in ("ROCR_VISIBLE_DEVICES", "HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES", "WORLD_SIZE", "RANK", "LOCAL_RANK", "IQGPU_BENCH_STUB"):
        env.pop(k, None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "3", "--dry-placement"], env=env,
                       capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-3000:]
    line = json.loads(p.stdout.strip().splitlines()[-1])
    assert line["ok"] and line["n_gpus"] == 8 and line["distinct_gpus"] == 8 and line["problems"] == []
    assert [d["rank"] for d in line["devices"]] == list(range(8)) and [d["ordinal"] for d in line["devices"]] == list(range(8))
    assert [d["numa_pci_bus_id"] for d in line["devices"]] == ["0000:%02x:00.0" % b for b in buses]
    assert [d["numa_node"] for d in line["devices"]] == [0] * 4 + [1] * 4
    assert all(d["numa_bound"] for d in line["devices"])
    assert all(d["cpus"] == (lo if d["rank"] < 4 else hi) for d in line["devices"])
    assert len({d["pid"] for d in line["devices"]}) == 8                                   # eight processes
    assert line["frames_per_step_per_gpu"] == 1 << 28 and line["devices"][0]["hbm_bytes"] > 1 << 30
    assert line["out_frames_per_step_per_gpu"] == -(-((1 << 27) << 24) // 27053208)       # the closed form, NRSC-5 (SPEC B.6)
    # a device selection that maps two ranks onto one GPU: refused, with the reason
    env2 = dict(env, HIP_VISIBLE_DEVICES="0,0,1,2,3,4,5,6")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "1", "--dry-placement"], env=env2,
                       capture_output=True, text=True, timeout=600)
    assert p.returncode != 0
    if p.stdout.strip():
        line = json.loads(p.stdout.strip().splitlines()[-1])
        assert not line["ok"] and any("distinct" in q for q in line["problems"]), line["problems"]


def test_harness_dry_placement_eight_shards_on_eight_devices(tmp_path):
    """... and `iqgpu_run --shards 8 --devices 8 --dry-placement` (BASELINE configs[4]'s shape: 8 x 2.5 G cs16 frames): the shard plan
    with its byte offsets beyond 16 GiB, every shard thread bound to its device's node, its device's PCI address, what it would pin --
    with no GPU call (this machine has no GPU and the command succeeds)."""
    import iq_tool_amd
    from iq_tool_amd.build import HARNESS_BIN
    if len(os.sched_getaffinity(0)) < 2:
        pytest.skip("needs two CPUs in the affinity mask")
    buses, lo, hi = _two_socket_node(tmp_path)
    env = dict(os.environ)
    for k in ("ROCR_VISIBLE_DEVICES", "HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        env.pop(k, None)
    total = 8 * 2_500_000_000
    cmd = [HARNESS_BIN, "--synthetic", str(total), "--synthetic-hash", "7", "--raw-file-input-rate", "2400000", "--raw-file-input-sample-format", "cs16",
           "--output-rate", "744187.5", "--output-sample-format", "cs16", "--freq-shift", "200000", "--shards", "8", "--devices", "8",
           "--dry-placement", "--debug", "sysfs_root=" + str(tmp_path)]
    p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=120)
    assert p.returncode == 0, p.stderr
    r = json.loads(p.stdout.strip().splitlines()[-1])
    assert r["dry_placement"] and r["shards"] == 8 and r["distinct_devices"] == 8
    off = 0
    for s, sh in enumerate(r["per_shard"]):
        assert (sh["shard"], sh["device"], sh["pci_bus_id"]) == (s, s, "0000:%02x:00.0" % buses[s])
        assert sh["numa_node"] == (0 if s < 4 else 1) and sh["cpus_allowed"] == len(lo if s < 4 else hi)
        assert sh["first_frame"] == s * 2_500_000_000 and sh["frames_in"] == 2_500_000_000
        assert sh["planned_out"] == iq_tool_amd.design_out_frames(2_500_000_000, in_format="cs16", out_format="cs16", input_rate_hz=2.4e6,
                                                                  target_rate_hz=744187.5, shift_hz=200e3)
        assert sh["out_offset_bytes"] == off and sh["pinned_bytes"] > 2 * 4194304 * 4
        off += sh["planned_out"] * 4
    assert off > 1 << 34 and r["frames_out"] == off // 4
    # sixteen shards round-robin over the eight devices: two per device, still eight distinct
    cmd[cmd.index("--shards") + 1] = "16"
    p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=120)
    assert p.returncode == 0, p.stderr
    r = json.loads(p.stdout.strip().splitlines()[-1])
    assert r["distinct_devices"] == 8 and [sh["device"] for sh in r["per_shard"]] == [s % 8 for s in range(16)]


def test_agc_chunk_of_output_inverts_the_chunk_map(tmp_path):
    """kernels.hpp agc_chunk_of_output (the closed form the fused AGC epilogue of k_fftconv16 finds its chunk with) against a scan of
    agc_out_end over random geometries: resampler phases and steps, open decimation groups, block-quantised filters with samples
    pending, chunk lengths that are no power of two, calls that end inside a chunk -- every output of every call."""
    src = tmp_path / "inv.cpp"
    src.write_text(textwrap.dedent(r"""
        #include <cstdio>
        #include <cstdlib>
        #include "kernels.hpp"
        using namespace iqgpu;
        int main() {
            unsigned long long checked = 0;
            srand(5);
            for (int trial = 0; trial < 400; ++trial) {
                AgcGeom g{};
                g.mode = trial % 5 == 0 ? 0 : 1;
                g.S = g.mode ? rand() % 4 : 0;
                g.rem = g.mode ? rand() % (1 << g.S) : 0;
                g.step = (uint32_t)((1.0 + (rand() % 1000) / 1000.0 * 0.999) * 16777216.0);
                g.phi = (uint64_t)(rand() % 1000) * g.step / 1000;
                g.block = trial % 3 == 0 ? 0u : (64u << (rand() % 4));
                g.fpending = g.block ? (uint64_t)(rand() % g.block) : 0;
                g.chunk_frames = 900 + rand() % 20000;
                g.frames_in = 1 + rand() % 200000;
                g.n_chunks = (int)((g.frames_in + g.chunk_frames - 1) / g.chunk_frames);
                const int64_t n_emit = agc_out_end(g, g.n_chunks - 1);
                int c = 0;
                for (int64_t k = 0; k < n_emit; ++k) {
                    while (agc_out_end(g, c) <= k) ++c;                    // the scan: smallest c with out_end(c) > k
                    const int64_t got = agc_chunk_of_output(g, k);
                    if (got != c) { printf("trial %d k %lld: %lld != %d\n", trial, (long long)k, (long long)got, c); return 1; }
                    ++checked;
                }
            }
            printf("ok %llu\n", checked);
            return 0;
        }"""))
    exe = tmp_path / "inv"
    subprocess.run(["g++", "-O2", "-std=c++17", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include", "-I", os.path.join(ROOT, "iq_tool_amd", "csrc"),
                    str(src), "-o", str(exe)], check=True)
    out = subprocess.run([str(exe)], capture_output=True, text=True)
    assert out.returncode == 0 and out.stdout.startswith("ok "), out.stdout + out.stderr
    assert int(out.stdout.split()[1]) > 1_000_000
