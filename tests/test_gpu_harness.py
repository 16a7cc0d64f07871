"""The raw-file harness (iq_tool_amd/csrc/harness/iqgpu_run.c): reader -> chain -> writer with
double-buffered pinned copies, and N independent file-range shards stitched at the writer."""
import json
import os
import subprocess

import numpy as np
import pytest

from iq_tool_amd import synth

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "iq_tool_amd", "lib", "iqgpu_run")
NRSC5 = dict(in_format="cs16", out_format="cs16", input_rate_hz=2.4e6, target_rate_hz=744187.5, shift_hz=200e3)
ARGS = ["--raw-file-input-rate", "2.4e6", "--raw-file-input-sample-format", "cs16", "--output-rate", "744187.5",
        "--output-sample-format", "cs16", "--freq-shift", "200e3"]


def run(*args):
    r = subprocess.run([EXE, *args], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr + r.stdout
    return json.loads(r.stdout.strip().splitlines()[-1])


def test_harness_matches_chain_and_oracle(gpu, oracle, tmp_path):
    n = 3_000_001
    raw = synth.raw_stream(n, 2.4e6, 1, "cs16")
    fin, fout = tmp_path / "in.cs16", tmp_path / "out.cs16"
    raw.tofile(fin)
    info = run("-i", str(fin), "-o", str(fout), *ARGS, "--chunk-frames", "262144")
    got = np.fromfile(fout, np.int16)
    want_gpu = gpu.Chain(**NRSC5).process(raw)
    assert info["frames_in"] == n and info["frames_out"] * 2 == got.size
    assert np.array_equal(got, want_gpu)                       # chunked + double-buffered == one call
    want = oracle.Chain(**NRSC5).process(raw)
    d = np.abs(got.astype(np.int64) - want.astype(np.int64))
    assert d.max() <= 1 and (d == 0).mean() >= 0.998


def oracle_shards(oracle, raw, bounds, kw):
    """what N runs of the reference + `cat` produce: every shard a FRESH oracle chain over its own file range"""
    return [oracle.Chain(**kw).process(raw[2 * a:2 * b]) for a, b in bounds]


def int_close(got, want, min_same=0.998):
    assert got.size == want.size, (got.size, want.size)
    d = np.abs(got.astype(np.int64) - want.astype(np.int64))
    assert d.max() <= 1, int(d.max())
    assert (d == 0).mean() >= min_same, float((d == 0).mean())


def test_harness_shards_are_independent_streams_stitched_in_order(gpu, oracle, tmp_path):
    """BASELINE configs[4] in miniature: N file ranges, fresh state each, concatenated -- against the concatenation of N ORACLE
    shard outputs (the parity oracle of configs[4]), and byte-equal to N separate HIP chains"""
    n = 4 * 500_000
    raw = synth.raw_stream(n, 2.4e6, 2, "cs16")
    fin, fout = tmp_path / "in.cs16", tmp_path / "out.cs16"
    raw.tofile(fin)
    info = run("-i", str(fin), "-o", str(fout), *ARGS, "--shards", "4", "--devices", "1", "--chunk-frames", "131072")
    got = np.fromfile(fout, np.int16)
    bounds = [(s * 500_000, (s + 1) * 500_000) for s in range(4)]
    int_close(got, np.concatenate(oracle_shards(oracle, raw, bounds, NRSC5)))
    parts = [gpu.Chain(**NRSC5).process(raw[2 * a:2 * b]) for a, b in bounds]
    assert np.array_equal(got, np.concatenate(parts))
    assert info["shards"] == 4


def test_harness_ragged_shards_cross_many_chunk_boundaries(gpu, oracle, tmp_path):
    """shards that are no multiple of the harness chunk, of a tile or of the decimation group (the last one takes the
    remainder), each many chunks long: the stitched file is the concatenation of the oracle's shard outputs"""
    shards, n = 3, 3 * 777_777 + 5
    raw = synth.raw_stream(n, 2.4e6, 21, "cs16")
    fin, fout = tmp_path / "in.cs16", tmp_path / "out.cs16"
    raw.tofile(fin)
    info = run("-i", str(fin), "-o", str(fout), *ARGS, "--shards", str(shards), "--devices", "1", "--chunk-frames", "49152")
    per = n // shards
    bounds = [(s * per, n if s == shards - 1 else (s + 1) * per) for s in range(shards)]
    int_close(np.fromfile(fout, np.int16), np.concatenate(oracle_shards(oracle, raw, bounds, NRSC5)))
    assert info["shards"] == shards and info["frames_in"] == n


def test_harness_filter_options(gpu, tmp_path):
    n = 600_000
    raw = synth.raw_stream(n, 10e6, 3, "cs16")
    fin, fout = tmp_path / "in.cs16", tmp_path / "out.cs16"
    raw.tofile(fin)
    run("-i", "raw-file", str(fin), "-o", "raw-file", str(fout), "--raw-file-input-rate", "10e6", "--raw-file-input-sample-format", "cs16",
        "--output-rate", "2.4e6", "--output-sample-format", "cs16", "--dc-block", "--iq-factors", "0.01:-0.005",
        "--pass-range", "102e3:215e3", "--filter-taps", "1024", "--chunk-frames", "100000")
    got = np.fromfile(fout, np.int16)
    want = gpu.Chain(in_format="cs16", out_format="cs16", input_rate_hz=10e6, target_rate_hz=2.4e6, dc_block=True, iq_correct=True,
                     iq_mag=0.01, iq_phase=-0.005, filters=(("passband", 158.5e3, 113e3),), filter_taps=1024).process(raw)
    assert got.size == want.size and got.size % 4096 == 0
    d = np.abs(got.astype(np.int64) - want.astype(np.int64))
    assert d.max() <= 1 and (d == 0).mean() > 0.99          # dc-blocker carries differ by rounding only


def _harness_args(in_fmt, in_rate, out_fmt, out_rate, extra=()):
    return ["--raw-file-input-rate", repr(in_rate), "--raw-file-input-sample-format", in_fmt, "--output-rate", repr(out_rate),
            "--output-sample-format", out_fmt, *extra]


@pytest.mark.parametrize("name,kw,extra", [
    # r = 4: arbitrary stage first, then two half-band interpolators (k_interp)
    ("interp4", dict(in_format="cs16", out_format="cs16", input_rate_hz=600e3, target_rate_hz=2.4e6), ()),
    # r = 2.5 with an FFT-kind filter IN FRONT of the resampler: block quantisation happens before it
    ("interp_fft", dict(in_format="cs16", out_format="cf32", input_rate_hz=1.0e6, target_rate_hz=2.5e6,
                        filters=(("lowpass", 200e3, 0.0),), filter_taps=257, filter_impl="fft"),
     ("--lowpass", "200e3", "--filter-taps", "257", "--filter-type", "fft")),
    # r = 1 exactly with a pre filter
    ("unit_fft", dict(in_format="cu8", out_format="cu8", input_rate_hz=2.0e6, target_rate_hz=2.0e6,
                      filters=(("lowpass", 300e3, 0.0),), filter_taps=129, filter_impl="fft"),
     ("--lowpass", "300e3", "--filter-taps", "129", "--filter-type", "fft")),
])
def test_harness_shards_on_non_decimating_chains(gpu, tmp_path, name, kw, extra):
    """shard output offsets come from iqgpu_design_out_frames: r >= 1 and pre-resample FFT filters included"""
    shards = 3
    n = 150_001 * shards + 17
    per = n // shards                                        # the harness's plan: equal ranges, the last shard takes the rest
    raw = synth.raw_stream(n, kw["input_rate_hz"], 6, kw["in_format"])
    fin, fout = tmp_path / "in.raw", tmp_path / "out.raw"
    raw.tofile(fin)
    info = run("-i", str(fin), "-o", str(fout), *_harness_args(kw["in_format"], kw["input_rate_hz"], kw["out_format"], kw["target_rate_hz"], extra),
               "--shards", str(shards), "--devices", "1", "--chunk-frames", "65536")
    ch = gpu.Chain(**kw)
    bpf = ch.in_bytes
    rb = raw.view(np.uint8)
    parts = []
    for s in range(shards):
        a = s * per
        b = n if s == shards - 1 else (s + 1) * per
        parts.append(gpu.Chain(**kw).process(rb[a * bpf:b * bpf]))
    want = np.concatenate(parts)
    got = np.fromfile(fout, want.dtype)
    assert info["shards"] == shards and got.size == want.size
    if want.dtype == np.float32:
        assert np.abs(got - want).max() <= 2e-6            # chunked calls vs one call per shard
    else:
        assert np.abs(got.astype(np.int64) - want.astype(np.int64)).max() <= 1


def test_harness_shards_over_all_visible_devices(gpu, oracle, tmp_path):
    """--shards K --devices min(K, device_count): one chain per GPU, outputs stitched in shard order (no collective)"""
    ndev = gpu.load().iqgpu_device_count()
    shards = 4
    n = shards * 400_000
    raw = synth.raw_stream(n, 2.4e6, 12, "cs16")
    fin, fout = tmp_path / "in.cs16", tmp_path / "out.cs16"
    raw.tofile(fin)
    info = run("-i", str(fin), "-o", str(fout), *ARGS, "--shards", str(shards), "--devices", str(min(shards, ndev)), "--chunk-frames", "131072")
    got = np.fromfile(fout, np.int16)
    bounds = [(s * 400_000, (s + 1) * 400_000) for s in range(shards)]
    int_close(got, np.concatenate(oracle_shards(oracle, raw, bounds, NRSC5)))
    parts = [gpu.Chain(**NRSC5).process(raw[2 * a:2 * b]) for a, b in bounds]
    assert np.array_equal(got, np.concatenate(parts))
    assert info["shards"] == shards


# --------------------------------------------------------------------------------------------
# SURVEY 8f-4 on the device: WAV capture -> iqgpu_wav_probe -> iqgpu_wav_shift_hz -> desc.shift_hz -> the HIP chain,
# against the oracle chain fed the shift that oracle/wav_oracle.py derives from the same bytes
# (src/input_wav.c:592-629 wav_initialize, src/frequency_shift.c:27-31 which shift the NCO gets)
# --------------------------------------------------------------------------------------------
def _riff(chunks):
    import struct
    body = b"WAVE" + b"".join(cid + struct.pack("<I", len(b)) + b + (b"\0" if len(b) & 1 else b"") for cid, b in chunks)
    return b"RIFF" + struct.pack("<I", len(body)) + body


@pytest.mark.parametrize("case", ["console_xml", "sdruno_binary_cu8", "sdrsharp_filename"])
@pytest.mark.parametrize("target_hz", [97.7e6, 97.70001e6])          # the second is not a float: the option is one (float)target
def test_wav_capture_shift_reaches_the_hip_chain(gpu, oracle, tmp_path, case, target_hz):
    import struct
    from iq_tool_amd import wav_meta
    from oracle import wav_oracle
    fixtures = os.path.join(ROOT, "tests", "golden", "wav")
    n = 300_000
    if case == "sdruno_binary_cu8":
        rate, fmt_name, bits, name = 2_000_000, "cu8", 8, "SDRuno_20240131_123456Z_97900kHz.wav"
        st = struct.pack("<8H", 2024, 1, 3, 31, 12, 34, 56, 0)
        auxi = st + st + struct.pack("<I", 97_900_000) + bytes(128)
    elif case == "console_xml":
        rate, fmt_name, bits, name = 2_400_000, "cs16", 16, "capture.wav"
        src = open(os.path.join(fixtures, "console_capture.wav"), "rb").read()
        p = src.index(b"auxi")
        auxi = src[p + 8:p + 8 + struct.unpack_from("<I", src, p + 4)[0]]
    else:
        rate, fmt_name, bits, name, auxi = 2_400_000, "cs16", 16, "SDRSharp_20240131_123456Z_97900000Hz_IQ.wav", None
    raw = synth.raw_stream(n, float(rate), 4, fmt_name)
    ba = 2 * bits // 8
    chunks = [(b"fmt ", struct.pack("<HHIIHH", 1, 2, rate, rate * ba, ba, bits))]
    if auxi is not None:
        chunks.append((b"auxi", auxi))
    chunks.append((b"data", raw.tobytes()))
    path = tmp_path / name
    path.write_bytes(_riff(chunks))

    # product side: the probe gives format, rate, frames, data range and the shift; nothing below is told what the file holds
    md = wav_meta.probe(str(path))
    assert md.frames == n and md.sample_rate == rate
    shift = wav_meta.shift_hz(md, target_hz, 0.0)
    in_fmt = {11: "cs16", 8: "cu8"}[md.in_format]
    assert in_fmt == fmt_name
    with open(path, "rb") as fh:
        fh.seek(md.data_offset)
        data = np.frombuffer(fh.read(md.data_bytes), np.int16 if in_fmt == "cs16" else np.uint8)
    kw = dict(in_format=in_fmt, out_format="cs16", input_rate_hz=float(md.sample_rate), target_rate_hz=744187.5)
    got = gpu.Chain(shift_hz=shift, **kw).process(data)

    # oracle side: expat-parsed (or binary / file-name) metadata, the reference's shift rule, the oracle chain
    ref = wav_oracle.new_md()
    if auxi is not None:
        wav_oracle.parse_auxi(auxi, ref)
    wav_oracle.parse_filename(name, ref)
    err, want_shift = wav_oracle.shift_hz(ref, target_hz, 0.0)
    assert err is None and want_shift == 97_900_000.0 - float(np.float32(target_hz))
    assert shift == want_shift
    want = oracle.Chain(shift_hz=want_shift, **kw).process(raw)
    assert got.size == want.size and got.size > 0
    d = np.abs(got.astype(np.int64) - want.astype(np.int64))
    assert d.max() <= 1 and (d == 0).mean() >= 0.99, (d.max(), (d == 0).mean())
    # and the shift matters: the same capture without it is a different signal
    plain = gpu.Chain(shift_hz=0.0, **kw).process(data)
    assert (plain != got).mean() > 0.5


# --------------------------------------------------------------------------------------------
# row e on one GPU: the REAL step under two ranks (the launcher, the gloo barrier / MAX and two processes on the card)
# --------------------------------------------------------------------------------------------
def test_bench_two_ranks_share_the_gpu():
    """bench.py --gpus 2 with both ranks on this box's one GPU (IQGPU_BENCH_SHARE_GPU): not a scaling figure -- a check that
    the first multi-GPU run the driver makes is not lost to plumbing (spawned fresh children, one JSON line, n_gpus 2)"""
    import sys
    env = dict(os.environ, IQGPU_BENCH_SHARE_GPU="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k, None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--log2-frames", "22",
                        "--settle-seconds", "0.2", "--no-cpu-baseline", "--no-host-leg", "--no-secondary", "--no-extra"],
                       env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [l for l in p.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, p.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 3 and d["scaling"] == "weak"
    # two ranks x 3 steps x 2^22 frames over the MAX-over-ranks time: above any CPU rate, below what one card can do
    assert 1e3 < d["value"] < 2e6, d["value"]
    assert abs(d["value"] - 2 * 3 * (1 << 22) / (d["ms_per_step"] * 3 * 1e-3) / 1e6) <= 0.01 * d["value"]
    assert d["roofline"]["launches"] == 3 and d["roofline"]["kernel_ms"] > 0


def test_harness_two_devices(gpu, oracle, tmp_path):
    """iqgpu_run --shards 4 --devices 2: shards dealt round-robin over two cards, stitched in file order -- against the
    concatenation of four ORACLE shard outputs, like every other shard test (the first multi-GPU box must not compare HIP with
    HIP); byte-equal to four chains on device 0 on top.  Skips on a one-GPU box."""
    if gpu.load().iqgpu_device_count() < 2:
        pytest.skip("needs two HIP devices (BASELINE configs[4] runs one shard per GPU); this box has one")
    n = 4 * 400_000
    raw = synth.raw_stream(n, 2.4e6, 6, "cs16")
    fin, fout = tmp_path / "in.cs16", tmp_path / "out.cs16"
    raw.tofile(fin)
    info = run("-i", str(fin), "-o", str(fout), *ARGS, "--shards", "4", "--devices", "2", "--chunk-frames", "131072")
    got = np.fromfile(fout, np.int16)
    bounds = [(s * 400_000, (s + 1) * 400_000) for s in range(4)]
    int_close(got, np.concatenate(oracle_shards(oracle, raw, bounds, NRSC5)))
    parts = [gpu.Chain(**NRSC5).process(raw[2 * a:2 * b]) for a, b in bounds]
    assert np.array_equal(got, np.concatenate(parts))
    assert info["shards"] == 4
    assert [ps["device"] for ps in info["per_shard"]] == [0, 1, 0, 1]


def test_harness_real_size_shards_from_the_hash_stream(gpu, oracle):
    """BASELINE configs[4]'s shard SIZE (more than 2^31 frames, more than 2^33 input bytes per shard) through the harness: two
    shards of 2^31 + 12345 frames of the counter-hash stream (--synthetic-hash: no 17 GB of files; synth.hash_stream regenerates
    any range), output stitched in /dev/shm.  Counts and offsets against iqgpu_design_out_frames, the head of BOTH shards
    against a fresh oracle chain each, the tail of shard 0 against a chain that ran the whole shard in calls of another size
    (sha256): 32-bit frame counts anywhere in the harness or the plan would show here.  tools/gpu/r5_config5.py is the same
    at 8 x 2.5 G frames (profiles/r05_config5_1gpu.json)."""
    import hashlib
    from iq_tool_amd.chain import design_out_frames
    per, shards, seed = (1 << 31) + 12345, 2, 40
    fout = "/dev/shm/iqgpu_test_real_size_%d.cs16" % os.getpid()
    try:
        info = run("--synthetic", str(per * shards), "--synthetic-hash", str(seed), "-o", fout, *ARGS, "--shards", str(shards), "--devices", "1")
        want_n = design_out_frames(frames_in=per, **NRSC5)
        assert info["frames_in"] == per * shards and info["input"] == "synthetic-hash"
        for s, ps in enumerate(info["per_shard"]):
            assert ps["frames_in"] == per and ps["frames_out"] == want_n == ps["planned_out"] and ps["out_offset_bytes"] == 4 * want_n * s, ps
        assert os.path.getsize(fout) == 4 * want_n * shards
        out = np.memmap(fout, dtype=np.int16, mode="r")
        head = 1 << 21
        for s in range(shards):
            want = oracle.Chain(**NRSC5).process(synth.hash_stream(head, seed + s, "cs16", 0))
            int_close(np.asarray(out[2 * want_n * s:2 * want_n * s + want.size]), want)
        # (the whole of shard 0 again as ONE stream in calls of another size, sha256 against its range of the file, costs half a
        #  minute of numpy hashing: IQGPU_TEST_FULL_STREAM=1 runs it here; tools/gpu/r5_config5.py always does, at 2.5 G frames)
        if os.environ.get("IQGPU_TEST_FULL_STREAM") == "1":
            ch, h, pos, n_out = gpu.Chain(**NRSC5), hashlib.sha256(), 0, 0
            while pos < per:
                n = min(1 << 27, per - pos)
                y = ch.process(synth.hash_stream(n, seed, "cs16", pos))
                h.update(y.tobytes()); n_out += y.size // 2; pos += n
            assert n_out == want_n
            assert hashlib.sha256(np.asarray(out[:2 * want_n]).tobytes()).hexdigest() == h.hexdigest()
        # the very last output frames of both shards exist and are not the zeros of an unwritten hole
        for s in range(shards):
            assert np.any(np.asarray(out[2 * want_n * (s + 1) - 4096:2 * want_n * (s + 1)]) != 0)
        del out
    finally:
        if os.path.exists(fout):
            os.remove(fout)
