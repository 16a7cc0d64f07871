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
    assert d.max() <= 1 and (d == 0).mean() > 0.97


def test_harness_shards_are_independent_streams_stitched_in_order(gpu, tmp_path):
    """BASELINE configs[4] in miniature: N file ranges, fresh state each, concatenated"""
    n = 4 * 500_000
    raw = synth.raw_stream(n, 2.4e6, 2, "cs16")
    fin, fout = tmp_path / "in.cs16", tmp_path / "out.cs16"
    raw.tofile(fin)
    info = run("-i", str(fin), "-o", str(fout), *ARGS, "--shards", "4", "--devices", "1", "--chunk-frames", "131072")
    got = np.fromfile(fout, np.int16)
    parts = [gpu.Chain(**NRSC5).process(raw[2 * s * 500_000:2 * (s + 1) * 500_000]) for s in range(4)]
    assert np.array_equal(got, np.concatenate(parts))
    assert info["shards"] == 4


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


def test_harness_shards_over_all_visible_devices(gpu, tmp_path):
    """--shards K --devices min(K, device_count): one chain per GPU, outputs stitched in shard order (no collective)"""
    ndev = gpu.load().iqgpu_device_count()
    shards = 4
    n = shards * 400_000
    raw = synth.raw_stream(n, 2.4e6, 12, "cs16")
    fin, fout = tmp_path / "in.cs16", tmp_path / "out.cs16"
    raw.tofile(fin)
    info = run("-i", str(fin), "-o", str(fout), *ARGS, "--shards", str(shards), "--devices", str(min(shards, ndev)), "--chunk-frames", "131072")
    got = np.fromfile(fout, np.int16)
    parts = [gpu.Chain(**NRSC5).process(raw[2 * s * 400_000:2 * (s + 1) * 400_000]) for s in range(shards)]
    assert np.array_equal(got, np.concatenate(parts))
    assert info["shards"] == shards
