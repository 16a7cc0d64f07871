#!/usr/bin/env python3
"""BASELINE configs[4] at its REAL shape on however many GPUs the box has (one here): 8 shards x 2.5 G cs16 frames (10 GB each,
> 2^31 frames and > 2^33 bytes per shard), the NRSC-5 chain, fresh state per shard, outputs stitched in shard order by the
harness (iq_tool_amd/csrc/harness/iqgpu_run.c) -- what 8 runs of the reference + `cat` produce
(/root/reference/src/input_rawfile.c:168-252, src/output_raw_file.c:146-184).

The input is the harness's counter-hash stream (--synthetic-hash: shard s = seed 10 + s, SURVEY 8d's seeds), so that no 80 GB of
files are needed and every range of every shard can be regenerated here.  Checks, all against things computed OUTSIDE the run:

  1. every shard's frames_out and byte offset == iqgpu_design_out_frames (closed form, no device), file size == their sum;
  2. sha256 of shard 0's byte range in the stitched file == sha256 of ONE iq_tool_amd.Chain fed the same 2.5 G frames through
     iqgpu_chain_process in calls of another size (2^26 frames): placement, chunk invariance and > 2^31 frames in one stream;
  3. oracle parity (+-1 code, >= 99.8 % identical) on the first 2^25 input frames' worth of output of shards 0 and 7, and
     shard 7's head starts at its planned offset (a fresh chain: the oracle restarts too).

Writes gpurun_out/r5_config5/config5_1gpu.json (copied to profiles/r05_config5_1gpu.json).  Progress lines on stdout.
"""
import hashlib
import json
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import iq_tool_amd                                   # noqa: E402
from iq_tool_amd import synth                        # noqa: E402
from oracle import pyoracle                          # noqa: E402  (the checker)

EXE = os.path.join(ROOT, "iq_tool_amd", "lib", "iqgpu_run")
NRSC5 = dict(in_format="cs16", out_format="cs16", input_rate_hz=2.4e6, target_rate_hz=744187.5, shift_hz=200e3)
ARGS = ["--raw-file-input-rate", "2.4e6", "--raw-file-input-sample-format", "cs16", "--output-rate", "744187.5",
        "--output-sample-format", "cs16", "--freq-shift", "200e3"]


def say(*a):
    print(time.strftime("%H:%M:%S"), *a, flush=True)


def main():
    shards = int(os.environ.get("R5_SHARDS", "8"))
    per = int(os.environ.get("R5_FRAMES_PER_SHARD", str(2_500_000_000)))
    seed0 = 10
    devices = iq_tool_amd.load().iqgpu_device_count()
    outdir = os.path.join(ROOT, "gpurun_out", "r5_config5")
    os.makedirs(outdir, exist_ok=True)
    fout = os.environ.get("R5_OUT", "/dev/shm/r5_config5_out.cs16")
    total = shards * per
    res = {"workload": "BASELINE configs[4]: %d independent raw cs16 shards of %d frames (%.1f GB) each, NRSC-5 chain, stitched at the writer"
                       % (shards, per, per * 4 / 1e9), "devices_on_box": devices, "checks": {}}

    # ---- the run: all devices of the box, then (PCIe only) the same with the constant buffer: what the generator costs
    runs = {}
    for name, extra, out in (("hash_input_written", ["--synthetic-hash", str(seed0)], fout),
                             ("constant_input_no_file", [], None)):
        cmd = [EXE, "--synthetic", str(total), *extra, *ARGS, "--shards", str(shards), "--devices", str(max(devices, 1))]
        if out:
            cmd += ["-o", out]
        say("run", name, " ".join(cmd[1:]))
        t0 = time.time()
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise SystemExit("harness failed: " + r.stderr + r.stdout)
        runs[name] = json.loads(r.stdout.strip().splitlines()[-1])
        runs[name]["wall_s"] = round(time.time() - t0, 3)
        say("  ->", {k: runs[name][k] for k in ("seconds", "msps_end_to_end", "h2d_GBs", "d2h_GBs")})
    res["runs"] = runs
    info = runs["hash_input_written"]

    # ---- 1. counts and offsets against the closed form
    from iq_tool_amd.chain import design_out_frames
    off = 0
    for s, ps in enumerate(info["per_shard"]):
        n_in = per if s < shards - 1 else total - per * (shards - 1)
        want = design_out_frames(frames_in=n_in, **NRSC5)
        assert ps["frames_in"] == n_in and ps["frames_out"] == want and ps["planned_out"] == want, (s, ps, want)
        assert ps["out_offset_bytes"] == off, (s, ps["out_offset_bytes"], off)
        off += 4 * want
    assert os.path.getsize(fout) == off, (os.path.getsize(fout), off)
    assert info["frames_out"] * 4 == off
    res["checks"]["counts_and_offsets"] = "ok: %d shards, %d output frames, file %d bytes == sum of iqgpu_design_out_frames" % (shards, info["frames_out"], off)
    say(res["checks"]["counts_and_offsets"])
    out = np.memmap(fout, dtype=np.int16, mode="r")

    # ---- 3 (first: cheap). oracle parity on the heads of shards 0 and last
    head = 1 << 25
    for s in (0, shards - 1):
        raw = synth.hash_stream(head, seed0 + s, "cs16", 0)
        want = pyoracle.Chain(**NRSC5).process(raw)
        o0 = info["per_shard"][s]["out_offset_bytes"] // 2
        got = np.asarray(out[o0:o0 + want.size])
        d = np.abs(got.astype(np.int64) - want.astype(np.int64))
        same = float((d == 0).mean())
        assert d.max() <= 1 and same >= 0.998, (s, int(d.max()), same)
        res["checks"]["oracle_head_shard%d" % s] = "ok: first 2^25 input frames -> %d output frames, max |diff| %d code, %.5f identical" % (want.size // 2, int(d.max()), same)
        say(res["checks"]["oracle_head_shard%d" % s])

    # ---- 2. shard 0 as ONE stream through iqgpu_chain_process in 2^26-frame calls
    t0 = time.time()
    ch = iq_tool_amd.Chain(device=0, **NRSC5)
    h_stream, n_out, pos, step = hashlib.sha256(), 0, 0, 1 << 26
    n0 = info["per_shard"][0]["frames_in"]
    while pos < n0:
        n = min(step, n0 - pos)
        y = ch.process(synth.hash_stream(n, seed0, "cs16", pos))
        h_stream.update(y.tobytes()); n_out += y.size // 2
        pos += n
        if (pos // step) % 8 == 0:
            say("  single stream: %d / %d frames" % (pos, n0))
    ch.close()
    assert n_out == info["per_shard"][0]["frames_out"], (n_out, info["per_shard"][0]["frames_out"])
    h_file = hashlib.sha256()
    end0 = info["per_shard"][0]["frames_out"] * 2
    for a in range(0, end0, 1 << 27):
        h_file.update(np.asarray(out[a:min(end0, a + (1 << 27))]).tobytes())
    assert h_file.hexdigest() == h_stream.hexdigest(), (h_file.hexdigest(), h_stream.hexdigest())
    res["checks"]["shard0_sha256"] = "ok: %s == one iqgpu_chain_process stream of the same %d frames in 2^26-frame calls (%.0f s)" % (h_file.hexdigest(), n0, time.time() - t0)
    say(res["checks"]["shard0_sha256"])

    del out
    os.remove(fout)
    with open(os.path.join(outdir, "config5_1gpu.json"), "w") as fh:
        json.dump(res, fh, indent=1)
    say("written", os.path.join(outdir, "config5_1gpu.json"))


if __name__ == "__main__":
    main()
