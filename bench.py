#!/usr/bin/env python3
"""bench.py -- complex MS/s of the NRSC-5 shift + resample chain on MI355X (BASELINE.json configs[1]).

    python bench.py --gpus N --steps K --warmup W

One process per GPU (launched with torch.distributed.run for N > 1).  A step is one pass of the
hot path (iqgpu_chain_process_device: unpack -> NCO +200 kHz -> half-band -> 256-arm polyphase ->
pack) over one batch of 2^28 synthetic cs16 frames that is already resident in HBM; the stream is
continuous from step to step.  Every rank works on its own independent shard (seed 10 + rank), as
BASELINE.json configs[4] prescribes: no data-path collective, weak scaling.

Rank 0 prints ONE JSON line.  `roofline` prices the dominant kernel (k_front) with HIP events
recorded on the chain's stream inside the timed region; `cpu_baseline` times the oracle's
float-accumulator build (a port, 1 core) on a bounded sample of the same workload (N = 1 only).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md)
CHAIN = dict(in_format="cs16", out_format="cs16", input_rate_hz=2.4e6, target_rate_hz=744187.5, shift_hz=200e3)
# the other single-GPU BASELINE.json configs (parity-test cases; timed only with --config 3 / 4, never the default line)
OTHER = {
    3: dict(chain=dict(in_format="cs16", out_format="cs16", input_rate_hz=10e6, target_rate_hz=2.4e6, dc_block=True,
                       iq_correct=True, iq_mag=0.01, iq_phase=-0.005, filters=(("passband", 158.5e3, 113e3),), filter_taps=1024),
            log2_frames=27, rate=10e6, fmt="cs16", bps=4,
            workload="BASELINE configs[2]: cs16 10 MS/s -> 2.4 MS/s, dc block + iq correct, 2 half-bands, 1025-tap complex band-pass (FFT kind, block 2048), cs16 out"),
    4: dict(chain=dict(in_format="cu8", out_format="cu8", input_rate_hz=61.44e6, target_rate_hz=1488375.0,
                       filters=(("lowpass", 300e3, 0.0),), filter_taps=4097, filter_impl="fir"),
            log2_frames=29, rate=61.44e6, fmt="cu8", bps=2,
            workload="BASELINE configs[3]: cu8 61.44 MS/s -> 1.488375 MS/s, 5 half-bands, 4097-tap real FIR (time domain), cu8 out"),
}
BLOCK_SAMPLES = 0               # auto: one contiguous run of tiles per resident wavefront
SEGMENT_LOG2 = 22              # synthetic segment generated on the host, tiled on the device


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--log2-frames", type=int, default=28, help="frames per step and GPU (default 2^28 = 1 GiB of cs16)")
    ap.add_argument("--cpu-frames-log2", type=int, default=28, help="bounded CPU-baseline sample (2^28 = one step's batch, ~15 s)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--config", type=int, default=2, choices=[2, 3, 4], help="2 = BASELINE configs[1] (the metric's config, default); 3 / 4 = configs[2] / configs[3], secondary timings")
    ap.add_argument("--traffic-bytes", type=float, default=1431220224.0,
                    help="HBM bytes per k_front launch from the separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes (profiles/r01_pmc_summary.txt: 2 x FETCH_SIZE + WRITE_SIZE, KiB -> bytes); reported as roofline.traffic when the workload is the default 2^28 frames")
    return ap.parse_args()


def shard_plan(world_size, rank, frames_per_gpu):
    """Independent file-offset shards: rank r owns frames [r*F, (r+1)*F) of the global job."""
    return dict(first_frame=rank * frames_per_gpu, frames=frames_per_gpu, seed=10 + rank if world_size > 1 else 1)


def timed_region(dist, sync, step_fn, steps):
    """barrier + sync | steps | sync + barrier; returns the MAX over ranks of the elapsed seconds."""
    import torch
    if dist is not None:
        dist.barrier()
    sync()
    t0 = time.perf_counter()
    for _ in range(steps):
        step_fn()
    sync()
    if dist is not None:
        dist.barrier()
    dt = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([dt], dtype=torch.float64)
        if torch.cuda.is_available() and dist.get_backend() == "nccl":
            t = t.cuda()
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    return dt


def cpu_baseline(frames_log2):
    """The oracle (float accumulators, -O3 -march=native) on a bounded sample of the same workload:
    one thread, and the reference's own arrangement -- three concurrent stage threads handing over
    16384-frame chunks (src/pipeline.c:96-116).  The faster of the two is reported."""
    from iq_tool_amd import synth
    from oracle import pyoracle
    pyoracle.build()
    n = 1 << frames_log2
    seg = synth.raw_stream(1 << 20, 2.4e6, 1, "cs16")
    raw = np.tile(seg, n >> 20)
    L = pyoracle.lib(fast=True)
    ch = pyoracle.Chain(L=L, **CHAIN)
    ch.process(raw[: 2 << 20])                       # warm caches / page in
    ch.reset()
    t0 = time.perf_counter()
    out = ch.process(raw)
    dt1 = time.perf_counter() - t0
    assert out.size > 0
    ch3 = pyoracle.Chain(L=L, **CHAIN)
    t0 = time.perf_counter()
    out3 = ch3.process_pipelined(raw)
    dt3 = time.perf_counter() - t0
    assert out3.size == out.size
    v1, v3 = n / dt1 / 1e6, n / dt3 / 1e6
    return dict(value=round(max(v1, v3), 3), unit="MS/s", cores=3 if v3 > v1 else 1, kind="port",
                sample="2^%d cs16 frames of the same NRSC-5 chain, oracle/liboracle_fast.so (float accumulators): "
                       "%.1f MS/s on 1 thread, %.1f MS/s as 3 stage threads (pre / resampler / post, 16384-frame chunks)"
                       % (frames_log2, v1, v3))


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        raise SystemExit("WORLD_SIZE (%d) != --gpus (%d)" % (world, args.gpus))

    import torch
    import iq_tool_amd
    from iq_tool_amd import synth

    lib = iq_tool_amd.load()                          # raises when libiqgpu.so is missing
    if lib.iqgpu_device_count() < 1 or not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: no HIP device visible (there is no CPU fallback)")

    dist = None
    if world > 1:
        import torch.distributed as dist_mod
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local_rank)
        dist_mod.init_process_group(backend="nccl", rank=rank, world_size=world)
        dist = dist_mod
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    chain_kw, rate, fmt, in_bps, workload = CHAIN, 2.4e6, "cs16", 4, None
    if args.config != 2:
        o = OTHER[args.config]
        chain_kw, rate, fmt, in_bps, workload = o["chain"], o["rate"], o["fmt"], o["bps"], o["workload"]
        if args.log2_frames == 28:
            args.log2_frames = o["log2_frames"]
    frames = 1 << args.log2_frames
    plan = shard_plan(world, rank, frames)
    seg_frames = min(frames, 1 << SEGMENT_LOG2)
    seg = synth.raw_stream(seg_frames, rate, plan["seed"], fmt)               # interleaved I,Q integers
    d_seg = torch.from_numpy(seg).to(dev)
    d_in = d_seg.repeat(frames // seg_frames).contiguous()                     # resident in HBM
    del d_seg

    chain = iq_tool_amd.Chain(device=local_rank, block_samples=BLOCK_SAMPLES, **chain_kw)
    chain.set_stream(torch.cuda.current_stream(dev).cuda_stream)
    cap_frames = chain.max_out_frames(frames)
    d_out = torch.empty(cap_frames * chain.out_bytes, dtype=torch.uint8, device=dev)
    out_frames = []

    def step():
        out_frames.append(chain.process_device(d_in.data_ptr(), frames, d_out.data_ptr(), d_out.numel()))

    def sync():
        torch.cuda.synchronize(dev)

    for _ in range(args.warmup):
        step()
    sync()
    out_frames.clear()
    chain.set_profiling(True)
    chain.profile()                                   # clear
    dt = timed_region(dist, sync, step, args.steps)
    prof = chain.profile()
    chain.set_profiling(False)

    total_frames = world * args.steps * frames
    value = total_frames / dt / 1e6
    front = prof["front"]
    k_ms = front["ms"] / max(front["launches"], 1)
    if args.config != 2:
        # secondary configs run several kernels per step (cascade, last stage, dc carries, filter):
        # price the whole step's device time, not one of them
        k_ms = sum(v["ms"] for v in prof.values()) / max(args.steps, 1)
    n_out_avg = float(np.mean(out_frames)) if out_frames else 0.0
    alg_bytes = frames * in_bps + n_out_avg * chain.out_bytes   # in_bytes + r * out_bytes per input frame
    achieved = alg_bytes / (k_ms * 1e-3) / 1e9 if k_ms > 0 else 0.0

    if rank == 0:
        line = {
            "metric": "complex MS/s end-to-end on NRSC-5 resample+filter chain; % HBM roofline",
            "value": round(value, 2), "unit": "MS/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic (seeded 2^%d-frame cs16 segment, 3 tones + noise + DC, tiled in HBM to 2^%d frames per GPU)" % (int(np.log2(seg_frames)), args.log2_frames),
            "config": {"workload": "BASELINE configs[1]: raw cs16 2.4 MS/s -> 744.1875 kS/s, +200 kHz NCO, 1 half-band (m=10) + 256-arm polyphase (14 taps), cs16 out",
                       "frames_per_step_per_gpu": frames, "block_samples": BLOCK_SAMPLES,
                       "sharding": "independent stream per GPU, no collective"},
            "roofline": {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": args.traffic_bytes if args.log2_frames == 28 else None,
                         "kernel": "k_front_s1<4, true>" if args.config == 2 else "k_front", "kernel_ms": round(k_ms, 4), "launches": front["launches"],
                         "algorithmic_bytes_per_launch": int(alg_bytes),
                         "read_only_frac": round(frames * in_bps / (k_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if k_ms > 0 else 0.0},
        }
        if args.config != 2:
            line["config"]["workload"] = workload
            line["roofline"]["traffic"] = None
            line["roofline"]["kernel"] = "all kernels of the step"
            line["roofline"]["note"] = "per-kernel ms: " + ", ".join("%s %.3f" % (k, v["ms"] / max(v["launches"], 1)) for k, v in prof.items() if v["launches"])
        if world == 1 and not args.no_cpu_baseline and args.config == 2:
            line["cpu_baseline"] = cpu_baseline(args.cpu_frames_log2)
        else:
            line["cpu_baseline"] = None
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
