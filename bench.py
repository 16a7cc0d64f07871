#!/usr/bin/env python3
"""bench.py -- complex MS/s of the NRSC-5 shift + resample chain on MI355X (BASELINE.json configs[1]).

    python bench.py --gpus N --steps K --warmup W

One process per GPU: under torch.distributed.run the ranks come from the environment; run plainly with
--gpus N > 1 the script spawns its N workers itself (before anything touches the GPU) and rank 0's
line is printed.  A step is one pass of the
hot path (iqgpu_chain_process_device: unpack -> NCO +200 kHz -> half-band -> 256-arm polyphase ->
pack) over one batch of 2^28 synthetic cs16 frames that is already resident in HBM; the stream is
continuous from step to step.  Every rank works on its own independent shard (seed 10 + rank), as
BASELINE.json configs[4] prescribes: no data-path collective, weak scaling.

Rank 0 prints ONE JSON line.  `roofline` prices the dominant kernel (k_front) with HIP events
recorded on the chain's stream inside the timed region; `host_end_to_end` is a second, separate timed
leg through the pipelined host entry point (pinned host buffers -> H2D -> kernels -> D2H, PCIe-bound,
never `value`); `cpu_baseline` times the oracle's float-accumulator build (a port) on a bounded
sample of the same workload (N = 1 only).  Barrier and MAX-over-ranks time go over gloo: the data path
has no collective and no RCCL.
"""
import argparse
import hashlib
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md)
CHAIN = dict(in_format="cs16", out_format="cs16", input_rate_hz=2.4e6, target_rate_hz=744187.5, shift_hz=200e3)
DIAG_TARGET_HZ = os.environ.get("IQGPU_BENCH_TARGET_HZ")      # diagnostic runs of tools/ only: another output rate for the headline chain
if DIAG_TARGET_HZ:
    CHAIN = dict(CHAIN, target_rate_hz=float(DIAG_TARGET_HZ))
# the other single-GPU BASELINE.json configs (parity-test cases; timed only with --config 3 / 4, never the default line)
OTHER = {
    3: dict(chain=dict(in_format="cs16", out_format="cs16", input_rate_hz=10e6, target_rate_hz=2.4e6, dc_block=True,
                       iq_correct=True, iq_mag=0.01, iq_phase=-0.005, filters=(("passband", 158.5e3, 113e3),), filter_taps=1024),
            log2_frames=27, rate=10e6, fmt="cs16", bps=4,
            workload="BASELINE configs[2]: cs16 10 MS/s -> 2.4 MS/s, dc block + iq correct, 2 half-bands, 1025-tap complex band-pass (FFT kind: fftfilt block 2048 output counts; executed as 4096-point overlap-save in LDS), cs16 out"),
    4: dict(chain=dict(in_format="cu8", out_format="cu8", input_rate_hz=61.44e6, target_rate_hz=1488375.0,
                       filters=(("lowpass", 300e3, 0.0),), filter_taps=4097, filter_impl="fir"),
            log2_frames=29, rate=61.44e6, fmt="cu8", bps=2,
            workload="BASELINE configs[3]: cu8 61.44 MS/s -> 1.488375 MS/s, 5 half-bands, 4097-tap real FIR-kind low-pass behind the resampler (the same linear convolution as the time-domain form, executed as 8192-point overlap-save in LDS; the direct-form k_fir takes 6.9 ms for it), cu8 out"),
}
# the seven presets the reference ships enabled (/root/reference/iq_tool_presets.conf:190-248), each as the chain it configures for a
# 2.4 MS/s capture in the preset's own sample format (an RTL-SDR's cu8 for the cu8 presets, cs16 otherwise): target rate, output
# format, digital output AGC, no dc block / iq correction, and for the -usb / -lsb ones a complex band-pass 102 .. 215 kHz off
# centre (pass_range a:b = centre (a + b) / 2, bandwidth b - a, src/config.c:203-214; transition width and taps by the
# reference's defaults) which the reference places BEHIND the resampler (src/filter.c:53-90)
def _preset(fmt, target, pass_range=None):
    kw = dict(in_format=fmt, out_format=fmt, input_rate_hz=2.4e6, target_rate_hz=target, agc=True, agc_profile="digital")
    if pass_range:
        a, b = pass_range
        kw["filters"] = (("passband", (a + b) / 2.0, b - a),)
    return kw


PRESETS = {
    "cu8-nrsc5": _preset("cu8", 1488375.0),
    "cu8-nrsc5-usb": _preset("cu8", 1488375.0, (102e3, 215e3)),
    "cu8-nrsc5-lsb": _preset("cu8", 1488375.0, (-215e3, -102e3)),
    "cs16-fm-nrsc5": _preset("cs16", 744187.5),
    "cs16-fm-nrsc5-usb": _preset("cs16", 744187.5, (102e3, 215e3)),
    "cs16-fm-nrsc5-lsb": _preset("cs16", 744187.5, (-215e3, -102e3)),
    "cs16-am-nrsc5": _preset("cs16", 46511.71875),
}
# the headline chain (BASELINE configs[1]: 2.4 MS/s -> 744.1875 kS/s, + 200 kHz) on the 8-bit sample formats of the common dongles -- an
# RTL-SDR's cu8, a HackRF's cs8 -- and with 8-bit output: not presets of the reference, but what its `--raw-file-input-sample-format
# cu8|cs8` users run (`secondary.eight_bit`)
NEAR = {
    "cu8-to-cu8": dict(in_format="cu8", out_format="cu8", input_rate_hz=2.4e6, target_rate_hz=744187.5, shift_hz=200e3),
    "cs8-to-cs16": dict(in_format="cs8", out_format="cs16", input_rate_hz=2.4e6, target_rate_hz=744187.5, shift_hz=200e3),
    "cs16-to-cu8": dict(in_format="cs16", out_format="cu8", input_rate_hz=2.4e6, target_rate_hz=744187.5, shift_hz=200e3),
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
    ap.add_argument("--config", default="2", choices=["2", "3", "4", "preset"],
                    help="2 = BASELINE configs[1] (the metric's config, default); 3 / 4 = configs[2] / configs[3], secondary timings; "
                         "preset = configs[1] as the shipped cs16-fm-nrsc5 preset runs it, digital output AGC on (iq_tool_presets.conf:216-222)")
    ap.add_argument("--settle-seconds", type=float, default=2.0,
                    help="untimed back-to-back steps before the W warm-up steps, so that the K timed steps see the clock the chip HOLDS under this load and not the first milliseconds after idle (measured: launches 1-3 0.48 ms, 4-12 up to 0.68 ms, steady state 0.49 ms)")
    ap.add_argument("--no-host-leg", action="store_true", help="skip the host_end_to_end leg")
    ap.add_argument("--no-secondary", action="store_true", help="skip the `secondary` legs (BASELINE configs[2], configs[3], the preset with its AGC)")
    ap.add_argument("--secondary-steps", type=int, default=10)
    ap.add_argument("--preset-settle", type=float, default=0.4, help="seconds of untimed steps in front of each leg of secondary.presets")
    ap.add_argument("--only-presets", action="store_true", help="diagnostic: skip the headline legs' extras and run secondary.presets only")
    ap.add_argument("--presets", default="", help="diagnostic: comma-separated names of the shipped presets to run in secondary.presets (default: all seven)")
    ap.add_argument("--secondary-settle", type=float, default=0.7, help="seconds of untimed steps in front of each secondary leg")
    ap.add_argument("--host-log2-frames", type=int, default=30, help="frames per GPU streamed through pinned host buffers in the host_end_to_end leg")
    ap.add_argument("--host-batch-log2", type=int, default=24, help="frames per submit() in the host_end_to_end leg")
    ap.add_argument("--stub-batch-frames", type=int, default=262144,
                    help="frames per submit() in the host_end_to_end_stub leg: what the INTEGRATION.md binding hands over (16 reference chunks of 16384)")
    ap.add_argument("--stub64-batch-frames", type=int, default=1048576,
                    help="frames per submit() in the host_end_to_end_stub64 leg: the stub's batch when the reader queue is deep (64 reference chunks)")
    ap.add_argument("--stub-log2-frames", type=int, default=28, help="frames per GPU streamed in the host_end_to_end_stub leg")
    ap.add_argument("--no-extra", action="store_true", help="skip the `extra` legs (stub-sized host batches, block_samples = 262144, reference-binary probe)")
    ap.add_argument("--dry-placement", action="store_true",
                    help="everything an N-GPU run does EXCEPT GPU work: spawn the N ranks, bind each to the NUMA node of its GPU (sysfs; "
                         "IQGPU_SYSFS_ROOT names a stand-in tree), rendezvous over gloo, gather and check the devices (N distinct PCI "
                         "addresses), size the pinned and HBM buffers of every leg, run the barrier + MAX plumbing on an empty step, print "
                         "ONE JSON line.  For rehearsing the first multi-GPU run on a machine without GPUs")
    a = ap.parse_args()
    a.preset = a.config == "preset"
    a.config = 2 if a.preset else int(a.config)
    return a


KERNEL_SOURCES = ("front_mid.hip", "front_fat_common.hpp", "front_tiles.hpp", "front_wave.hip", "wave_common.hpp", "dsp_device.hpp", "kernels.hpp")


def kernel_sha():
    """identity of the headline kernel's sources: PMC traffic measured for another build is not reported"""
    h = hashlib.sha256()
    for f in KERNEL_SOURCES:
        with open(os.path.join(ROOT, "iq_tool_amd", "csrc", f), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def measured_traffic(log2_frames):
    """HBM bytes per k_front launch from the separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes
    (tools/traffic_from_pmc.py writes profiles/traffic.json: 2 x FETCH_SIZE + WRITE_SIZE, KiB -> bytes).
    Only reported for the very kernel sources and workload size it was measured on; otherwise null."""
    try:
        with open(os.path.join(ROOT, "profiles", "traffic.json")) as fh:
            t = json.load(fh)
        if t.get("kernel_sha") == kernel_sha() and int(t.get("log2_frames", -1)) == log2_frames:
            return float(t["traffic_bytes"])
    except (OSError, ValueError, KeyError):
        pass
    return None


def cpu_model():
    try:
        with open("/proc/cpuinfo") as fh:
            for line in fh:
                if line.startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def spawn_workers(args):
    """`python bench.py --gpus N` without a launcher: start the N ranks as fresh child processes.  The parent
    never imports torch or touches HIP (a process that has initialised the GPU must not be replaced or
    forked on this pool); it relays rank 0's line and watches ALL children: the first one that exits
    non-zero (no GPU for its local rank, an import error, a kernel fault) ends the others at once --
    they would otherwise sit in the gloo barrier until its timeout -- and its code is returned."""
    import tempfile
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs = []
    out0 = tempfile.TemporaryFile()                   # a file, not a pipe: nobody has to drain it while we poll
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=out0 if r == 0 else subprocess.DEVNULL))
    deadline = time.monotonic() + float(os.environ.get("IQGPU_BENCH_TIMEOUT_S", "1500"))
    rc = 0
    live = list(procs)
    while live:
        for p in list(live):
            code = p.poll()
            if code is None:
                continue
            live.remove(p)
            if code != 0 and rc == 0:
                rc = abs(code) or 1
        if rc == 0 and time.monotonic() > deadline:
            sys.stderr.write("bench.py: ranks still running after the watchdog period, ending them\n")
            rc = 124
        if rc != 0 and live:
            sys.stderr.write("bench.py: a rank failed (exit %d): ending the other %d\n" % (rc, len(live)))
            for p in live:
                p.terminate()                         # exact children of this process, never a pattern
            for p in live:
                try:
                    p.wait(timeout=10)
                except subprocess.TimeoutExpired:
                    p.kill()
                    p.wait()
            live = []
        elif live:
            time.sleep(0.05)
    out0.seek(0)
    sys.stdout.write(out0.read().decode("utf-8", "replace"))
    sys.stdout.flush()
    out0.close()
    return rc


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
        t = torch.tensor([dt], dtype=torch.float64)          # host tensor: the control plane is gloo
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    return dt


def host_leg(dist, chain, seg, frames_total, batch):
    """The same chain fed from pinned HOST memory through iqgpu_chain_submit / _collect: H2D copy, kernels
    and D2H copy of consecutive batches overlap on the library's internal streams (what the reference's
    reader -> DSP threads -> writer hand-off becomes, src/pipeline.c:96-116).  PCIe-bound; returns
    (seconds MAX over ranks, frames, bytes up, bytes down) for this rank."""
    from iq_tool_amd.chain import PinnedBuffer
    depth = chain._lib.iqgpu_chain_pipeline_depth()
    in_b, out_b = chain.in_bytes, chain.out_bytes
    cap = chain.max_out_frames(batch) * out_b
    seg_u8 = np.ascontiguousarray(seg).view(np.uint8).reshape(-1)
    slots = []
    for _ in range(depth):
        ib, ob = PinnedBuffer(batch * in_b), PinnedBuffer(cap)
        reps = -(-ib.nbytes // seg_u8.size)
        ib.array[:] = np.tile(seg_u8, reps)[:ib.nbytes]
        slots.append((ib, ob))
    n_batches = max(depth, frames_total // batch)
    state = dict(down=0)

    def run():
        flight = []
        for i in range(n_batches):
            if len(flight) == depth:
                chain.collect(flight.pop(0))
            ib, ob = slots[i % depth]
            got, t = chain.submit(ib.ptr, batch, ob.ptr, cap)
            state["down"] += got * out_b
            flight.append(t)
        for t in flight:
            chain.collect(t)

    run()                                             # warm: buffers grown, pages touched
    state["down"] = 0
    dt = timed_region(dist, lambda: None, run, 1)
    for ib, ob in slots:
        ib.free(); ob.free()
    return dt, n_batches * batch, n_batches * batch * in_b, state["down"]


def step_flops(info, frames, n_res, n_emit, desc_kw, ntaps, taps_complex):
    """FP32 flops of one step.  Everything in front of the user filter in the reference's own formulation (SURVEY 8d:
    real x complex MAC = 4, complex x complex = 8; half-band stage i: 2 m_i taps per output at rate / 2^(i+1); polyphase
    14 taps per output).  The user filter twice: `direct` as the direct form it is specified as (what liquid's firfilt
    executes and SURVEY 8d counts), `executed` as the overlap-save block convolution the product runs for filters of
    96 taps and more (two N-point transforms at 5 N log2 N plus the N-point product per N - L + 1 outputs, N by the rule
    of abi.cpp).  Returns (executed, direct)."""
    f = frames * (2.0 + (6.0 if desc_kw.get("shift_hz") else 0.0) + (8.0 if desc_kw.get("dc_block") else 0.0)
                  + (3.0 if desc_kw.get("iq_correct") else 0.0))
    for i in range(info.num_halfband_stages):
        f += (frames / float(1 << (i + 1))) * (2 * info.stage_m[i] * 4 + 2)
    f += n_res * 14 * 4
    f += n_emit * 4.0
    direct = n_emit * ntaps * (8 if taps_complex else 4)
    executed = direct
    if ntaps >= 96:
        lg = 10
        while (1 << lg) < 4 * (ntaps - 1) and (1 << lg) < 4096:
            lg += 1
        while (1 << lg) < 2 * (ntaps - 1):
            lg += 1
        n = 1 << lg
        executed = (n_emit / float(n - (ntaps - 1))) * (2 * 5.0 * n * lg + 6.0 * n)
    return f + executed, f + direct


def run_case(args, dist, dev, local_rank, world, rank, config, preset, steps, warmup, settle_s, log2_frames, shipped=None):
    """One timed leg: build the chain of `config` (2, 3, 4; preset = config 2 with the output AGC; shipped = the name of one of
    PRESETS), put its batch into HBM, settle, warm up, time `steps` steps between barriers.  Returns the raw figures; the caller
    formats them."""
    import torch
    import iq_tool_amd
    from iq_tool_amd import synth
    chain_kw, rate, fmt, in_bps, workload = (dict(CHAIN, agc=True) if preset else CHAIN), 2.4e6, "cs16", 4, None
    if shipped:
        chain_kw = PRESETS[shipped] if shipped in PRESETS else NEAR[shipped]
        fmt = chain_kw["in_format"]
        in_bps = 2 if fmt in ("cu8", "cs8") else 4
        workload = "%s %s on a 2.4 MS/s %s capture" % ("preset" if shipped in PRESETS else "the headline chain", shipped, fmt)
    elif config != 2:
        o = OTHER[config]
        chain_kw, rate, fmt, in_bps, workload = o["chain"], o["rate"], o["fmt"], o["bps"], o["workload"]
        if log2_frames == 28:
            log2_frames = o["log2_frames"]
    frames = 1 << log2_frames
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

    t_settle = time.perf_counter()
    while time.perf_counter() - t_settle < settle_s:                # sustained-load clock, see --settle-seconds
        for _ in range(50):
            step()
        sync()
        out_frames.clear()
    for _ in range(warmup):
        step()
    sync()
    out_frames.clear()
    chain.set_profiling(True)
    chain.profile()                                   # clear
    dt = timed_region(dist, sync, step, steps)
    prof = chain.profile()
    chain.set_profiling(False)

    front = prof["front"]
    k_ms = front["ms"] / max(front["launches"], 1)
    if config != 2 or preset or shipped:
        # these run several kernels per step (cascade, last stage, dc carries, filter, AGC verdict):
        # price the whole step's device time, not one of them
        k_ms = sum(v["ms"] for v in prof.values()) / max(steps, 1)
    n_out_avg = float(np.mean(out_frames)) if out_frames else 0.0
    alg_bytes = frames * in_bps + n_out_avg * chain.out_bytes   # in_bytes + r * out_bytes per input frame
    achieved = alg_bytes / (k_ms * 1e-3) / 1e9 if k_ms > 0 else 0.0
    del d_in, d_out
    return dict(chain=chain, seg=seg, frames=frames, log2_frames=log2_frames, dt=dt, k_ms=k_ms, prof=prof, front=front, n_out_avg=n_out_avg,
                alg_bytes=alg_bytes, achieved=achieved, chain_kw=chain_kw, in_bps=in_bps, workload=workload, steps=steps)


def fp32_roofline(case):
    """configs 3 / 4 are ALU-bound: executed flops of the step against the FP32 vector peak, HBM fraction beside it"""
    chain, frames, k_ms, prof = case["chain"], case["frames"], case["k_ms"], case["prof"]
    info = chain.info()
    n_res = float(-(-((frames >> info.num_halfband_stages) << 24) // info.arb_step))
    flops, flops_direct = step_flops(info, frames, n_res, case["n_out_avg"], case["chain_kw"], int(info.filter_ntaps), info.filter_impl in (2, 4))
    tf = flops / (k_ms * 1e-3) / 1e12 if k_ms > 0 else 0.0
    return {"bound": "fp32", "achieved": round(tf, 2), "peak": 157.3, "unit": "TFLOP/s", "frac": round(tf / 157.3, 4),
            "traffic": None, "kernel": "all kernels of the step", "kernel_ms": round(k_ms, 4), "launches": case["front"]["launches"],
            "executed_flops_per_step": int(flops), "direct_form_flops_per_step": int(flops_direct),
            "direct_form_TFLOPs": round(flops_direct / (k_ms * 1e-3) / 1e12, 2) if k_ms > 0 else 0.0,
            "algorithmic_bytes_per_step": int(case["alg_bytes"]),
            "hbm_GBs": round(case["achieved"], 1), "hbm_frac": round(case["achieved"] / HBM_PEAK_GBS, 4),
            "note": "per-kernel ms: " + ", ".join("%s %.3f" % (k, v["ms"] / max(v["launches"], 1)) for k, v in prof.items() if v["launches"])}


def secondary_traffic(name):
    """counter-traffic ratio (HBM bytes by PMC / algorithmic bytes) of a secondary config, when profiles/traffic.json holds
    a figure measured on this very build of the kernels (same source hash); otherwise null"""
    try:
        with open(os.path.join(ROOT, "profiles", "traffic.json")) as fh:
            t = json.load(fh)
        sec = t.get("secondary", {}).get(name)
        if sec and t.get("all_sources_sha") == all_sources_sha():
            return float(sec["traffic_bytes"])
    except (OSError, ValueError, KeyError):
        pass
    return None


def all_sources_sha():
    h = hashlib.sha256()
    d = os.path.join(ROOT, "iq_tool_amd", "csrc")
    for f in sorted(os.listdir(d)):
        if f.endswith((".hip", ".hpp", ".cpp")):
            with open(os.path.join(d, f), "rb") as fh:
                h.update(fh.read())
    return h.hexdigest()[:16]


def secondary_cases(args, dist, dev, local_rank, world, rank):
    """BASELINE configs[2], configs[3] and the shipped preset (configs[1] + output AGC), each a short timed leg of its own
    AFTER the main timed region: they never touch `value`; they are in the line so that the driver's run carries them."""
    out = {}
    for name, cfg, preset in (("config3", 3, False), ("config4", 4, False), ("preset", 2, True)):
        try:
            out[name] = secondary_case(args, dist, dev, local_rank, world, rank, name, cfg, preset)
        except Exception as exc:       # a secondary leg never costs the line its `value`: the failure is recorded in its place
            out[name] = {"error": "%s: %s" % (type(exc).__name__, exc)}
        import torch
        torch.cuda.empty_cache()
    return out


def secondary_case(args, dist, dev, local_rank, world, rank, name, cfg, preset):
    c = run_case(args, dist, dev, local_rank, world, rank, cfg, preset, args.secondary_steps, 2, args.secondary_settle, 28)
    e = {"ms_per_step": round(c["dt"] / c["steps"] * 1e3, 4), "steps": c["steps"], "frames_per_step": c["frames"],
         "MSps": round(world * c["steps"] * c["frames"] / c["dt"] / 1e6, 1)}
    if preset:
        e["workload"] = "cs16-fm-nrsc5 preset: BASELINE configs[1] + digital output AGC fused past the lock"
        e["roofline"] = {"bound": "hbm", "achieved": round(c["achieved"], 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(c["achieved"] / HBM_PEAK_GBS, 4), "kernel": "all kernels of the step", "kernel_ms": round(c["k_ms"], 4),
                         "note": "per-kernel ms per step: " + ", ".join("%s %.3f" % (k, v["ms"] / max(c["steps"], 1)) for k, v in c["prof"].items() if v["launches"])}
    else:
        e["workload"] = c["workload"]
        e["roofline"] = fp32_roofline(c)
    tb = secondary_traffic(name)
    e["traffic"] = tb
    e["traffic_over_algorithmic"] = round(tb / c["alg_bytes"], 3) if tb else None
    c["chain"].close()
    return e


def shipped_presets(args, dist, dev, local_rank, world, rank):
    """every preset of iq_tool_presets.conf:190-248 as a short device-resident leg: ms per 2^28-frame step, the HBM fraction of the
    whole step (algorithmic bytes in + out over the device time of all its kernels), which kernels ran and what each took"""
    import torch
    out = {}
    only = [x for x in args.presets.split(",") if x]
    for name in PRESETS:
        if only and name not in only:
            continue
        try:
            c = run_case(args, dist, dev, local_rank, world, rank, 2, False, args.secondary_steps, 2, args.preset_settle, 28, shipped=name)
            out[name] = {"ms_per_step": round(c["dt"] / c["steps"] * 1e3, 4), "frames_per_step": c["frames"],
                         "MSps": round(world * c["steps"] * c["frames"] / c["dt"] / 1e6, 1),
                         "hbm_GBs": round(c["achieved"], 1), "frac": round(c["achieved"] / HBM_PEAK_GBS, 4), "kernel_ms": round(c["k_ms"], 4),
                         "front_kernel": c["chain"].front_kernel(),
                         "kernels": {k: round(v["ms"] / max(c["steps"], 1), 4) for k, v in c["prof"].items() if v["launches"]},
                         "timed_stages_per_step": round(sum(v["launches"] for v in c["prof"].values()) / max(c["steps"], 1), 2),
                         "chain": "%s 2.4 MS/s -> %s %.5f kS/s, digital AGC%s" % (PRESETS[name]["in_format"], PRESETS[name]["out_format"],
                                  PRESETS[name]["target_rate_hz"] / 1e3, ", complex band-pass behind the resampler" if "filters" in PRESETS[name] else "")}
            c["chain"].close()
        except Exception as exc:
            out[name] = {"error": "%s: %s" % (type(exc).__name__, exc)}
        torch.cuda.empty_cache()
    return out


def eight_bit_shapes(args, dist, dev, local_rank, world, rank):
    """the headline chain on 8-bit frames (NEAR): ms per 2^28-frame step, HBM fraction, the kernel that ran"""
    import torch
    out = {}
    for name in NEAR:
        try:
            c = run_case(args, dist, dev, local_rank, world, rank, 2, False, args.secondary_steps, 2, args.preset_settle, 28, shipped=name)
            out[name] = {"ms_per_step": round(c["dt"] / c["steps"] * 1e3, 4), "frames_per_step": c["frames"],
                         "MSps": round(world * c["steps"] * c["frames"] / c["dt"] / 1e6, 1),
                         "hbm_GBs": round(c["achieved"], 1), "frac": round(c["achieved"] / HBM_PEAK_GBS, 4), "kernel_ms": round(c["k_ms"], 4),
                         "front_kernel": c["chain"].front_kernel()}
            c["chain"].close()
        except Exception as exc:
            out[name] = {"error": "%s: %s" % (type(exc).__name__, exc)}
        torch.cuda.empty_cache()
    return out


def reference_binary_probe(frames_log2=24):
    """BASELINE.md section 3, optional: when the reference's own `iq_tool` binary (and the libliquid it links) is on this
    box, time it on the configs[1] command line of SURVEY.md 8(d) over a synthetic cs16 file; otherwise say so.  Never
    required, never `value`: this image and the GPU boxes carry neither, so the line normally reports "skipped"."""
    import shutil
    import tempfile
    exe = os.environ.get("IQGPU_REFERENCE_BIN") or shutil.which("iq_tool")
    if not exe or not os.path.exists(exe):
        return {"status": "reference binary not available -- skipped", "looked_for": "iq_tool on PATH, IQGPU_REFERENCE_BIN"}
    from iq_tool_amd import synth
    n = 1 << frames_log2
    with tempfile.TemporaryDirectory() as td:
        src, dst = os.path.join(td, "in.cs16"), os.path.join(td, "out.cs16")
        np.tile(synth.raw_stream(1 << 20, 2.4e6, 1, "cs16"), n >> 20).tofile(src)
        cmd = [exe, "-i", "raw-file", src, "--raw-file-input-rate", "2.4e6", "--raw-file-input-sample-format", "cs16",
               "-o", "raw-file", dst, "--output-rate", "744187.5", "--output-sample-format", "cs16", "--freq-shift", "200e3"]
        env = dict(os.environ)
        if os.environ.get("IQGPU_LIQUID_SO"):
            env["LD_LIBRARY_PATH"] = os.path.dirname(os.environ["IQGPU_LIQUID_SO"]) + ":" + env.get("LD_LIBRARY_PATH", "")
        try:
            t0 = time.perf_counter()
            r = subprocess.run(cmd, env=env, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, timeout=300)
            dt = time.perf_counter() - t0
        except (OSError, subprocess.TimeoutExpired) as exc:
            return {"status": "reference binary found but did not run: %s" % exc, "binary": exe}
        if r.returncode != 0:
            return {"status": "reference binary exited %d: %s" % (r.returncode, r.stderr.decode("utf-8", "replace")[-200:]), "binary": exe}
        out_frames = os.path.getsize(dst) // 4 if os.path.exists(dst) else 0
    return {"status": "ran", "binary": exe, "value": round(n / dt / 1e6, 3), "unit": "MS/s", "frames": n, "out_frames": out_frames,
            "threads": "the reference's own (reader, pre-processor, resampler, post-processor, writer)", "kind": "reference",
            "note": "whole process, file I/O included: `%s`" % " ".join(cmd[:1] + ["..."] + cmd[3:])}


def device_residency_leg(args, dist, dev, local_rank, world, rank, block_samples, steps):
    """configs[1] exactly as BASELINE.json words it -- '256 k-sample blocks': the same device-resident step with
    block_samples = 262144 (fixed runs of tiles per wavefront instead of one run per resident wave).  Same bytes (tested:
    results do not depend on the partition); never `value`."""
    global BLOCK_SAMPLES
    keep = BLOCK_SAMPLES
    BLOCK_SAMPLES = block_samples
    try:
        c = run_case(args, dist, dev, local_rank, world, rank, 2, False, steps, 2, 0.7, args.log2_frames)
    finally:
        BLOCK_SAMPLES = keep
    kern = c["chain"].front_kernel()
    c["chain"].close()
    return {"block_samples": block_samples, "ms_per_step": round(c["dt"] / c["steps"] * 1e3, 4), "steps": c["steps"],
            "MSps": round(world * c["steps"] * c["frames"] / c["dt"] / 1e6, 1), "kernel": kern, "kernel_ms": round(c["k_ms"], 4),
            "frac": round(c["achieved"] / HBM_PEAK_GBS, 4)}


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
    return dict(value=round(max(v1, v3), 3), unit="MS/s", cores=3 if v3 > v1 else 1, cpu=cpu_model(), host_cores=os.cpu_count(), kind="port",
                sample="2^%d cs16 frames of the same NRSC-5 chain, oracle/liboracle_fast.so (float accumulators): "
                       "%.1f MS/s on 1 thread, %.1f MS/s as 3 stage threads (pre / resampler / post, 16384-frame chunks)"
                       % (frames_log2, v1, v3))


def main():
    args = parse()
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        raise SystemExit(spawn_workers(args))         # parent: no torch, no HIP
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("WORLD_SIZE (%d) != --gpus (%d)" % (world, args.gpus))

    if os.environ.get("IQGPU_BENCH_STUB") == "1" and os.environ.get("IQGPU_BENCH_STUB_FAIL_RANK") == str(rank):
        raise SystemExit(3)                            # launcher test: a rank that dies before the rendezvous

    import torch

    # This rank onto the socket of ITS GPU -- CPU mask and preferred memory node -- before anything initialises the GPU (the
    # runtime's helper threads inherit the mask; `import torch` comes first on purpose -- it makes no GPU call, and its bundled HIP
    # runtime has to be the one the process loads: libiqgpu.so loaded ahead of it brings /opt/rocm's and torch then sees no device) and before any pinned buffer exists (iqgpu_bind_thread_to_device reads sysfs
    # only: no HIP call).  The reference is one host process (src/pipeline.c:96-116); the placement of N feeding processes on a
    # two-socket node is this build's to own.  IQGPU_BENCH_NO_BIND=1 leaves the rank where the launcher put it.
    numa = {"numa_node": -1, "numa_pci_bus_id": "", "numa_bound": False}
    if os.environ.get("IQGPU_BENCH_STUB") != "1" and os.environ.get("IQGPU_BENCH_NO_BIND") != "1":
        try:
            import iq_tool_amd as _pkg
            node, bdf, err = _pkg.bind_thread_to_device(local_rank)
            numa = {"numa_node": node, "numa_pci_bus_id": bdf, "numa_bound": err is None and node >= 0}
            if err:
                numa["numa_note"] = err
            numa["cpus_allowed"] = len(os.sched_getaffinity(0))
        except Exception as exc:                       # placement is best effort: never a reason not to measure
            numa["numa_note"] = "%s: %s" % (type(exc).__name__, exc)

    dist = None
    if world > 1:
        import torch.distributed as dist_mod
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        sys.stdout.flush()
        keep = os.dup(1)
        os.dup2(2, 1)                                  # gloo announces its mesh on stdout: rank 0's stdout is ONE JSON line
        try:
            import datetime
            dist_mod.init_process_group(backend="gloo", rank=rank, world_size=world,      # barrier + MAX only
                                        timeout=datetime.timedelta(seconds=float(os.environ.get("IQGPU_BENCH_RDZV_S", "300"))))
            dist_mod.barrier()
        finally:
            os.dup2(keep, 1)
            os.close(keep)
        dist = dist_mod
    if os.environ.get("IQGPU_BENCH_STUB") == "1":
        return stub_main(args, dist, world, rank)     # launcher / reduction plumbing test (tests/test_host_logic.py)
    if args.dry_placement:
        return dry_placement(args, dist, world, rank, local_rank, numa)

    import iq_tool_amd
    from iq_tool_amd import synth

    lib = iq_tool_amd.load()                          # raises when libiqgpu.so is missing
    if lib.iqgpu_device_count() < 1 or not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: no HIP device visible (there is no CPU fallback)")
    local_rank_asked = local_rank
    if os.environ.get("IQGPU_BENCH_SHARE_GPU") == "1":
        local_rank %= lib.iqgpu_device_count()        # plumbing check on a box with fewer GPUs than ranks (never a scaling number)
    if local_rank >= lib.iqgpu_device_count():
        raise SystemExit("rank %d has no GPU: %d devices visible" % (local_rank, lib.iqgpu_device_count()))
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    # which physical device every rank drives (ordinal + PCI bus id, gathered over gloo): an N-rank line shows N distinct GPUs
    import ctypes
    bus = ctypes.create_string_buffer(64)
    bus_rc = lib.iqgpu_device_pci_bus_id(local_rank, bus, 64)
    mine = {"rank": rank, "ordinal": local_rank, "pci_bus_id": bus.value.decode(), "host": socket.gethostname()}
    if bus_rc != 0:
        mine["pci_bus_id_error"] = lib.iqgpu_last_error().decode("utf-8", "replace")
    mine.update(numa)
    if mine["numa_pci_bus_id"]:                        # what sysfs said before the runtime was up against what the runtime says now
        mine["numa_matches_runtime"] = mine["numa_pci_bus_id"].lower() == mine["pci_bus_id"].lower()
    share = os.environ.get("IQGPU_BENCH_SHARE_GPU") == "1"
    if local_rank_asked != local_rank:
        mine["note"] = "rank asked for ordinal %d, shares ordinal %d (IQGPU_BENCH_SHARE_GPU)" % (local_rank_asked, local_rank)
    devices = [mine]
    if dist is not None:
        devices = [None] * world
        dist.all_gather_object(devices, mine)
    if world > 1 and not share:
        ids = [(d["host"], d["pci_bus_id"]) for d in devices]
        if any(d.get("pci_bus_id_error") for d in devices) or len(set(ids)) != world:
            raise SystemExit("bench.py: %d ranks do not sit on %d distinct GPUs: %s" % (world, world, devices))

    case = run_case(args, dist, dev, local_rank, world, rank, args.config, args.preset, args.steps, args.warmup, args.settle_seconds, args.log2_frames)
    front_kernel = case["chain"].front_kernel()
    chain, seg, frames, dt, k_ms, prof, front, n_out_avg, alg_bytes, achieved, chain_kw, in_bps, workload = (
        case[k] for k in ("chain", "seg", "frames", "dt", "k_ms", "prof", "front", "n_out_avg", "alg_bytes", "achieved", "chain_kw", "in_bps", "workload"))
    value = world * args.steps * frames / dt / 1e6

    # ---- second leg: the same chain fed from pinned host memory (PCIe-inclusive; never `value`) ----
    host = None
    if not args.no_host_leg and args.config == 2 and not args.preset:
        chain.set_stream(0)                            # back to the chain's own stream
        chain.reset()
        h_dt, h_frames, h_up, h_down = host_leg(dist, chain, seg, 1 << args.host_log2_frames, 1 << args.host_batch_log2)
        host = {"value": round(world * h_frames / h_dt / 1e6, 2), "unit": "MS/s",
                "h2d_GBs": round(h_up / h_dt / 1e9, 2), "d2h_GBs": round(h_down / h_dt / 1e9, 2),
                "frames_per_gpu": h_frames, "batch_frames": 1 << args.host_batch_log2,
                "path": "pinned host buffers -> iqgpu_chain_submit (H2D, kernels, D2H staged by the host, %d batches in flight) -> iqgpu_chain_collect; h2d/d2h per GPU" % chain._lib.iqgpu_chain_pipeline_depth()}
    # ---- ... and at the batch size of the INTEGRATION.md binding: 16 reference chunks = 262144 frames per submit ----
    host_stub = None
    if host is not None and not args.no_extra:
        chain.reset()
        s_dt, s_frames, s_up, s_down = host_leg(dist, chain, seg, 1 << args.stub_log2_frames, args.stub_batch_frames)
        host_stub = {"value": round(world * s_frames / s_dt / 1e6, 2), "unit": "MS/s", "batch_frames": args.stub_batch_frames,
                     "us_per_batch": round(s_dt / (s_frames / args.stub_batch_frames) * 1e6, 2),
                     "h2d_GBs": round(s_up / s_dt / 1e9, 2), "d2h_GBs": round(s_down / s_dt / 1e9, 2), "frames_per_gpu": s_frames,
                     "kernel": chain.front_kernel(),
                     "path": "as host_end_to_end, at the batch the INTEGRATION.md section 2 stub submits when its reader queue holds 16 chunks (16 x PIPELINE_CHUNK_BASE_SAMPLES, include/constants.h:123)"}
    # ... and at the batch the stub takes when the reader runs ahead (a file input: the pool holds 512 chunks, include/constants.h:110)
    host_stub64 = None
    if host_stub is not None:
        chain.reset()
        s_dt, s_frames, s_up, s_down = host_leg(dist, chain, seg, 1 << (args.stub_log2_frames + 1), args.stub64_batch_frames)
        host_stub64 = {"value": round(world * s_frames / s_dt / 1e6, 2), "unit": "MS/s", "batch_frames": args.stub64_batch_frames,
                       "us_per_batch": round(s_dt / (s_frames / args.stub64_batch_frames) * 1e6, 2),
                       "h2d_GBs": round(s_up / s_dt / 1e9, 2), "d2h_GBs": round(s_down / s_dt / 1e9, 2), "frames_per_gpu": s_frames,
                       "kernel": chain.front_kernel(),
                       "path": "as host_end_to_end_stub, at the 64-chunk batch the stub takes when reader_output_queue is deep (file inputs)"}

    if rank == 0:
        line = {
            "metric": "complex MS/s end-to-end on NRSC-5 resample+filter chain; % HBM roofline",
            "value": round(value, 2), "unit": "MS/s", "value_is": "device_resident (inputs in HBM when the timed region starts; host_end_to_end is the PCIe-inclusive rate)",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 4), "settle_s": args.settle_seconds,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic (seeded 2^%d-frame cs16 segment, 3 tones + noise + DC, tiled in HBM to 2^%d frames per GPU)" % (int(np.log2(min(frames, 1 << SEGMENT_LOG2))), case["log2_frames"]),
            "config": {"workload": "BASELINE configs[1]: raw cs16 2.4 MS/s -> 744.1875 kS/s, +200 kHz NCO, 1 half-band (m=10) + 256-arm polyphase (14 taps), cs16 out",
                       "frames_per_step_per_gpu": frames, "block_samples": BLOCK_SAMPLES, "devices": devices,
                       # the library reads NO switch from the environment (ABI v6): "debug" is what iqgpu_debug_list reports -- the
                       # diagnostic switches the ctypes mirror forwarded from this process's IQGPU_<NAME> variables; "env" keeps every
                       # IQGPU_* variable the process saw (bench.py's own knobs included), so a stray export is on record
                       "debug": iq_tool_amd._lib.debug_switches(),
                       "env": {k: v for k, v in sorted(os.environ.items()) if k.startswith("IQGPU_")},
                       "sharding": "independent stream per GPU, no collective (gloo barrier + MAX only)"
                                   + (" -- RANKS SHARE ONE GPU (IQGPU_BENCH_SHARE_GPU): launcher check, not a scaling figure" if os.environ.get("IQGPU_BENCH_SHARE_GPU") == "1" else "")},
            "roofline": {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": measured_traffic(case["log2_frames"]) if args.config == 2 else None,
                         "kernel": front_kernel + " (iqgpu_chain_front_kernel: what the timed launches ran)", "kernel_ms": round(k_ms, 4), "launches": front["launches"],
                         "algorithmic_bytes_per_launch": int(alg_bytes),
                         "read_only_frac": round(frames * in_bps / (k_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if k_ms > 0 else 0.0},
            "host_end_to_end": host,
        }
        if args.config != 2:
            line["config"]["workload"] = workload
            line["roofline"] = fp32_roofline(case)
        if DIAG_TARGET_HZ:
            line["config"]["workload"] = "DIAGNOSTIC (IQGPU_BENCH_TARGET_HZ=%s), not a BASELINE config: " % DIAG_TARGET_HZ + str(line["config"]["workload"])
        if args.preset:
            line["config"]["workload"] = ("cs16-fm-nrsc5 preset (iq_tool_presets.conf:216-222): BASELINE configs[1] + digital output AGC -- fused into the "
                                          "front kernel past the 2 s lock, verified by k_agc_classify (classification + verdict, one launch); per-kernel ms: "
                                          + ", ".join("%s %.3f" % (k, v["ms"] / max(args.steps, 1)) for k, v in prof.items() if v["launches"]))
            line["roofline"]["kernel"] = front_kernel + " with the fused AGC + k_agc_classify (+ verdict)"
            line["roofline"]["traffic"] = None
        line["cpu_baseline"] = None                    # filled in below, once every GPU leg of every rank is done
    # ---- the other single-GPU BASELINE configs and the shipped preset: short legs of their own, after everything that feeds `value`
    # (N = 1 only: they are per-GPU figures, and a leg that failed on one rank of several would leave the others in a barrier)
    sec = extra = None
    if args.config == 2 and not args.preset and world == 1:
        case = chain = None
        torch.cuda.empty_cache()
        if not args.no_extra:
            extra = {"host_end_to_end_stub": host_stub, "host_end_to_end_stub64": host_stub64}
            try:
                extra["block_samples_262144"] = device_residency_leg(args, dist, dev, local_rank, world, rank, 262144, args.secondary_steps)
            except Exception as exc:
                extra["block_samples_262144"] = {"error": "%s: %s" % (type(exc).__name__, exc)}
            torch.cuda.empty_cache()
            try:
                extra["reference_binary"] = reference_binary_probe()
            except Exception as exc:
                extra["reference_binary"] = {"status": "probe failed: %s: %s" % (type(exc).__name__, exc)}
        if not args.no_secondary:
            sec = {} if args.only_presets else secondary_cases(args, dist, dev, local_rank, world, rank)
            try:
                sec["presets"] = shipped_presets(args, dist, dev, local_rank, world, rank)
            except Exception as exc:
                sec["presets"] = {"error": "%s: %s" % (type(exc).__name__, exc)}
            try:
                sec["eight_bit"] = eight_bit_shapes(args, dist, dev, local_rank, world, rank)
            except Exception as exc:
                sec["eight_bit"] = {"error": "%s: %s" % (type(exc).__name__, exc)}
    # ---- the CPU pipeline "in the same run" (north_star), at every N: rank 0 times it on its own host cores once the GPU legs of
    # all ranks are behind the barrier (the other ranks are done and idle; a 1-rank run needs no barrier)
    if dist is not None:
        dist.barrier()
    if rank == 0:
        if not args.no_cpu_baseline and args.config == 2 and not args.preset:
            try:
                line["cpu_baseline"] = cpu_baseline(args.cpu_frames_log2)
            except Exception as exc:
                line["cpu_baseline"] = {"error": "%s: %s" % (type(exc).__name__, exc)}
        line["secondary"] = sec
        line["extra"] = extra
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.destroy_process_group()


def dry_placement(args, dist, world, rank, local_rank, numa):
    """--dry-placement: the N-rank run up to -- and without -- its first GPU call.  What has run by the time this
    function is entered is the real thing: fresh child processes, `import torch`, the sysfs walk and the binding of this rank to its
    GPU's NUMA node, the gloo rendezvous and its first barrier.  Here: every rank reports where it sits and what it WOULD allocate,
    rank 0 checks that N ranks mean N distinct PCI addresses (from sysfs: there is no runtime to ask) on the right nodes, and one
    empty timed region exercises barrier + MAX.  No HIP call anywhere (iqgpu_device_count is never asked)."""
    import iq_tool_amd
    from iq_tool_amd import chain as chain_mod
    frames = 1 << args.log2_frames
    kw = dict(in_format="cs16", out_format="cs16", input_rate_hz=2.4e6, target_rate_hz=744187.5, shift_hz=200e3)
    n_out = chain_mod.design_out_frames(frames, **kw)                     # the create-time design path: host only
    depth = 8
    host_batch = 1 << args.host_batch_log2
    host_cap = chain_mod.design_out_frames(host_batch, **kw) + 64
    mine = {"rank": rank, "ordinal": local_rank, "host": socket.gethostname(), "pid": os.getpid(),
            "hbm_bytes": frames * 4 + (n_out + 64) * 4,
            "pinned_bytes": depth * (host_batch * 4 + host_cap * 4),
            "cpus": sorted(os.sched_getaffinity(0))}
    mine.update(numa)
    devices = [mine]
    if dist is not None:
        devices = [None] * world
        dist.all_gather_object(devices, mine)
    dt = timed_region(dist, lambda: None, lambda: None, args.steps)
    problems = []
    ids = [(d["host"], d["numa_pci_bus_id"]) for d in devices]
    if any(not d["numa_pci_bus_id"] for d in devices):
        problems.append("a rank could not find the PCI address of its GPU in sysfs")
    elif len(set(ids)) != world:
        problems.append("%d ranks do not sit on %d distinct GPUs" % (world, world))
    if os.environ.get("IQGPU_BENCH_NO_BIND") != "1":
        for d in devices:
            if not d["numa_bound"]:
                problems.append("rank %d is not bound: %s" % (d["rank"], d.get("numa_note", "no reason given")))
    if rank == 0:
        print(json.dumps({"metric": "dry-placement (no GPU work)", "n_gpus": world, "steps": args.steps, "barrier_and_max_s": dt,
                          "frames_per_step_per_gpu": frames, "out_frames_per_step_per_gpu": n_out, "devices": devices,
                          "distinct_gpus": len(set(ids)), "ok": not problems, "problems": problems,
                          "sharding": "independent stream per GPU, no collective (gloo barrier + MAX only)"}), flush=True)
    if dist is not None:
        dist.destroy_process_group()
    if problems:
        raise SystemExit(4)


def stub_main(args, dist, world, rank):
    """IQGPU_BENCH_STUB=1: the launcher, barrier and MAX-over-ranks plumbing with a sleep in place of the GPU
    step -- what tests/test_host_logic.py runs on a machine without a GPU.  Never a measurement."""
    plan = shard_plan(world, rank, 1 << args.log2_frames)
    dt = timed_region(dist, lambda: None, lambda: time.sleep(0.002 * (1 + rank)), args.steps)
    if rank == 0:
        print(json.dumps({"metric": "stub", "value": world * args.steps * plan["frames"] / dt / 1e6, "unit": "MS/s", "n_gpus": world,
                          "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3, "data": "stub (no GPU work)",
                          "scaling": "weak"}), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
