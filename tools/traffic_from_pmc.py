#!/usr/bin/env python3
"""HBM bytes per step from the separate rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; KiB as rocprofv3 reports them):
traffic = 2 x FETCH_SIZE + WRITE_SIZE (gfx950 tallies the 128-byte requests of 16-byte-per-lane streaming reads at
64 bytes: MI355X_MICROARCH.md, HBM), summed over every kernel of the step (all iqgpu kernels: one dispatch of each per step).
Writes profiles/traffic.json: the headline kernel keyed by the hash of its sources, the secondary configs keyed by the hash
of all kernel sources; bench.py reports a figure only for the very build and workload size it was measured on.
usage: tools/traffic_from_pmc.py <gpurun_out/prof_rNN directory of tools/profile_round.sh>"""
import collections
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def per_step(path, counter, steps):
    """sum over all iqgpu kernels of (total counter / steps): bench runs `steps` timed + warm-up steps, every step launches
    the same kernels, so per-kernel averages per dispatch add up to the step's figure"""
    tot = collections.defaultdict(float)
    ids = collections.defaultdict(set)
    for r in csv.DictReader(open(path)):
        if "iqgpu" in r["Kernel_Name"] and r["Counter_Name"] == counter:
            k = r["Kernel_Name"].split("(")[0]
            tot[k] += float(r["Counter_Value"])
            ids[k].add(r["Dispatch_Id"])
    if not tot:
        raise SystemExit("no %s rows in %s" % (counter, path))
    n_steps = max(len(v) for v in ids.values())
    return {k: tot[k] / n_steps for k in tot}, n_steps


def one(out, cfg):
    f = glob.glob(os.path.join(out, "hbm_%s_FETCH_SIZE" % cfg, "**", "*counter_collection.csv"), recursive=True)
    w = glob.glob(os.path.join(out, "hbm_%s_WRITE_SIZE" % cfg, "**", "*counter_collection.csv"), recursive=True)
    if not f or not w:
        return None
    fetch, nf = per_step(f[0], "FETCH_SIZE", None)
    write, nw = per_step(w[0], "WRITE_SIZE", None)
    fk, wk = sum(fetch.values()), sum(write.values())
    return dict(fetch_size_KiB=fk, write_size_KiB=wk, steps=[nf, nw], traffic_bytes=(2.0 * fk + wk) * 1024.0,
                per_kernel_KiB={k: [round(fetch.get(k, 0.0), 1), round(write.get(k, 0.0), 1)] for k in sorted(set(fetch) | set(write))})


def main():
    out = sys.argv[1]
    head = one(out, "2")
    res = dict(kernel="k_front_mid<6, false, 4, 0, false, false>", kernel_sha=bench.kernel_sha(), all_sources_sha=bench.all_sources_sha(), log2_frames=28,
               rule="2 x FETCH_SIZE + WRITE_SIZE (KiB -> bytes), per step, separate --pmc passes (tools/profile_round.sh)")
    if head:
        res.update(head)
    res["secondary"] = {}
    # (the preset is left out: its first steps run the unfused AGC kernels until the 2 s lock, so a per-step average over the
    #  profiled run mixes two regimes)
    for cfg, name in (("3", "config3"), ("4", "config4")):
        r = one(out, cfg)
        if r:
            res["secondary"][name] = r
    with open(os.path.join(ROOT, "profiles", "traffic.json"), "w") as fh:
        json.dump(res, fh, indent=1)
    print(json.dumps({k: v for k, v in res.items() if k != "secondary"}))
    for k, v in res["secondary"].items():
        print(k, v["traffic_bytes"])


if __name__ == "__main__":
    main()
