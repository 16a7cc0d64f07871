#!/usr/bin/env python3
"""HBM bytes per k_front_s1 launch from the two separate rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; KiB as
rocprofv3 reports them): traffic = 2 x FETCH_SIZE + WRITE_SIZE (gfx950 tallies the 128-byte requests of
16-byte-per-lane streaming reads at 64 bytes: MI355X_MICROARCH.md, HBM).  Writes profiles/traffic.json keyed by
the hash of the kernel's sources; bench.py reports the figure only for that very build and workload size.
usage: tools/traffic_from_pmc.py <fetch counter_collection.csv> <write counter_collection.csv> [log2_frames]"""
import csv
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def avg(path, counter, pat="k_front_s1<4, true"):
    tot, ids = 0.0, set()
    for r in csv.DictReader(open(path)):
        if pat in r["Kernel_Name"] and r["Counter_Name"] == counter:
            tot += float(r["Counter_Value"])
            ids.add(r["Dispatch_Id"])
    if not ids:
        raise SystemExit("no %s rows for %s in %s" % (counter, pat, path))
    return tot / len(ids), len(ids)


def main():
    fetch, nf = avg(sys.argv[1], "FETCH_SIZE")
    write, nw = avg(sys.argv[2], "WRITE_SIZE")
    log2_frames = int(sys.argv[3]) if len(sys.argv) > 3 else 28
    traffic = (2.0 * fetch + write) * 1024.0
    out = dict(kernel="k_front_s1<4, true, false, false, false>", kernel_sha=bench.kernel_sha(), log2_frames=log2_frames,
               fetch_size_KiB=fetch, write_size_KiB=write, dispatches=[nf, nw], traffic_bytes=traffic,
               rule="2 x FETCH_SIZE + WRITE_SIZE (KiB -> bytes), per launch, separate --pmc passes")
    with open(os.path.join(ROOT, "profiles", "traffic.json"), "w") as fh:
        json.dump(out, fh, indent=1)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
