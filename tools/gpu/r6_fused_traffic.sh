#!/bin/bash
# round 6: HBM traffic and kernel statistics of the cu8-nrsc5-usb preset on the two-kernel path and on k_p0fft16 (IQGPU_FUSE_FILTER=1):
# rocprofv3 --kernel-trace --stats, then FETCH_SIZE and WRITE_SIZE in separate --pmc passes (never combined with tracing).
# -> gpurun_out/r6/fused_traffic/, summary in gpurun_out/r6/fused_traffic.txt (copied to profiles/r06_fused_traffic.txt by hand)
set -u
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/r6/fused_traffic
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
B="python3 $REPO/bench.py --only-presets --presets cu8-nrsc5-usb --no-cpu-baseline --no-host-leg --no-extra --secondary-steps 5 --preset-settle 0"
for v in two fused; do
  unset IQGPU_FUSE_FILTER; [ $v = fused ] && export IQGPU_FUSE_FILTER=1
  rocprofv3 --kernel-trace --stats -d "$OUT/stats_$v" -o stats --output-format csv -- $B > "$OUT/stats_$v.log" 2>&1
  for c in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $c -d "$OUT/hbm_${v}_$c" -o pmc --output-format csv -- $B > "$OUT/hbm_${v}_$c.log" 2>&1
  done
  echo "$v done"
done
unset IQGPU_FUSE_FILTER
cd "$REPO"
python3 - <<'PY' | tee gpurun_out/r6/fused_traffic.txt
import csv, glob, collections, os
out = "gpurun_out/r6/fused_traffic"
print("# cu8-nrsc5-usb preset, 2^28 cu8 frames per step: algorithmic bytes = 2^28 * 2 (in) + n_out * 2 (out)")
for v in ("two", "fused"):
    print("## %s" % ("two kernels (k_front_p0<cf32> + k_fftconv16<10>)" if v == "two" else "k_p0fft16 (IQGPU_FUSE_FILTER=1)"))
    f = glob.glob(os.path.join(out, "stats_%s" % v, "**", "*kernel_stats.csv"), recursive=True)
    if f:
        for r in csv.DictReader(open(f[0])):
            if "iqgpu" in r["Name"] and float(r["Percentage"]) > 0.5:
                print("   stats %-90s calls %5s avg %10.1f us" % (r["Name"].split("(")[0][:90], r["Calls"], float(r["AverageNs"]) / 1e3))
    tot = {}
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        g = glob.glob(os.path.join(out, "hbm_%s_%s" % (v, c), "**", "*counter_collection.csv"), recursive=True)
        if not g:
            continue
        per = collections.defaultdict(list)
        for r in csv.DictReader(open(g[0])):
            if "iqgpu" in r["Kernel_Name"] and r["Counter_Name"] == c:
                per[r["Kernel_Name"].split("(")[0]].append(float(r["Counter_Value"]))
        for k, vals in sorted(per.items()):
            big = [x for x in vals if x > 0.5 * max(vals)]          # (the steady-state launches: the first calls run the unfused AGC kernels)
            avg = sum(big) / len(big)
            if avg > 1024:
                print("   %-11s %-90s %10.1f MiB per launch (%d launches)" % (c, k[:90], avg / 1024.0, len(big)))
                tot[(c, k)] = avg
    fetch = sum(x for (c, k), x in tot.items() if c == "FETCH_SIZE"); write = sum(x for (c, k), x in tot.items() if c == "WRITE_SIZE")
    traffic = (2.0 * fetch + write) * 1024.0
    import sys; sys.path.insert(0, '.')
    import iq_tool_amd
    n_out = iq_tool_amd.design_out_frames(1 << 28, in_format='cu8', out_format='cu8', input_rate_hz=2.4e6, target_rate_hz=1488375.0, filters=(('passband', 158.5e3, 113e3),), agc=True)
    alg = (1 << 28) * 2 + n_out * 2.0
    print("   traffic 2 x FETCH + WRITE = %.3f GB per step; algorithmic %.3f GB; ratio %.2f" % (traffic / 1e9, alg / 1e9, traffic / alg))
PY
find "$OUT" -name '*.csv' -size +1M -delete
