#!/bin/bash
# round 6: LDS / VALU / wait counters of the cu8-nrsc5-usb preset's kernels on the two-kernel path and on k_p0fft16
# (IQGPU_FUSE_FILTER=1), one rocprofv3 --pmc pass per counter set (never combined with tracing).
set -u
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out/r6/fused_pmc
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
B="python3 $REPO/bench.py --only-presets --presets cu8-nrsc5-usb --no-cpu-baseline --no-host-leg --no-extra --secondary-steps 5 --preset-settle 0 --steps 2 --warmup 1 --settle-seconds 0"
SETS=("SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES" "SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" "GRBM_GUI_ACTIVE SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY" "SQ_BUSY_CYCLES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU")
for v in two fused; do
  unset IQGPU_FUSE_FILTER; [ $v = fused ] && export IQGPU_FUSE_FILTER=1
  i=0
  for set in "${SETS[@]}"; do
    i=$((i+1))
    rocprofv3 --pmc $set -d "$OUT/${v}_$i" -o pmc --output-format csv -- $B > "$OUT/${v}_$i.log" 2>&1
  done
  echo "$v done"
done
unset IQGPU_FUSE_FILTER
cd "$REPO"
python3 - <<'PY' | tee gpurun_out/r6/fused_pmc.txt
import csv, glob, collections
print("# cu8-nrsc5-usb preset, 2^28 cu8 frames per launch: counters per launch (median over the steady-state launches), rocprofv3 --pmc, one pass per set")
WANT = ("k_front_p0", "k_fftconv16<10, false>", "k_p0fft16")
for v in ("two", "fused"):
    print("## %s" % ("two kernels" if v == "two" else "k_p0fft16 (IQGPU_FUSE_FILTER=1)"))
    agg = collections.defaultdict(dict)
    for p in sorted(glob.glob("gpurun_out/r6/fused_pmc/%s_*/**/*counter_collection.csv" % v, recursive=True)):
        per = collections.defaultdict(list)
        for r in csv.DictReader(open(p)):
            if any(w in r["Kernel_Name"] for w in WANT):
                per[(r["Kernel_Name"].split("(")[0], r["Counter_Name"])].append(float(r["Counter_Value"]))
        for (k, c), vals in per.items():
            vals = sorted(vals); agg[k][c] = vals[len(vals) // 2]
    for k, cs in sorted(agg.items()):
        print("  %s" % k)
        print("     " + "  ".join("%s %.4g" % kv for kv in sorted(cs.items())))
PY
find "$OUT" -name '*.csv' -size +1M -delete
