#!/bin/bash
# round 6: the consumer role of a producer / consumer split of k_front_mid by itself (tools/pc_consumer_bench.hip): timing, then the
# LDS / VALU counters of its kernels in separate rocprofv3 --pmc passes.
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd "$REPO"; O=$REPO/gpurun_out/r6; mkdir -p $O
timeout -k 10 120 tools/pc_consumer_bench 4000 | tee $O/pc_consumer.txt
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS"; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --pmc $set -d $O/pc_pmc_$i -o pmc --output-format csv -- $REPO/tools/pc_consumer_bench 1000 > $O/pc_pmc_$i.log 2>&1
done
cd "$REPO"
python3 - <<'PY' | tee -a gpurun_out/r6/pc_consumer.txt
import csv, glob, collections
print("# counters per dispatch (rocprofv3 --pmc, 1000 steps per wave; 4 launches per configuration, in the order of the table above)")
for p in sorted(glob.glob("gpurun_out/r6/pc_pmc_*/**/*counter_collection.csv", recursive=True)):
    rows = collections.OrderedDict()
    for r in csv.DictReader(open(p)):
        key = (r["Dispatch_Id"], r["Kernel_Name"].split("(")[0][-40:], r.get("Workgroup_Size", r.get("Workgroup_Size_X", "")))
        rows.setdefault(key, {})[r["Counter_Name"]] = float(r["Counter_Value"])
    for (d, k, wg), c in rows.items():
        print("  dispatch %3s %-32s wg %5s  " % (d, k, wg) + "  ".join("%s %.4g" % kv for kv in sorted(c.items())))
PY
