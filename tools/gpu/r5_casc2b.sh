#!/bin/bash
# k_cascade2 on 16-bit frames: parity, the cs16-am-nrsc5 preset under IQGPU_NO_CASC2, LDS counters of the preset's cascade
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd "$REPO"; mkdir -p gpurun_out/r5_casc2b
timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "two_tile_trips" > gpurun_out/r5_casc2b/tests.log 2>&1 || { tail -30 gpurun_out/r5_casc2b/tests.log; exit 1; }
tail -3 gpurun_out/r5_casc2b/tests.log
for i in 1 2; do
  for v in new old; do
    E=""; [ $v = old ] && E="IQGPU_NO_CASC2=1"
    env $E python3 tools/gpu/r5_am.py $v
  done
done | tee gpurun_out/r5_casc2b/ab_am.txt
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS -d "$REPO/gpurun_out/r5_casc2b/pmc" -o pmc --output-format csv -- python3 $REPO/tools/gpu/r5_am.py fused > "$REPO/gpurun_out/r5_casc2b/pmc.log" 2>&1
cd "$REPO"
f=$(find gpurun_out/r5_casc2b/pmc -name '*counter_collection.csv' < /dev/null | head -1)
[ -n "$f" ] && python3 tools/pmc_summary.py "$f" k_cascade | tee gpurun_out/r5_casc2b/pmc.txt
find gpurun_out/r5_casc2b -name '*.csv' -size +1M -delete
