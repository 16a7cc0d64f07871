#!/bin/bash
# round 6: k_p0fft16 (resampler -> user filter in one kernel).  Parity first, then the presets' timing and the NCO-hold A/B of the
# headline kernel (profiles/r06_headline.md).  Run from the repo root on the GPU box: bash tools/gpu/r6_fuse.sh
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd "$REPO"; O=gpurun_out/r6; mkdir -p $O
timeout -k 10 900 python3 -m pytest tests -m gpu -x -q --timeout 600 -k "resampler_and_filter_in_one_kernel or p0_kernel_equals or fused_in_the_filter_epilogue or s0_chain or non_finite or process_device_behind or random_chain" > $O/fuse_tests.log 2>&1
echo "tests rc=$?" | tee -a $O/fuse_tests.log
tail -5 $O/fuse_tests.log
timeout -k 10 300 python3 bench.py --only-presets --no-cpu-baseline --no-host-leg --no-extra > $O/fuse_presets.json 2> $O/fuse_presets.err
python3 - <<'PY'
import json
j = json.load(open("gpurun_out/r6/fuse_presets.json"))
for k, v in j["secondary"]["presets"].items():
    print(k, v["ms_per_step"], v["frac"], v["front_kernel"], v["kernels"])
PY
if [ -f iq_tool_amd/lib/libiqgpu_ncohold.so ]; then bash tools/abn.sh new ncohold > $O/ncohold_ab.txt 2>&1; cat $O/ncohold_ab.txt; fi
