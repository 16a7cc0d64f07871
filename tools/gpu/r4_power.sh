cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4p
hipcc --offload-arch=gfx950 -O3 -o /tmp/power_probe tools/power_probe.hip || exit 1
timeout -k 10 500 /tmp/power_probe ${1:-1.5} ${2:-0} > gpurun_out/r4p/power_probe.txt 2>&1; echo rc=$?
cat gpurun_out/r4p/power_probe.txt
