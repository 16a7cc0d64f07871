cd $GRAFT_REPO_ROOT
hipcc --offload-arch=gfx950 -O3 -o /tmp/occ_probe tools/occ_probe.hip && /tmp/occ_probe
