#!/bin/bash
# round 6: k_front's pointwise emission (chains that do not decimate: r >= 1 in front of k_interp, --no-resample) with one store per
# thread instead of four: previous library against the new one, same box
cd ${GRAFT_REPO_ROOT:-/root/repo}
for i in 1 2 3; do for v in prev new; do L=iq_tool_amd/lib/libiqgpu.so; [ $v = prev ] && L=iq_tool_amd/lib/libiqgpu_prev.so
[ -f $L ] || continue
echo "$v $(IQGPU_LIB=$PWD/$L python3 tools/bench_chain.py --in-rate 2.0e6 --out-rate 2.4e6 --log2-frames 25 --steps 30 2>&1 | tail -1)"
echo "$v $(IQGPU_LIB=$PWD/$L python3 tools/bench_chain.py --in-rate 2.4e6 --out-rate 2.4e6 --shift 100e3 --log2-frames 26 --steps 30 2>&1 | tail -1)"
echo "$v $(IQGPU_LIB=$PWD/$L python3 tools/bench_chain.py --in-format cu8 --out-format cu8 --in-rate 2.4e6 --out-rate 2.4e6 --shift 100e3 --log2-frames 26 --steps 30 2>&1 | tail -1)"
done; done
timeout -k 10 500 python3 -m pytest tests -m gpu -x -q --timeout 400 -k "convert or interp or ratio or no_resample or pointwise or operator or random_chain" 2>&1 | tail -3
