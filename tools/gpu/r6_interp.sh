#!/bin/bash
# round 6: k_interp's last stage storing its (even, odd) output pair as one piece: previous library against the new one
cd ${GRAFT_REPO_ROOT:-/root/repo}
for i in 1 2 3; do for v in prev new; do L=iq_tool_amd/lib/libiqgpu.so; [ $v = prev ] && L=iq_tool_amd/lib/libiqgpu_prev.so
[ -f $L ] || continue
echo "$v $(IQGPU_LIB=$PWD/$L python3 tools/bench_chain.py --in-rate 600e3 --out-rate 2.4e6 --log2-frames 25 --steps 30 2>&1 | tail -1)"
echo "$v $(IQGPU_LIB=$PWD/$L python3 tools/bench_chain.py --in-rate 1.0e6 --out-rate 2.5e6 --log2-frames 25 --steps 30 2>&1 | tail -1)"
echo "$v $(IQGPU_LIB=$PWD/$L python3 tools/bench_chain.py --in-format cu8 --out-format cu8 --in-rate 250e3 --out-rate 2.4e6 --log2-frames 24 --steps 30 2>&1 | tail -1)"
done; done
timeout -k 10 500 python3 -m pytest tests -m gpu -x -q --timeout 400 -k "interp or ratio or random_chain or late" 2>&1 | tail -3
