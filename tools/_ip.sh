OUT=$PWD/gpurun_out/ip; mkdir -p $OUT; REPO=$PWD
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $OUT -o s --output-format csv -- python3 $REPO/tools/bench_chain.py --in-rate 1e6 --out-rate 1.6e6 --log2-frames 26 --steps 5 > $OUT/log 2>&1
head -5 $(find $OUT -name '*kernel_stats.csv' | head -1) | cut -c1-120
rocprofv3 --kernel-trace --stats -d $OUT/b -o s --output-format csv -- python3 $REPO/tools/bench_chain.py --in-rate 250e3 --out-rate 2.4e6 --log2-frames 24 --steps 5 > $OUT/log2 2>&1
head -5 $(find $OUT/b -name '*kernel_stats.csv' | head -1) | cut -c1-120
