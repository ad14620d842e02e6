#!/bin/bash
# round 3, GPU call H: kernel stats of the per-tile depth sort (single lane, kernels alone)
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r03h; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $OUT/prof -o run -- python $R/bench.py --steps 3 --warmup 2 --lanes 1 --cpu-baseline off --targets off --timed-prof off > $OUT/bench.log 2>&1
f=$(find $OUT/prof -name "*kernel_stats.csv" | head -1); head -25 $f | cut -c1-160 > $OUT/kernel_stats_head.csv; cat $OUT/kernel_stats_head.csv
rm -rf $OUT/prof
