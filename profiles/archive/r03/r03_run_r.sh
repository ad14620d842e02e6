#!/bin/bash
# round 3, GPU call R: kernel trace of the reference's default workload (which launches make up an iteration?)
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r03r; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace -d $OUT/prof -o run -- python $R/bench.py --workload ref-default --ref-res 512 --steps 90 --warmup 10 --cpu-baseline off < /dev/null > $OUT/bench.log 2>&1
ls -la $OUT/prof | head -5
