#!/bin/bash
# round 3, GPU call J: kernel timeline of the default step under 4 view lanes (sqlite from rocprofv3 --kernel-trace)
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r03j; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace -d $OUT/prof -o run -- python $R/bench.py --steps 4 --warmup 3 --cpu-baseline off --targets off --timed-prof off < /dev/null > $OUT/bench.log 2>&1
ls -la $OUT/prof | head
