#!/bin/bash
# rocprofv3 kernel trace of the default bench command (lanes 1: kernels run alone) -> per-kernel stats
T=${1:-r04kt}
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/$T
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/kt
timeout 240 rocprofv3 --kernel-trace -d /tmp/kt -o kt -- python $R/bench.py --lanes 1 --steps 3 --warmup 1 --cpu-baseline off --targets off --timed-prof off < /dev/null > /tmp/kt.log 2>&1
python $R/profiles/summarize_rocpd.py $(find /tmp/kt -name "*.db" | head -1) > $OUT/${T}_fwdbwd_kernel_stats.csv
head -40 $OUT/${T}_fwdbwd_kernel_stats.csv
