#!/bin/bash
# rocprofv3 kernel trace of a single-lane bench run (kernels alone): per-kernel durations.  usage: bash profiles/run_ktrace.sh <tag> [bench args]
TAG=${1:-rXX}; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/kt
timeout 600 rocprofv3 --kernel-trace -d /tmp/kt -o kt -- python $R/bench.py --lanes 1 --steps 3 --warmup 1 --cpu-baseline off --timed-prof off "$@" > $OUT/ktrace.log 2>&1
python $R/profiles/summarize_rocpd.py $(find /tmp/kt -name "*.db" | head -1) > $OUT/ktrace_kernel_stats.csv
head -40 $OUT/ktrace_kernel_stats.csv | cut -c1-160
