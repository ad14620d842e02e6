#!/bin/bash
# rocprofv3 kernel traces: forward-only (64 views, groups of 8, one group in flight) and the default fwd+bwd command
T=${1:-r04kt}
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/$T
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/kt /tmp/kf
timeout 240 rocprofv3 --kernel-trace -d /tmp/kf -o kf -- python $R/bench.py --mode fwd --views-per-gpu 64 --lanes 1 --group 8 --steps 2 --warmup 1 --cpu-baseline off --targets off --timed-prof off < /dev/null > /tmp/kf.log 2>&1
python $R/profiles/summarize_rocpd.py $(find /tmp/kf -name "*.db" | head -1) > $OUT/${T}_fwd64_kernel_stats.csv
grep 'k_' $OUT/${T}_fwd64_kernel_stats.csv | head -16
timeout 240 rocprofv3 --kernel-trace -d /tmp/kt -o kt -- python $R/bench.py --lanes 1 --steps 3 --warmup 1 --cpu-baseline off --targets off --timed-prof off < /dev/null > /tmp/kt.log 2>&1
python $R/profiles/summarize_rocpd.py $(find /tmp/kt -name "*.db" | head -1) > $OUT/${T}_fwdbwd_kernel_stats.csv
grep 'k_' $OUT/${T}_fwdbwd_kernel_stats.csv | head -22
