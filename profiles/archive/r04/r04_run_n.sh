#!/bin/bash
# kernel trace of the training step (config 3) and of the mesh step: where MS-SSIM's 1.8 ms per 8 views goes
T=r04n
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/$T
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/ktt /tmp/km
timeout 240 rocprofv3 --kernel-trace -d /tmp/ktt -o ktt -- python $R/bench.py --mode train --steps 3 --warmup 1 --cpu-baseline off --targets off --timed-prof off < /dev/null > /tmp/ktt.log 2>&1
python $R/profiles/summarize_rocpd.py $(find /tmp/ktt -name "*.db" | head -1) > $OUT/${T}_train_kernel_stats.csv
timeout 240 rocprofv3 --kernel-trace -d /tmp/km -o km -- python $R/bench.py --workload mesh --steps 3 --warmup 1 --cpu-baseline off --timed-prof off < /dev/null > /tmp/km.log 2>&1
python $R/profiles/summarize_rocpd.py $(find /tmp/km -name "*.db" | head -1) > $OUT/${T}_mesh_kernel_stats.csv
head -45 $OUT/${T}_train_kernel_stats.csv | cut -c1-150
head -30 $OUT/${T}_mesh_kernel_stats.csv | cut -c1-150
