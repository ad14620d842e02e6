#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/r06i
rm -rf /tmp/km
timeout 300 rocprofv3 --kernel-trace -d /tmp/km -o km -- python $R/bench.py --workload mesh --steps 3 --warmup 1 --cpu-baseline off < /dev/null > /tmp/km.log 2>&1
python $R/profiles/summarize_rocpd.py $(find /tmp/km -name "*.db" | head -1) > $R/gpurun_out/r06i/mesh_kernel_stats.csv
head -25 $R/gpurun_out/r06i/mesh_kernel_stats.csv
