#!/bin/bash
# Collect the committed evidence of a round on the GPU box: bench lines, rocprofv3 kernel stats, PMC traffic (separate passes).
# usage (from the repo root, through gpurun):  bash profiles/collect.sh r01g
set -x
TAG=${1:-rXX}
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
python bench.py 2>/dev/null | tail -1 > $OUT/${TAG}_bench_n1.json
python bench.py --mode train --cpu-baseline off 2>/dev/null | tail -1 > $OUT/${TAG}_bench_train_n1.json
python bench.py --mode fwd --cpu-baseline off 2>/dev/null | tail -1 > $OUT/${TAG}_bench_fwd_n1.json
python bench.py --workload mesh 2>/dev/null | tail -1 > $OUT/${TAG}_bench_mesh_n1.json
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/kt /tmp/pf /tmp/pw /tmp/km
rocprofv3 --kernel-trace -d /tmp/kt -o kt -- python $R/bench.py --steps 3 --warmup 1 --cpu-baseline off > /tmp/kt.log 2>&1
python $R/profiles/summarize_rocpd.py $(find /tmp/kt -name "*.db" | head -1) > $OUT/${TAG}_fwdbwd_kernel_stats.csv
rocprofv3 --kernel-trace -d /tmp/km -o km -- python $R/bench.py --workload mesh --steps 2 --warmup 1 > /tmp/km.log 2>&1
python $R/profiles/summarize_rocpd.py $(find /tmp/km -name "*.db" | head -1) > $OUT/${TAG}_mesh_kernel_stats.csv
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d /tmp/pf -o pf -- python $R/bench.py --lanes 1 --steps 2 --warmup 1 --cpu-baseline off > /tmp/pf.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d /tmp/pw -o pw -- python $R/bench.py --lanes 1 --steps 2 --warmup 1 --cpu-baseline off > /tmp/pw.log 2>&1
python $R/profiles/summarize_pmc.py $(find /tmp/pf -name "*.db" | head -1) $(find /tmp/pw -name "*.db" | head -1) $OUT/${TAG}_pmc_traffic.json > $OUT/${TAG}_pmc_traffic.csv
head -12 $OUT/${TAG}_fwdbwd_kernel_stats.csv
cat $OUT/${TAG}_pmc_traffic.csv | head -12
cut -c1-400 $OUT/${TAG}_bench_n1.json
