#!/bin/bash
# re-entry baseline of round 4: whole -m gpu suite, kernel traces (default step, 64-view forward), bench lines of configs 2-5
T=r04l
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/$T
mkdir -p $OUT
cd $R
timeout 1200 python -m pytest tests -m gpu -q -rP < /dev/null > $OUT/${T}_gpu_pytest.log 2>&1
echo "pytest rc=$?" | tee -a $OUT/${T}_gpu_pytest.log
grep -E "passed|failed|^FAILED|^ERROR" $OUT/${T}_gpu_pytest.log | tail -20
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/kt /tmp/kf
timeout 240 rocprofv3 --kernel-trace -d /tmp/kt -o kt -- python $R/bench.py --steps 3 --warmup 1 --cpu-baseline off --targets off --timed-prof off < /dev/null > /tmp/kt.log 2>&1
python $R/profiles/summarize_rocpd.py $(find /tmp/kt -name "*.db" | head -1) > $OUT/${T}_fwdbwd_kernel_stats.csv
timeout 240 rocprofv3 --kernel-trace -d /tmp/kf -o kf -- python $R/bench.py --mode fwd --views-per-gpu 64 --steps 2 --warmup 1 --cpu-baseline off --targets off --timed-prof off < /dev/null > /tmp/kf.log 2>&1
python $R/profiles/summarize_rocpd.py $(find /tmp/kf -name "*.db" | head -1) > $OUT/${T}_fwd64_kernel_stats.csv
cd $R
timeout 400 python bench.py --steps 20 --warmup 5 < /dev/null 2>/dev/null | tail -1 > $OUT/${T}_bench_n1.json
timeout 300 python bench.py --mode fwd --views-per-gpu 64 --steps 20 --warmup 3 --cpu-baseline off --targets off < /dev/null 2>/dev/null | tail -1 > $OUT/${T}_bench_fwd64_n1.json
timeout 300 python bench.py --mode train --steps 20 --warmup 5 --cpu-baseline off --targets off < /dev/null 2>/dev/null | tail -1 > $OUT/${T}_bench_train_n1.json
timeout 400 python bench.py --workload mesh --steps 40 --warmup 5 --cpu-baseline off < /dev/null 2>/dev/null | tail -1 > $OUT/${T}_bench_mesh_n1.json
cat $OUT/${T}_fwdbwd_kernel_stats.csv | head -40
cat $OUT/${T}_fwd64_kernel_stats.csv | head -30
for f in $OUT/${T}_bench_*.json; do echo $f; timeout 20 python profiles/benchline.py < $f; done
