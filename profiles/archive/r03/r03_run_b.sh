#!/bin/bash
# round 3, GPU call B: GPU suite on the wave-per-quadrant forward + backward compositing kernels, A/B against the workgroup-per-tile ones
set -x
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r03b; mkdir -p $OUT; cd $R
timeout 1500 python -m pytest tests -m gpu -q -s 2>&1 | grep -v "amdgpu.ids" > $OUT/pytest.log; tail -30 $OUT/pytest.log
python bench.py --steps 10 --warmup 3 --cpu-baseline off 2>/dev/null | tail -1 > $OUT/bench_new.json
C3D_BWD_KERNEL=0 python bench.py --steps 10 --warmup 3 --cpu-baseline off 2>/dev/null | tail -1 > $OUT/bench_bwd_old.json
C3D_FWD_KERNEL=0 C3D_BWD_KERNEL=0 python bench.py --steps 10 --warmup 3 --cpu-baseline off 2>/dev/null | tail -1 > $OUT/bench_both_old.json
python bench.py --mode fwd --views-per-gpu 64 --steps 3 --cpu-baseline off 2>/dev/null | tail -1 > $OUT/bench_fwd64_new.json
python bench.py --mode train --steps 10 --warmup 3 --cpu-baseline off 2>/dev/null | tail -1 > $OUT/bench_train_new.json
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/kt
rocprofv3 --kernel-trace -d /tmp/kt -o kt -- python $R/bench.py --lanes 1 --steps 2 --warmup 1 --cpu-baseline off --timed-prof off > /tmp/kt.log 2>&1
python $R/profiles/summarize_rocpd.py $(find /tmp/kt -name "*.db" | head -1) > $OUT/kernel_stats_lanes1.csv
cd $R
for f in $OUT/bench_*.json; do echo $f; python profiles/benchline.py < $f; done
head -16 $OUT/kernel_stats_lanes1.csv
grep -n "\[1M\|passed\|failed" $OUT/pytest.log | cut -c1-400
