#!/bin/bash
# round 3, GPU call A: full GPU suite (incl. the new 1M multi-camera tests) + A/B of the two forward compositing kernels
set -x
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r03a; mkdir -p $OUT; cd $R
timeout 1500 python -m pytest tests -m gpu -q -s 2>&1 | grep -v "amdgpu.ids" > $OUT/pytest.log; tail -40 $OUT/pytest.log
python bench.py --steps 10 --warmup 3 --cpu-baseline off 2>/dev/null | tail -1 > $OUT/bench_new.json
C3D_FWD_KERNEL=0 python bench.py --steps 10 --warmup 3 --cpu-baseline off 2>/dev/null | tail -1 > $OUT/bench_old.json
python bench.py --mode fwd --views-per-gpu 64 --steps 3 --cpu-baseline off 2>/dev/null | tail -1 > $OUT/bench_fwd64_new.json
C3D_FWD_KERNEL=0 python bench.py --mode fwd --views-per-gpu 64 --steps 3 --cpu-baseline off 2>/dev/null | tail -1 > $OUT/bench_fwd64_old.json
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/kt
rocprofv3 --kernel-trace -d /tmp/kt -o kt -- python $R/bench.py --lanes 1 --steps 2 --warmup 1 --cpu-baseline off --timed-prof off > /tmp/kt.log 2>&1
python $R/profiles/summarize_rocpd.py $(find /tmp/kt -name "*.db" | head -1) > $OUT/kernel_stats_lanes1.csv
cd $R
for f in $OUT/bench_*.json; do echo $f; python profiles/benchline.py < $f; done
head -14 $OUT/kernel_stats_lanes1.csv
