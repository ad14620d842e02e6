#!/bin/bash
# round 3, GPU call I: two-stage loss sum + staggered projection (C3D_PRE_SPLIT) -- fused-step tests, bench A/B
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r03i; mkdir -p $OUT; cd $R
timeout 600 python -m pytest tests/test_gs_hip.py tests/test_zz_replay_gpu.py -m gpu -q -x -k "fused or step or binning_chains or train or replay or loss" < /dev/null 2>&1 | grep -v "amdgpu.ids" > $OUT/pytest_a.log; tail -6 $OUT/pytest_a.log
for v in 1 0 1 0; do
C3D_PRE_SPLIT=$v timeout 300 python bench.py --steps 30 --warmup 5 --cpu-baseline off --targets off < /dev/null 2>/dev/null | tail -1 > $OUT/bench_default_split${v}_$RANDOM.json
done
C3D_PRE_SPLIT=1 timeout 300 python bench.py --mode train --steps 20 --warmup 5 --cpu-baseline off --targets off < /dev/null 2>/dev/null | tail -1 > $OUT/bench_train_split1.json
C3D_PRE_SPLIT=0 timeout 300 python bench.py --mode train --steps 20 --warmup 5 --cpu-baseline off --targets off < /dev/null 2>/dev/null | tail -1 > $OUT/bench_train_split0.json
for f in $OUT/bench_*.json; do echo $f; head -c 210 $f | tail -c 110; echo; done
