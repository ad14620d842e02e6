#!/bin/bash
# round 3, GPU call K: compositing token (C3D_COMP_TOKENS) A/B on the default and the training line
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r03k; mkdir -p $OUT; cd $R
for v in 0 1 2 0 1 2; do
C3D_COMP_TOKENS=$v timeout 300 python bench.py --steps 30 --warmup 5 --cpu-baseline off --targets off < /dev/null 2>/dev/null | tail -1 > $OUT/bench_default_tok${v}_$RANDOM.json
done
for v in 0 1 2; do
C3D_COMP_TOKENS=$v timeout 300 python bench.py --mode train --steps 20 --warmup 5 --cpu-baseline off --targets off < /dev/null 2>/dev/null | tail -1 > $OUT/bench_train_tok$v.json
done
for f in $OUT/bench_*.json; do echo $f; head -c 210 $f | tail -c 120; echo; done
timeout 600 python -m pytest tests/test_gs_hip.py tests/test_zz_replay_gpu.py -m gpu -q -x -k "fused or step or train or replay or loss" < /dev/null 2>&1 | grep -v "amdgpu.ids" > $OUT/pytest_a.log; tail -4 $OUT/pytest_a.log
