#!/bin/bash
# round 3, GPU call G: per-tile depth order (C3D_BIN_LOCAL=1) -- primitive test, 3DGS suite, bench A/B
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r03g; mkdir -p $OUT; cd $R
timeout 600 python -m pytest tests/test_gs_hip.py -m gpu -q -x -k "segment_sort or scan or sort_pairs or forward or internal_state or golden or backward" 2>&1 | grep -v "amdgpu.ids" > $OUT/pytest_a.log; tail -15 $OUT/pytest_a.log
for v in 1 0; do
C3D_BIN_LOCAL=$v timeout 300 python bench.py --steps 20 --warmup 5 --cpu-baseline off 2>/dev/null | tail -1 > $OUT/bench_default_local$v.json
C3D_BIN_LOCAL=$v timeout 300 python bench.py --mode fwd --views-per-gpu 64 --steps 5 --warmup 2 --cpu-baseline off --targets off 2>/dev/null | tail -1 > $OUT/bench_fwd64_local$v.json
done
for f in $OUT/bench_*.json; do echo $f; head -c 230 $f; echo; done
