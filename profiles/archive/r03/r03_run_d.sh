#!/bin/bash
# round 3, GPU call D: GPU suite (device densify, level gradients, SSAA, 1M attribution), bench lines of the current code
set -x
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r03d; mkdir -p $OUT; cd $R
timeout 1500 python -m pytest tests -m gpu -q -s 2>&1 | grep -v "amdgpu.ids" > $OUT/pytest.log; tail -5 $OUT/pytest.log
python bench.py --steps 20 --warmup 5 --cpu-baseline off 2>/dev/null | tail -1 > $OUT/bench_default.json
python bench.py --mode train --steps 20 --warmup 5 --cpu-baseline off --targets off 2>/dev/null | tail -1 > $OUT/bench_train.json
python bench.py --workload mesh --steps 20 --warmup 5 --cpu-baseline off 2>/dev/null | tail -1 > $OUT/bench_mesh.json
GPU_MAX_HW_QUEUES=8 C3D_BIN_AHEAD=1 python bench.py --steps 20 --warmup 5 --cpu-baseline off --targets off 2>/dev/null | tail -1 > $OUT/bench_binahead_q8.json
GPU_MAX_HW_QUEUES=8 python bench.py --steps 20 --warmup 5 --cpu-baseline off --targets off 2>/dev/null | tail -1 > $OUT/bench_q8.json
for f in $OUT/bench_*.json; do echo $f; head -c 300 $f; echo; done
grep -n "passed\|failed\|FAILED" $OUT/pytest.log | cut -c1-300
