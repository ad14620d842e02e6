#!/bin/bash
# round 3, GPU call E: GPU suite on the fused mesh pixel passes; mesh bench with the fusion on / off; default line
set -x
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r03e; mkdir -p $OUT; cd $R
timeout 1500 python -m pytest tests -m gpu -q -s 2>&1 | grep -v "amdgpu.ids" > $OUT/pytest.log; tail -5 $OUT/pytest.log
C3D_MESH_PIXEL_FUSED=1 python bench.py --workload mesh --steps 40 --warmup 5 --cpu-baseline off 2>/dev/null | tail -1 > $OUT/bench_mesh_fused.json
C3D_MESH_PIXEL_FUSED=0 python bench.py --workload mesh --steps 40 --warmup 5 --cpu-baseline off 2>/dev/null | tail -1 > $OUT/bench_mesh_unfused.json
C3D_MESH_PIXEL_FUSED=1 python bench.py --workload mesh --mode fwd --steps 40 --warmup 5 --cpu-baseline off 2>/dev/null | tail -1 > $OUT/bench_mesh_fwd_fused.json
C3D_MESH_PIXEL_FUSED=0 python bench.py --workload mesh --mode fwd --steps 40 --warmup 5 --cpu-baseline off 2>/dev/null | tail -1 > $OUT/bench_mesh_fwd_unfused.json
python bench.py --steps 20 --warmup 5 --cpu-baseline off 2>/dev/null | tail -1 > $OUT/bench_default.json
for f in $OUT/bench_*.json; do echo $f; head -c 260 $f; echo; done
grep -n "passed\|failed\|FAILED" $OUT/pytest.log | cut -c1-300
