#!/bin/bash
# round 3, GPU call M: mesh op extensions ('zero' with the mip filters, rast_db gradient in rasterize) -- mesh test files + mesh bench
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r03m; mkdir -p $OUT; cd $R
timeout 900 python -m pytest tests/test_mesh_hip.py tests/test_zmesh_ext.py tests/test_ref_pin.py -m gpu -q < /dev/null 2>&1 | grep -v "amdgpu.ids" > $OUT/pytest_mesh.log; tail -25 $OUT/pytest_mesh.log | cut -c1-250
timeout 300 python bench.py --workload mesh --steps 40 --warmup 5 --cpu-baseline off < /dev/null 2>/dev/null | tail -1 > $OUT/bench_mesh.json; head -c 250 $OUT/bench_mesh.json
