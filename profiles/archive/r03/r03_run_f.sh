#!/bin/bash
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r03f; mkdir -p $OUT; cd $R
timeout 900 python -m pytest tests/test_mesh_hip.py tests/test_zmesh_ext.py tests/test_zz_replay_gpu.py tests/test_ref_train_loop.py -m gpu -q -s 2>&1 | grep -v "amdgpu.ids" > $OUT/pytest_mesh.log; tail -30 $OUT/pytest_mesh.log
timeout 300 python bench.py --workload mesh --steps 40 --warmup 5 --cpu-baseline off 2>/dev/null | tail -1 > $OUT/bench_mesh.json; head -c 400 $OUT/bench_mesh.json
