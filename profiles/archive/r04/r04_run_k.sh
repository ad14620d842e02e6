#!/bin/bash
# mesh tests + mesh bench
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04k2
timeout 1200 python -m pytest tests/test_mesh_hip.py tests/test_zmesh_ext.py tests/test_zz_ref_consumers.py tests/test_zz_replay_gpu.py -m gpu -x -q 2>&1 | tail -4
for i in 1 2; do timeout 600 python bench.py --workload mesh --steps 20 --warmup 3 --cpu-baseline off 2>/dev/null | tail -1 | tee gpurun_out/r04k2/bench_mesh_$i.json | python profiles/benchline.py; done
