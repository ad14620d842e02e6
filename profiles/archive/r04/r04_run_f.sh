#!/bin/bash
# round 4: mesh step with the views as a grid dimension -- mesh parity tests, consumer replays, mesh bench
mkdir -p gpurun_out/r04f
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 1500 python -m pytest tests/test_mesh_hip.py tests/test_zz_ref_consumers.py tests/test_ref_runs.py -m gpu -x -q > gpurun_out/r04f/pytest_mesh.log 2>&1; echo "pytest rc $?" >> gpurun_out/r04f/pytest_mesh.log
tail -5 gpurun_out/r04f/pytest_mesh.log
for i in 1 2 3; do timeout 600 python bench.py --workload mesh --steps 20 --warmup 3 --cpu-baseline off > gpurun_out/r04f/bench_mesh_$i.json 2> gpurun_out/r04f/bench_mesh_$i.err; tail -c 1500 gpurun_out/r04f/bench_mesh_$i.json | head -c 600; echo; done
