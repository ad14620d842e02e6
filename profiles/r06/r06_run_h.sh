#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06h
timeout 1500 python -m pytest tests/test_mesh_hip.py tests/test_zmesh_ext.py tests/test_ref_pin.py tests/test_zz_ref_consumers.py tests/test_zz_replay_gpu.py -m gpu -q -x --deselect tests/test_zz_replay_gpu.py::test_the_drivers_eight_rank_command_with_the_all_gather_exchange 2>&1 | grep -v "^$" | tail -30 > gpurun_out/r06h/pytest_mesh.log; tail -30 gpurun_out/r06h/pytest_mesh.log
echo "== mesh step (config 5, 8 views): global atomics | tile-binned"
bash profiles/ab_run.sh r06h "ras_atomic ras_binned" 3 - --workload mesh --steps 40 --warmup 5
