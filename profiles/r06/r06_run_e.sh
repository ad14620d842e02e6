#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06e
timeout 900 python -m pytest tests/test_zz_baseline_1m.py -m gpu -q -s -x -k "64_orbit" 2>&1 | grep -v "^$" | tail -45 > gpurun_out/r06e/pytest_64.log; tail -45 gpurun_out/r06e/pytest_64.log
timeout 900 python -m pytest tests/test_zz_replay_gpu.py -m gpu -q -s -x -k "overflow_on_one_rank" 2>&1 | grep -v "^$" | tail -30 > gpurun_out/r06e/pytest_ovf.log; tail -30 gpurun_out/r06e/pytest_ovf.log
timeout 1500 python -m pytest tests/test_gs_hip.py tests/test_zz_replay_gpu.py tests/test_zz_ref_consumers.py -m gpu -q -x --deselect tests/test_zz_replay_gpu.py::test_the_drivers_eight_rank_command_with_the_all_gather_exchange 2>&1 | grep -v "^$" | tail -30 > gpurun_out/r06e/pytest_rest.log; tail -30 gpurun_out/r06e/pytest_rest.log
echo "== default step: r05 tree | work"
bash profiles/ab_tree_run.sh r06e/step "r05 work" 3 --steps 30 --warmup 5
echo "== forward 64 cameras: r05 | work"
bash profiles/ab_tree_run.sh r06e/fwd64 "r05 work" 2 --mode fwd --views-per-gpu 64 --steps 10 --warmup 3
