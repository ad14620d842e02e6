#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06g
timeout 2400 python -m pytest tests -m gpu -q -x --deselect tests/test_zz_replay_gpu.py::test_the_drivers_eight_rank_command_with_the_all_gather_exchange 2>&1 | grep -v "^$" | tail -30 > gpurun_out/r06g/pytest_gpu.log; tail -30 gpurun_out/r06g/pytest_gpu.log
echo "== default step: r05 tree | work"
bash profiles/ab_tree_run.sh r06g/step "r05 work" 3 --steps 30 --warmup 5
