#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06f
timeout 900 python -m pytest tests/test_zz_replay_gpu.py -m gpu -q -s -x -k "overflow_on_one_rank" 2>&1 | grep -v "^$" | grep -v "^E  *File\|^E  *return\|^E  *raise\|^E  *main\|^E  *run(\|^E  *exec\|^E  *elastic" | head -60 > gpurun_out/r06f/pytest_ovf.log; cat gpurun_out/r06f/pytest_ovf.log
timeout 1500 python -m pytest tests/test_gs_hip.py tests/test_zz_baseline_1m.py tests/test_zz_ref_consumers.py -m gpu -q -x 2>&1 | grep -v "^$" | tail -30 > gpurun_out/r06f/pytest_rest.log; tail -30 gpurun_out/r06f/pytest_rest.log
echo "== default step: r05 tree | work"
bash profiles/ab_tree_run.sh r06f/step "r05 work" 3 --steps 30 --warmup 5
echo "== forward 64 cameras: r05 | work"
bash profiles/ab_tree_run.sh r06f/fwd64 "r05 work" 2 --mode fwd --views-per-gpu 64 --steps 10 --warmup 3
