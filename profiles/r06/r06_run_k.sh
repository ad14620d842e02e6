#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06k
timeout 1500 python -m pytest tests/test_gs_hip.py tests/test_mesh_hip.py tests/test_zmesh_ext.py tests/test_zz_replay_gpu.py -m gpu -q -x --deselect tests/test_zz_replay_gpu.py::test_the_drivers_eight_rank_command_with_the_all_gather_exchange 2>&1 | grep -v "^$" | tail -15 > gpurun_out/r06k/pytest.log; tail -15 gpurun_out/r06k/pytest.log
echo "== node defaults (10k Gaussians, 512^2, batch 1): r05 tree with its event timing inside the timed region | r05 tree --timed-prof off | work"
for i in 1 2 3; do
  bash profiles/ab_tree_run.sh r06k/ref_r05_timed_$i "r05" 1 --workload ref-default --ref-res 512 --steps 600 --warmup 50
  bash profiles/ab_tree_run.sh r06k/ref_r05_untimed_$i "r05" 1 --workload ref-default --ref-res 512 --steps 600 --warmup 50 --timed-prof off
  bash profiles/ab_tree_run.sh r06k/ref_work_$i "work" 1 --workload ref-default --ref-res 512 --steps 600 --warmup 50
done
