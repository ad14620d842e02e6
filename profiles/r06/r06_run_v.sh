#!/bin/bash
# round 6, second session: histogram kernel counts the upper digits by ballot (up to three values per wave); k_sort_small takes its digits from key - min(key) and skips passes
# that have nothing to move (profiles/ab_prepare.sh v_head head; v_work work)
cd $GRAFT_REPO_ROOT
bash profiles/ab_run.sh r06v/step "v_head v_work" 3 "scan or sort or forward_matches or internal_state or edge or config1 or golden or unequal or densif" | cut -c1-400
bash profiles/ab_run.sh r06v/boundary "v_head v_work" 2 - --render-path boundary --mode fwd --inference-mode on --steps 20 | cut -c1-400
bash profiles/ab_run.sh r06v/refdefault "v_head v_work" 3 - --workload ref-default --ref-res 512 --steps 600 --warmup 50 --timed-prof off | cut -c1-100
timeout 600 python -m pytest tests/test_knn.py tests/test_zz_replay_gpu.py tests/test_mesh_hip.py -m gpu -x -q 2>&1 | tail -3
