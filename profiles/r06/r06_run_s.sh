#!/bin/bash
# round 6, second session: k_sum_group_loss with the views' terms in flight together, no key write in the tile sort's last pass, histogram workgroups over several tiles
# (profiles/ab_prepare.sh s_head head; s_work work)
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_gs_hip.py tests/test_knn.py -m gpu -x -q 2>&1 | tail -5
bash profiles/ab_run.sh r06s/step "s_head s_work" 3 - | cut -c1-400
bash profiles/ab_run.sh r06s/fwd64 "s_head s_work" 2 - --mode fwd --views-per-gpu 64 --steps 10 | cut -c1-400
bash profiles/ab_run.sh r06s/boundary "s_head s_work" 2 - --render-path boundary --steps 20 | cut -c1-400
