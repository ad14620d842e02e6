#!/bin/bash
# round 6, second session: the histogram kernel counts the TOP digit's two values by ballot (profiles/ab_prepare.sh ae_old work "-DC3D_NO_HIST_TOP2"; ae_new work)
cd $GRAFT_REPO_ROOT
bash profiles/ab_run.sh r06ae/step "ae_old ae_new" 3 "sort or forward_matches or internal_state or edge or golden or unequal" | cut -c1-400
bash profiles/ab_run.sh r06ae/inference "ae_old ae_new" 2 - --render-path boundary --mode fwd --inference-mode on --steps 20 | cut -c1-300
timeout 600 python -m pytest tests/test_knn.py tests/test_mesh_hip.py -m gpu -x -q 2>&1 | tail -2
