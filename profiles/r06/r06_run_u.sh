#!/bin/bash
# round 6, second session: the depth sort's last pass leaves the packed rects in rank order (k_onesweep GATHER), the emit-offset scan streams them
# (profiles/ab_prepare.sh u_nog work "-DC3D_NO_SORT_GATHER"; u_gat work)
cd $GRAFT_REPO_ROOT
bash profiles/ab_run.sh r06u/step "u_nog u_gat" 3 "scan or sort or forward_matches or internal_state or edge or wider or unequal or config1 or golden" | cut -c1-400
bash profiles/ab_run.sh r06u/fwd64 "u_nog u_gat" 2 - --mode fwd --views-per-gpu 64 --steps 10 | cut -c1-400
bash profiles/ab_run.sh r06u/inference "u_nog u_gat" 2 - --render-path boundary --mode fwd --inference-mode on --steps 20 | cut -c1-400
bash profiles/ab_run.sh r06u/refdefault "u_nog u_gat" 2 - --workload ref-default --ref-res 512 --steps 600 --warmup 50 --timed-prof off | cut -c1-100
