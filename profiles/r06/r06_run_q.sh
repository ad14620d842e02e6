#!/bin/bash
# round 6, whole-round same-box A/B: round 5's final tree (profiles/_ab/trees/r05, `ab_tree_prepare.sh r05 15bffc3`) against the final tree of round 6, every bench line
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06q
for i in 1 2 3; do
  bash profiles/ab_tree_run.sh r06q/step "r05 work" 1 --steps 30 --warmup 5 | cut -c1-44
done
for i in 1 2; do
  bash profiles/ab_tree_run.sh r06q/fwd64 "r05 work" 1 --mode fwd --views-per-gpu 64 --steps 10 --warmup 3 | cut -c1-44
  bash profiles/ab_tree_run.sh r06q/train "r05 work" 1 --mode train --steps 20 --warmup 5 | cut -c1-44
  bash profiles/ab_tree_run.sh r06q/mesh "r05 work" 1 --workload mesh --steps 40 --warmup 5 | cut -c1-44
  bash profiles/ab_tree_run.sh r06q/boundary "r05 work" 1 --render-path boundary --steps 20 --warmup 5 | cut -c1-44
  bash profiles/ab_tree_run.sh r06q/inference "r05 work" 1 --render-path boundary --mode fwd --steps 20 --warmup 5 | cut -c1-44
  bash profiles/ab_tree_run.sh r06q/refdefault "r05 work" 1 --workload ref-default --ref-res 512 --steps 600 --warmup 50 --timed-prof off | cut -c1-44
done
