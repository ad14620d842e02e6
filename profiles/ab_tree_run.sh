#!/bin/bash
# same-box A/B of whole trees (profiles/ab_tree_prepare.sh): profiles/ab_tree_run.sh <tag> "<tree names; work = the working tree>" <rounds> [bench args]
cd $GRAFT_REPO_ROOT
T=$1; VARS=$2; R=${3:-3}; shift 3
mkdir -p gpurun_out/$T
for i in $(seq 1 $R); do for v in $VARS; do
  if [ "$v" = work ]; then d=$GRAFT_REPO_ROOT; else d=$GRAFT_REPO_ROOT/profiles/_ab/trees/$v; fi
  (cd $d && timeout 300 python bench.py --cpu-baseline off --targets off "$@" 2>/dev/null | tail -1) > gpurun_out/$T/${v}_$i.json; echo "[$v]"; python profiles/benchline.py < gpurun_out/$T/${v}_$i.json
done; done
