#!/bin/bash
# The boxes of the pool fall into two classes.  Probe the default step first; run the full collection only on a box of the faster class (step < $2 ms, default 4.90),
# so that one set of evidence files exists for each class: usage  bash profiles/collect_if_fast.sh r05z [4.90]
cd $GRAFT_REPO_ROOT
LIM=${2:-4.90}
ms=$(timeout 300 python bench.py --steps 20 --warmup 5 --cpu-baseline off --targets off 2>/dev/null | tail -1 | python -c "import sys, json; print(json.loads(sys.stdin.read())['ms_per_step'])")
echo "probe: $ms ms per step (limit $LIM)"
if python -c "import sys; sys.exit(0 if float('$ms') < float('$LIM') else 1)"; then bash profiles/collect.sh $1; else echo "slower-class box: not collecting"; fi
