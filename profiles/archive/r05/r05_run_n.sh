#!/bin/bash
# the whole round on one box: round 4's final tree (commit 2abb685, unpacked under _old_tree/) against the working tree, alternating, default bench line + training step
cd $GRAFT_REPO_ROOT
T=r05n
mkdir -p gpurun_out/$T
for i in 0 1 2; do
  timeout 300 python bench.py --steps 30 --warmup 3 --targets off --cpu-baseline off 2>/dev/null | tail -1 > gpurun_out/$T/new_$i.json; echo "[round 5]"; python profiles/benchline.py < gpurun_out/$T/new_$i.json
  (cd _old_tree && timeout 400 python bench.py --steps 30 --warmup 3 --targets off --cpu-baseline off 2>/dev/null | tail -1) > gpurun_out/$T/old_$i.json; echo "[round 4]"; python profiles/benchline.py < gpurun_out/$T/old_$i.json
done
for i in 0 1; do
  timeout 300 python bench.py --mode train --steps 20 --warmup 3 --targets off --cpu-baseline off 2>/dev/null | tail -1 > gpurun_out/$T/train_new_$i.json; echo "[round 5 train]"; python profiles/benchline.py < gpurun_out/$T/train_new_$i.json
  (cd _old_tree && timeout 400 python bench.py --mode train --steps 20 --warmup 3 --targets off --cpu-baseline off 2>/dev/null | tail -1) > gpurun_out/$T/train_old_$i.json; echo "[round 4 train]"; python profiles/benchline.py < gpurun_out/$T/train_old_$i.json
done
