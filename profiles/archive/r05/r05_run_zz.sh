#!/bin/bash
# round 5, last A/B: the tree before the sort / scan work of the round's second half (commit e02c21b, digest 43fe0b14c1acd5b8) against the final tree, alternating on one box;
# default step, the 64-camera forward pass, the training step
bash profiles/ab_run.sh r05zz2 "before final" 4 -
for i in 1 2; do for v in before final; do
  P=comfyui-3d-pack_amd; C=$P/csrc; rm -rf $C; cp -r profiles/_ab/$v/csrc $C; cp profiles/_ab/$v/libc3d_hip.so profiles/_ab/$v/libc3d_hip.digest $P/lib/
  echo "[$v] fwd64"; timeout 300 python bench.py --mode fwd --views-per-gpu 64 --steps 10 --warmup 3 --cpu-baseline off --targets off 2>/dev/null | python profiles/benchline.py
  echo "[$v] train"; timeout 300 python bench.py --mode train --steps 15 --warmup 4 --cpu-baseline off --targets off 2>/dev/null | python profiles/benchline.py
done; done
