#!/bin/bash
# round 5: emit + histogram fusion, second look at the 8-view step: six alternations of 40 steps, eh first this time
cd $GRAFT_REPO_ROOT
P=comfyui-3d-pack_amd; C=$P/csrc
for i in 1 2 3 4 5 6; do for v in eh base; do
  rm -rf $C; cp -r profiles/_ab/$v/csrc $C; cp profiles/_ab/$v/libc3d_hip.so profiles/_ab/$v/libc3d_hip.digest $P/lib/
  echo "[$v]"; timeout 300 python bench.py --steps 40 --warmup 5 --targets off --cpu-baseline off 2>/dev/null | python profiles/benchline.py
done; done
