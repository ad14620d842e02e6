#!/bin/bash
# same-box A/B of one source file: working tree vs the version in profiles/_ab_head_<name>.txt
cd $GRAFT_REPO_ROOT
F=comfyui-3d-pack_amd/csrc/$1
for round in 1 2; do
  echo "[new]"; timeout 300 python bench.py --steps 20 --warmup 3 --lanes 1 --targets off --cpu-baseline off 2>/dev/null | python profiles/benchline.py
  cp $F /tmp/new.hip; cp profiles/_ab_head_${1%.hip}.txt $F
  echo "[head]"; timeout 300 python bench.py --steps 20 --warmup 3 --lanes 1 --targets off --cpu-baseline off 2>/dev/null | python profiles/benchline.py
  cp /tmp/new.hip $F
done
