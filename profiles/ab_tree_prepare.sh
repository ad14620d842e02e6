#!/bin/bash
# A whole TREE of an earlier commit beside the working tree, library built HERE, for same-box A/Bs of changes that cross the C-ABI (bench.py, the Python boundary and the
# library must match: profiles/ab_prepare.sh swaps csrc + .so only):   profiles/ab_tree_prepare.sh <name> <git ref>
# -> profiles/_ab/trees/<name>/  (git-ignored; travels with gpurun).  profiles/ab_tree_run.sh runs bench.py of each tree in turn; `work` there = the working tree itself.
set -e
cd "$(dirname "$0")/.."
name=$1; ref=$2
D=profiles/_ab/trees/$name
rm -rf $D; mkdir -p $D
git archive $ref -- bench.py comfyui-3d-pack_amd include profiles/benchline.py | tar -x -C $D
python $D/comfyui-3d-pack_amd/c3d_hip/build.py > /dev/null
echo "$name: $(git rev-parse --short $ref)  digest $(cut -c1-16 $D/comfyui-3d-pack_amd/lib/libc3d_hip.digest)"
