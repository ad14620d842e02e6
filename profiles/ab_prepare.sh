#!/bin/bash
# Build one variant of the library HERE (hipcc cross-compiles) for a same-box A/B on the GPU box without rebuilds there:
#   profiles/ab_prepare.sh <name> <head|work|path of a csrc tree> ["extra hipcc flags"]
# -> profiles/_ab/<name>/{csrc/, libc3d_hip.so, libc3d_hip.digest, flags}   (git-ignored; travels with gpurun).  `head` = the csrc tree of the last commit.
set -e
cd "$(dirname "$0")/.."
name=$1; which=$2; flags=$3
P=comfyui-3d-pack_amd; C=$P/csrc
rm -rf profiles/_ab/$name; mkdir -p profiles/_ab/$name
rm -rf /tmp/_ab_keep; cp -r $C /tmp/_ab_keep
restore() { rm -rf $C; cp -r /tmp/_ab_keep $C; }
trap restore EXIT
if [ -d "$which" ]; then rm -rf /tmp/_ab_src_x; cp -r "$which" /tmp/_ab_src_x; rm -rf $C; cp -r /tmp/_ab_src_x $C; elif [ "$which" = head ]; then rm -rf /tmp/_ab_head_x; mkdir -p /tmp/_ab_head_x; git archive HEAD $C | tar -x -C /tmp/_ab_head_x; rm -rf $C; cp -r /tmp/_ab_head_x/$C $C; fi
C3D_EXTRA_HIPCC_FLAGS="$flags" python $P/c3d_hip/build.py --force > /dev/null
cp -r $C profiles/_ab/$name/csrc
cp $P/lib/libc3d_hip.so $P/lib/libc3d_hip.digest profiles/_ab/$name/
echo "$flags" > profiles/_ab/$name/flags
echo "$name: $(cut -c1-16 profiles/_ab/$name/libc3d_hip.digest)  flags='$flags'"
