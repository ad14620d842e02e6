#!/bin/bash
# round 5, onesweep front: bitop3 ranking + bounds-free full tiles ("new"), and the one-reorder-buffer form at 5 / 6 workgroups per CU, against the committed tree ("old")
cd $GRAFT_REPO_ROOT
T=r05q; mkdir -p gpurun_out/$T
P=comfyui-3d-pack_amd; C=$P/csrc
use() { rm -rf $C; cp -r profiles/_ab/$1/csrc $C; cp profiles/_ab/$1/libc3d_hip.so profiles/_ab/$1/libc3d_hip.digest $P/lib/; export C3D_EXTRA_HIPCC_FLAGS="$(cat profiles/_ab/$1/flags)"; }
for v in old new one5 one6; do use $v; echo "[$v] sort phases"; timeout 120 python profiles/microbench/sort_phases.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/$T/sort_phases_$v.txt; done
bash profiles/ab_run.sh $T "new one5 one6 old" 3 "sort_pairs or forward_matches or fused_multi_view or golden or sync_free"
