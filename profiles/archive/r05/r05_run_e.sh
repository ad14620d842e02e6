#!/bin/bash
# round 5, call e: same-box A/B of the whole csrc tree: working tree (record-base scan inside the compositing launch) vs commit b1503db (scan as its own 256-thread launch)
cd $GRAFT_REPO_ROOT
T=r05f
mkdir -p gpurun_out/$T
C=comfyui-3d-pack_amd/csrc
cp -r $C /tmp/csrc_new
timeout 600 python -m pytest tests/test_gs_hip.py -m gpu -q -x -k "backward_matches or fused_multi_view or halves or medium or golden or bit_reproducible or edge_cases" 2>&1 | tail -3
for i in 0 1 2; do
  rm -rf $C; cp -r /tmp/csrc_new $C
  timeout 300 python bench.py --steps 30 --warmup 3 --targets off --cpu-baseline off 2>/dev/null > gpurun_out/$T/new_$i.json; echo "[new]"; python profiles/benchline.py < gpurun_out/$T/new_$i.json
  rm -rf $C; mkdir -p $C; cp profiles/_ab_csrc_b1503db/* $C/
  timeout 300 python bench.py --steps 30 --warmup 3 --targets off --cpu-baseline off 2>/dev/null > gpurun_out/$T/old_$i.json; echo "[old]"; python profiles/benchline.py < gpurun_out/$T/old_$i.json
done
rm -rf $C; cp -r /tmp/csrc_new $C
