#!/bin/bash
# same-box A/B of the csrc tree: working tree vs profiles/_ab_csrc_head (the committed HEAD); args: tag, pytest -k expression
cd $GRAFT_REPO_ROOT
T=${1:-r05g}
mkdir -p gpurun_out/$T
C=comfyui-3d-pack_amd/csrc
cp -r $C /tmp/csrc_new
timeout 600 python -m pytest tests/test_gs_hip.py -m gpu -q -x -k "${2:-sort or forward_matches or backward_matches or fused_multi_view or golden or recorded_pair or internal_state or edge_cases}" 2>&1 | tail -3
for i in 0 1 2; do
  rm -rf $C; cp -r /tmp/csrc_new $C
  timeout 300 python bench.py --steps 30 --warmup 3 --targets off --cpu-baseline off 2>/dev/null > gpurun_out/$T/new_$i.json; echo "[new]"; python profiles/benchline.py < gpurun_out/$T/new_$i.json
  rm -rf $C; mkdir -p $C; cp profiles/_ab_csrc_head/* $C/
  timeout 300 python bench.py --steps 30 --warmup 3 --targets off --cpu-baseline off 2>/dev/null > gpurun_out/$T/old_$i.json; echo "[old]"; python profiles/benchline.py < gpurun_out/$T/old_$i.json
done
rm -rf $C; cp -r /tmp/csrc_new $C
