#!/bin/bash
# round 5: same box -- bitop3 only (new), + view = XCD without stealing (vx), + stealing and the scan (pin)
cd $GRAFT_REPO_ROOT
P=comfyui-3d-pack_amd; C=$P/csrc
rm -rf $C; cp -r profiles/_ab/pin/csrc $C; cp profiles/_ab/pin/libc3d_hip.so profiles/_ab/pin/libc3d_hip.digest $P/lib/
timeout 600 python -m pytest tests/test_gs_hip.py -m gpu -q -x -k "sort_pairs or scan or fused_multi_view or golden or unequal or render_views or reproducible or halves" 2>&1 | grep -v "^$" | tail -40
bash profiles/ab_run.sh r05t "new vx pin" 3 -
