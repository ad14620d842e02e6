#!/bin/bash
# round 5: k_emit counts the digits of the tile keys it writes (workgroups stay; the tile sort skips its histogram kernel) = eh, against the committed tree = base
bash profiles/ab_run.sh r05e2 "eh" 0 "sort_pairs or forward_matches or backward_matches or fused_multi_view or golden or render_views or reproducible or unequal or halves or edge or overflow or capacity or sync_free or trainer_densify or internal_state or recorded_pair"
bash profiles/ab_run.sh r05e2 "base eh" 4 -
for i in 1 2; do for v in base eh; do
  P=comfyui-3d-pack_amd; C=$P/csrc; rm -rf $C; cp -r profiles/_ab/$v/csrc $C; cp profiles/_ab/$v/libc3d_hip.so profiles/_ab/$v/libc3d_hip.digest $P/lib/
  echo "[$v] fwd64"; timeout 300 python bench.py --mode fwd --views-per-gpu 64 --steps 10 --warmup 3 --cpu-baseline off --targets off 2>/dev/null | python profiles/benchline.py
  echo "[$v] boundary"; timeout 300 python bench.py --render-path boundary --steps 10 --warmup 3 --cpu-baseline off --targets off 2>/dev/null | python profiles/benchline.py
done; done
