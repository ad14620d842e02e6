#!/bin/bash
# round 6, second GPU call: hint-sized launches + the count word in pinned memory ("verified" / "unverified") -- tests, then same-box A/Bs against round 5's tree
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06b
timeout 900 python -m pytest tests/test_gs_hip.py -m gpu -x -q 2>&1 | tail -15 | tee gpurun_out/r06b/pytest_gs_hip.log
echo "== boundary path, differentiated: r05 tree | work verified | work unverified | work off"
for i in 1 2 3; do
  bash profiles/ab_tree_run.sh r06b/boundary_r05_$i "r05" 1 --render-path boundary --steps 20 --warmup 5
  for m in verified unverified off; do
    timeout 300 python bench.py --cpu-baseline off --targets off --render-path boundary --sync-free $m --steps 20 --warmup 5 2>/dev/null | tail -1 > gpurun_out/r06b/boundary_${m}_$i.json
    echo "[work $m]"; python profiles/benchline.py < gpurun_out/r06b/boundary_${m}_$i.json
  done
done
echo "== inference caller (--mode fwd, torch.inference_mode()): verified + forward-only on/off | off (synchronous) | r05 tree"
for i in 1 2 3; do
  for v in "verified on" "verified off" "unverified on" "off on"; do set -- $v
    timeout 300 python bench.py --cpu-baseline off --targets off --render-path boundary --mode fwd --inference-mode on --sync-free $1 --forward-only $2 --steps 20 --warmup 5 2>/dev/null | tail -1 > gpurun_out/r06b/inference_${1}_fo${2}_$i.json
    echo "[$1, forward-only $2]"; python profiles/benchline.py < gpurun_out/r06b/inference_${1}_fo${2}_$i.json
  done
  bash profiles/ab_tree_run.sh r06b/inference_r05_$i "r05" 1 --render-path boundary --mode fwd --steps 20 --warmup 5
done
