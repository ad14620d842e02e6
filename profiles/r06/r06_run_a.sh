#!/bin/bash
# round 6, first GPU call: the exact sync-free drop-in forward (two attempts on the device) and the forward-only flag -- tests, then same-box A/Bs against round 5's tree
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06a
timeout 900 python -m pytest tests/test_gs_hip.py -m gpu -x -q 2>&1 | tail -15 | tee gpurun_out/r06a/pytest_gs_hip.log
echo "== boundary path (the reference's loop over the drop-in API, differentiated): round 5's tree against the working tree"
bash profiles/ab_tree_run.sh r06a/boundary "r05 work" 3 --render-path boundary --steps 20 --warmup 5
echo "== inference caller (--mode fwd under torch.inference_mode()): forward-only flag on / off, working tree; round 5's tree (sync-free, 3 x capacity) for reference"
for i in 1 2 3; do for fo in on off; do
  timeout 300 python bench.py --cpu-baseline off --targets off --render-path boundary --mode fwd --inference-mode on --forward-only $fo --steps 20 --warmup 5 2>/dev/null | tail -1 > gpurun_out/r06a/inference_fo_${fo}_$i.json
  echo "[forward-only $fo]"; python profiles/benchline.py < gpurun_out/r06a/inference_fo_${fo}_$i.json
done; done
bash profiles/ab_tree_run.sh r06a/inference_r05 "r05" 2 --render-path boundary --mode fwd --steps 20 --warmup 5
