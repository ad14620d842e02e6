#!/bin/bash
# MS-SSIM with the pooling inside the forward kernel: tests, training-step lines, the node-default line
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04zp
timeout 900 python -m pytest tests -m gpu -x -q -k "msssim or ssim or fused_multi_view or replay or mesh_step or diffmesh or trainer" 2>&1 | tail -3
for i in 0 1; do
  timeout 300 python bench.py --mode train --steps 20 --warmup 5 --targets off --cpu-baseline off 2>/dev/null | tail -1 > gpurun_out/r04zp/train_$i.json
  python profiles/benchline.py < gpurun_out/r04zp/train_$i.json
done
timeout 300 python bench.py --workload ref-default --ref-res 512 --steps 700 --warmup 50 --cpu-baseline off 2>/dev/null | tail -1 > gpurun_out/r04zp/ref_default.json
python profiles/benchline.py < gpurun_out/r04zp/ref_default.json
