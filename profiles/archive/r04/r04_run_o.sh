#!/bin/bash
# MS-SSIM with NT tiles per workgroup and next-tile prefetch: tests, then same-box A/B over NT x residency (train step's msssim group)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04o
timeout 900 python -m pytest tests -m gpu -x -q -k "msssim or ssim or fused_multi_view or replay or mesh_step or diffmesh" 2>&1 | tail -3
i=0
for F in "" "-DMS_TILES_PER_GROUP=1" "-DMS_TILES_PER_GROUP=8" "-DMS_MIN_BLOCKS=5" "-DMS_TILES_PER_GROUP=2" ""; do
  export C3D_EXTRA_HIPCC_FLAGS="$F"
  timeout 300 python bench.py --mode train --steps 10 --warmup 3 --targets off --cpu-baseline off 2>/dev/null | tail -1 > gpurun_out/r04o/train_$i.json
  echo "[$F]"; python profiles/benchline.py < gpurun_out/r04o/train_$i.json
  i=$((i+1))
done
