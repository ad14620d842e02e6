#!/bin/bash
# round 3, GPU call N: small sweeps on the final kernels (compositing tokens 2/3/4, mesh lanes)
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r03n; mkdir -p $OUT; cd $R
for v in 2 3 4; do C3D_COMP_TOKENS=$v timeout 200 python bench.py --steps 30 --warmup 5 --cpu-baseline off --targets off < /dev/null 2>/dev/null | tail -1 | head -c 215 | tail -c 70; echo " tokens=$v"; done
for l in 2 4 6 8; do timeout 200 python bench.py --workload mesh --lanes $l --steps 40 --warmup 5 --cpu-baseline off < /dev/null 2>/dev/null | tail -1 | head -c 225 | tail -c 70; echo " mesh lanes=$l"; done
