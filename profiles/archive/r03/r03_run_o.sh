#!/bin/bash
# round 3, GPU call O: onesweep tile size (C3D_SORT_ITEMS = 16 | 8): sort tests with both, default and forward-only lines
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r03o; mkdir -p $OUT; cd $R
for it in 32 16; do
C3D_SORT_ITEMS=$it timeout 300 python -m pytest tests/test_gs_hip.py -m gpu -q -x -k "sort or scan or forward_matches or internal_state or golden" < /dev/null 2>&1 | tail -2
done
for it in 16 32 16 32; do
C3D_SORT_ITEMS=$it timeout 200 python bench.py --steps 30 --warmup 5 --cpu-baseline off --targets off < /dev/null 2>/dev/null | tail -1 | head -c 215 | tail -c 70; echo " default items=$it"
C3D_SORT_ITEMS=$it timeout 200 python bench.py --mode fwd --views-per-gpu 64 --steps 10 --warmup 3 --cpu-baseline off --targets off < /dev/null 2>/dev/null | tail -1 | head -c 205 | tail -c 70; echo " fwd64 items=$it"
done
