#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04dbg
for i in 1 2; do timeout 300 python -m pytest tests/test_gs_hip.py -m gpu -x -q -k "medium" 2>&1 | tail -3; done
export C3D_EXTRA_HIPCC_FLAGS="-DGS_BWD_REDUCE_DPP=1"
timeout 600 python -m pytest tests/test_gs_hip.py -m gpu -x -q -k "medium" 2>&1 | tail -3
timeout 300 python bench.py --steps 20 --warmup 3 --lanes 1 --targets off --cpu-baseline off 2>/dev/null | python profiles/benchline.py
