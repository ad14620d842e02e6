#!/bin/bash
# round 3, GPU call P: multi-view projection with the SH coefficients in registers (C3D_PRE_REGS) -- tests, A/B
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r03p; mkdir -p $OUT; cd $R
timeout 600 python -m pytest tests/test_gs_hip.py -m gpu -q -x -k "views or fused or step" < /dev/null 2>&1 | tail -2
for v in 1 0 1 0; do
C3D_PRE_REGS=$v timeout 200 python bench.py --steps 30 --warmup 5 --cpu-baseline off --targets off < /dev/null 2>/dev/null | tail -1 > $OUT/b.json; python - <<PY
import json; d=json.load(open("$OUT/b.json")); print("default regs=$v", d["ms_per_step"], d["kernels"]["gs_preprocess"]["avg_ms"])
PY
C3D_PRE_REGS=$v timeout 200 python bench.py --mode fwd --views-per-gpu 64 --steps 10 --warmup 3 --cpu-baseline off --targets off < /dev/null 2>/dev/null | tail -1 > $OUT/b.json; python - <<PY
import json; d=json.load(open("$OUT/b.json")); print("fwd64 regs=$v", d["ms_per_step"], d["kernels"]["gs_preprocess"])
PY
done
