#!/bin/bash
cd $GRAFT_REPO_ROOT
T=${1:-r04k}
mkdir -p gpurun_out/$T
timeout 900 python -m pytest tests/test_gs_hip.py -m gpu -x -q -k "fused_multi or render_views or halves or deferred or ranges or sh_storage or trainer_fused" > gpurun_out/$T/pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/$T/pytest.log
tail -4 gpurun_out/$T/pytest.log
for L in 1 2 4; do timeout 300 python bench.py --steps 20 --warmup 3 --lanes $L --targets off --cpu-baseline off 2>/dev/null | python profiles/benchline.py; done
for cfg in "1 8" "2 8" "2 4" "2 16" "2 2"; do set -- $cfg; timeout 300 python bench.py --mode fwd --views-per-gpu 64 --steps 5 --warmup 2 --lanes $1 --group $2 --cpu-baseline off 2>/dev/null | python profiles/benchline.py; done
for L in 1 2; do timeout 300 python bench.py --mode train --steps 10 --warmup 3 --lanes $L --targets off --cpu-baseline off 2>/dev/null | python profiles/benchline.py; done
