#!/bin/bash
# round 5, third call: record-base scan inside the recording forward compositing launch -- parity tests, bench lines (default / train / boundary), the failing sync-free test with its traceback
cd $GRAFT_REPO_ROOT
T=r05c
mkdir -p gpurun_out/$T
timeout 600 python -m pytest tests/test_gs_hip.py -m gpu -q -x -k "sync_free" 2>&1 | tail -40
timeout 900 python -m pytest tests/test_gs_hip.py tests/test_mesh_hip.py -m gpu -q -k "not baseline_size and not full_size and not longer_run" 2>&1 | tail -6
for i in 0 1; do
  timeout 300 python bench.py --steps 20 --warmup 3 --targets off --cpu-baseline off 2>/dev/null > gpurun_out/$T/fwdbwd_$i.json; python profiles/benchline.py < gpurun_out/$T/fwdbwd_$i.json
done
timeout 300 python bench.py --mode train --steps 20 --warmup 3 --targets off --cpu-baseline off 2>/dev/null > gpurun_out/$T/train.json; python profiles/benchline.py < gpurun_out/$T/train.json
timeout 300 python bench.py --render-path boundary --steps 20 --warmup 5 --cpu-baseline off --targets off 2>/dev/null | tail -1 > gpurun_out/$T/boundary.json; python profiles/benchline.py < gpurun_out/$T/boundary.json
