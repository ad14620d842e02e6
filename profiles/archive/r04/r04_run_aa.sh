#!/bin/bash
# A8 without record gathers (conic recomputed, opacity read once): tests incl. the 1M parity, then the default line twice
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04aa
timeout 900 python -m pytest tests/test_gs_hip.py tests/test_zz_baseline_1m.py tests/test_zz_replay_gpu.py -m gpu -x -q -k "fused or step or baseline or replay or sh_storage or trainer or ranges" 2>&1 | tail -3
for i in 0 1; do
  timeout 300 python bench.py --steps 20 --warmup 3 --targets off --cpu-baseline off 2>/dev/null > gpurun_out/r04aa/fwdbwd_$i.json
  python profiles/benchline.py < gpurun_out/r04aa/fwdbwd_$i.json
done
