#!/bin/bash
cd $GRAFT_REPO_ROOT
T=${1:-r04c2}
mkdir -p gpurun_out/$T
timeout 900 python -m pytest tests/test_gs_hip.py tests/test_zz_replay_gpu.py -m gpu -x -q > gpurun_out/$T/pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/$T/pytest.log
tail -4 gpurun_out/$T/pytest.log
timeout 300 python bench.py --mode train --steps 10 --warmup 3 --targets off --cpu-baseline off 2>/dev/null > gpurun_out/$T/train.json; python profiles/benchline.py < gpurun_out/$T/train.json
timeout 300 python bench.py --steps 20 --warmup 3 --targets off --cpu-baseline off 2>/dev/null > gpurun_out/$T/fwdbwd.json; python profiles/benchline.py < gpurun_out/$T/fwdbwd.json
