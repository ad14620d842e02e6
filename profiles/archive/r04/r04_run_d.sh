#!/bin/bash
cd $GRAFT_REPO_ROOT
T=${1:-r04h}
mkdir -p gpurun_out/$T
timeout 900 python -m pytest tests/test_zz_rccl_world1.py -m gpu -x -q -s > gpurun_out/$T/pytest_rccl.log 2>&1; echo "pytest rc $?" >> gpurun_out/$T/pytest_rccl.log
tail -25 gpurun_out/$T/pytest_rccl.log
timeout 400 python bench.py --steps 10 --warmup 3 2>gpurun_out/$T/bench.err > gpurun_out/$T/bench.json; python profiles/benchline.py < gpurun_out/$T/bench.json; tail -3 gpurun_out/$T/bench.err
timeout 400 python bench.py --workload mesh --steps 20 --warmup 3 --cpu-baseline off 2>/dev/null > gpurun_out/$T/mesh.json; python profiles/benchline.py < gpurun_out/$T/mesh.json
