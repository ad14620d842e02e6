#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06d
timeout 2400 python -m pytest tests/test_zz_baseline_1m.py tests/test_zz_rccl_world1.py tests/test_zz_ref_consumers.py tests/test_zz_replay_gpu.py -m gpu -q -s 2>&1 | grep -v "^$" | tail -60 > gpurun_out/r06d/pytest_gpu_tail.log; tail -40 gpurun_out/r06d/pytest_gpu_tail.log
timeout 600 python bench.py 2>gpurun_out/r06d/bench_default.err | tail -1 > gpurun_out/r06d/bench_default.json; python -c "
import json; d=json.load(open('gpurun_out/r06d/bench_default.json')); print(d['value'], d['ms_per_step'], json.dumps(d['targets'])[:2500])"
