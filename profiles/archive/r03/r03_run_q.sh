#!/bin/bash
# round 3, GPU call Q: one-launch Adam -- tests, the reference's default workload, the training line
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r03q; mkdir -p $OUT; cd $R
timeout 900 python -m pytest tests/test_gs_hip.py tests/test_zz_replay_gpu.py tests/test_mesh_hip.py -m gpu -q -x -k "adam or trainer or replay or train or node" < /dev/null 2>&1 | tail -3
timeout 300 python bench.py --workload ref-default --ref-res 512 --steps 700 --warmup 50 --cpu-baseline off < /dev/null 2>/dev/null | tail -1 > $OUT/bench_ref512.json; head -c 200 $OUT/bench_ref512.json | tail -c 100; echo
timeout 300 python bench.py --mode train --steps 20 --warmup 5 --cpu-baseline off --targets off < /dev/null 2>/dev/null | tail -1 > $OUT/bench_train.json; head -c 230 $OUT/bench_train.json | tail -c 100; echo
