#!/bin/bash
# round 3, GPU call S: cached reference cameras in the fused trainer step -- replay tests, the reference's default workload (twice)
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r03s; mkdir -p $OUT; cd $R
timeout 900 python -m pytest tests/test_zz_replay_gpu.py tests/test_gs_hip.py -m gpu -q -x -k "replay or trainer or node or reduce_the_loss" < /dev/null 2>&1 | tail -3
for i in 1 2; do timeout 300 python bench.py --workload ref-default --ref-res 512 --steps 700 --warmup 50 --cpu-baseline off < /dev/null 2>/dev/null | tail -1 > $OUT/bench_ref512_$i.json; python profiles/benchline.py < $OUT/bench_ref512_$i.json; done
timeout 300 python bench.py --workload ref-default --ref-res 1024 --steps 700 --warmup 50 --cpu-baseline off < /dev/null 2>/dev/null | tail -1 > $OUT/bench_ref1024.json; python profiles/benchline.py < $OUT/bench_ref1024.json
