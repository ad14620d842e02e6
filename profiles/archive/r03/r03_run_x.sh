#!/bin/bash
# round 3, GPU call X: the remaining lines of the r03z collection on the final digest
R=$GRAFT_REPO_ROOT; TAG=r03z; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT; cd $R
timeout 300 python bench.py --mode train --loss full-torch --steps 5 --warmup 3 --cpu-baseline off --targets off < /dev/null 2>/dev/null | tail -1 > $OUT/${TAG}_bench_train_torchloss_n1.json
timeout 300 python bench.py --mode train --loss l1alpha --steps 20 --warmup 5 --cpu-baseline off --targets off < /dev/null 2>/dev/null | tail -1 > $OUT/${TAG}_bench_train_l1alpha_n1.json
timeout 300 python bench.py --render-path fused --steps 20 --warmup 5 --cpu-baseline off --targets off < /dev/null 2>/dev/null | tail -1 > $OUT/${TAG}_bench_renderer_api_n1.json
timeout 300 python bench.py --render-path boundary --steps 20 --warmup 5 --cpu-baseline off --targets off < /dev/null 2>/dev/null | tail -1 > $OUT/${TAG}_bench_boundary_api_n1.json
timeout 300 python bench.py --workload ref-default --ref-res 512 --steps 700 --warmup 50 --cpu-baseline off < /dev/null 2>/dev/null | tail -1 > $OUT/${TAG}_bench_ref_default_512.json
timeout 300 python bench.py --workload ref-default --ref-res 512 --steps 700 --warmup 50 --cpu-baseline off < /dev/null 2>/dev/null | tail -1 > $OUT/${TAG}_bench_ref_default_512.json
timeout 120 python profiles/microbench/sort_phases.py < /dev/null > $OUT/${TAG}_sort_phases.txt 2>&1
for f in $OUT/${TAG}_bench_*.json; do echo $(basename $f) $(timeout 20 python profiles/benchline.py < $f | cut -c1-40); done
