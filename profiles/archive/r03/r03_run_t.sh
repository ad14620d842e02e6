#!/bin/bash
# round 3, GPU call T: view-lane sweeps on the final code (one box)
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r03t; mkdir -p $OUT; cd $R
{
for l in 1 2 4 6 8; do echo -n "fwdbwd lanes=$l "; timeout 200 python bench.py --lanes $l --steps 20 --warmup 5 --cpu-baseline off --targets off < /dev/null 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"; done
for l in 2 4 8; do echo -n "fwd64 lanes=$l "; timeout 200 python bench.py --mode fwd --views-per-gpu 64 --lanes $l --steps 5 --warmup 2 --cpu-baseline off --targets off < /dev/null 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"; done
for l in 2 4 8; do echo -n "train lanes=$l "; timeout 200 python bench.py --mode train --lanes $l --steps 20 --warmup 5 --cpu-baseline off --targets off < /dev/null 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"; done
for l in 1 2 4 8; do echo -n "mesh lanes=$l "; timeout 200 python bench.py --workload mesh --lanes $l --steps 40 --warmup 5 --cpu-baseline off < /dev/null 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"; done
} > $OUT/r03z_lane_sweeps.txt 2>&1
cat $OUT/r03z_lane_sweeps.txt
