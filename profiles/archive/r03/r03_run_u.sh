#!/bin/bash
# round 3, GPU call U: HIP_FORCE_DEV_KERNARG (kernel arguments in device memory) on the three lines
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r03u; mkdir -p $OUT; cd $R
for v in 0 1 0 1; do
echo -n "kernarg=$v ref-default "; HIP_FORCE_DEV_KERNARG=$v timeout 300 python bench.py --workload ref-default --ref-res 512 --steps 700 --warmup 50 --cpu-baseline off < /dev/null 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"
echo -n "kernarg=$v default "; HIP_FORCE_DEV_KERNARG=$v timeout 300 python bench.py --steps 30 --warmup 5 --cpu-baseline off --targets off < /dev/null 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"
echo -n "kernarg=$v mesh "; HIP_FORCE_DEV_KERNARG=$v timeout 300 python bench.py --workload mesh --steps 40 --warmup 5 --cpu-baseline off < /dev/null 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"
done
