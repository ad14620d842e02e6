#!/bin/bash
# round 3, GPU call V: does the in-process default of HIP_FORCE_DEV_KERNARG take effect? (unset vs explicit 0 vs explicit 1)
R=$GRAFT_REPO_ROOT; cd $R
for v in unset 0 1 unset 0; do
if [ $v = unset ]; then unset HIP_FORCE_DEV_KERNARG; else export HIP_FORCE_DEV_KERNARG=$v; fi
echo -n "kernarg=$v ref-default "; timeout 300 python bench.py --workload ref-default --ref-res 512 --steps 700 --warmup 50 --cpu-baseline off < /dev/null 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"
echo -n "kernarg=$v default "; timeout 300 python bench.py --steps 30 --warmup 5 --cpu-baseline off --targets off < /dev/null 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"
done
