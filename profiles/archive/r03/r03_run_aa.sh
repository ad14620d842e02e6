#!/bin/bash
# round 3, GPU call AA: compositing tokens re-checked under the final defaults (kernel arguments in device memory)
R=$GRAFT_REPO_ROOT; cd $R
for v in 2 0 2 0 3; do echo -n "tokens=$v default "; C3D_COMP_TOKENS=$v timeout 200 python bench.py --steps 40 --warmup 5 --cpu-baseline off --targets off < /dev/null 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"; done
for v in 2 0; do echo -n "tokens=$v train "; C3D_COMP_TOKENS=$v timeout 200 python bench.py --mode train --steps 30 --warmup 5 --cpu-baseline off --targets off < /dev/null 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"; done
