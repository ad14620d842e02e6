#!/bin/bash
# round 3, GPU call W: forward-only workspace slices -- render tests, then the final collection
R=$GRAFT_REPO_ROOT; cd $R
timeout 600 python -m pytest tests/test_gs_hip.py -m gpu -q -x -k "render_views or views or orbit or node" < /dev/null 2>&1 | tail -2
bash profiles/collect.sh r03z quick > /dev/null 2>&1
for f in gpurun_out/r03z/r03z_bench_*.json; do echo $(basename $f) $(timeout 20 python profiles/benchline.py < $f | cut -c1-40); done
