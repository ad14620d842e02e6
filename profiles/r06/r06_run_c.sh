#!/bin/bash
# round 6, third GPU call: the whole GPU suite on the reworked drop-in path + multi-rank agreement + bench line with configs 3 / 5; boundary A/B after the micro-fixes
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06c
timeout 1500 python -m pytest tests -m gpu -x -q -s 2>&1 | grep -v "^$" | tail -40 > gpurun_out/r06c/pytest_gpu.log; tail -25 gpurun_out/r06c/pytest_gpu.log
timeout 600 python bench.py 2>gpurun_out/r06c/bench_default.err | tail -1 > gpurun_out/r06c/bench_default.json; python -c "
import json; d=json.load(open('gpurun_out/r06c/bench_default.json')); print(d['value'], d['ms_per_step'], json.dumps(d['targets'])[:1500])"
echo "== boundary path, differentiated: r05 tree | work verified | work unverified"
for i in 1 2 3; do
  bash profiles/ab_tree_run.sh r06c/boundary_r05_$i "r05" 1 --render-path boundary --steps 20 --warmup 5
  for m in verified unverified; do
    timeout 300 python bench.py --cpu-baseline off --targets off --render-path boundary --sync-free $m --steps 20 --warmup 5 2>/dev/null | tail -1 > gpurun_out/r06c/boundary_${m}_$i.json
    echo "[work $m]"; python profiles/benchline.py < gpurun_out/r06c/boundary_${m}_$i.json
  done
done
