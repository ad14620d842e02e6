#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06m
timeout 1500 python -m pytest tests/test_gs_hip.py tests/test_zz_ref_consumers.py -m gpu -q -x 2>&1 | grep -v "^$" | tail -8 > gpurun_out/r06m/pytest.log; tail -8 gpurun_out/r06m/pytest.log
echo "== GaussianSplattingRenderer.render per view (--render-path fused) and the bare drop-in call (boundary), differentiated: verified | unverified | off ; r05 tree"
for i in 1 2; do
  for rp in fused boundary; do
    for m in verified unverified off; do
      timeout 300 python bench.py --cpu-baseline off --targets off --render-path $rp --sync-free $m --steps 20 --warmup 5 2>/dev/null | tail -1 > gpurun_out/r06m/${rp}_${m}_$i.json
      echo "[work $rp $m]"; python profiles/benchline.py < gpurun_out/r06m/${rp}_${m}_$i.json | cut -c1-40
    done
    bash profiles/ab_tree_run.sh r06m/${rp}_r05_$i "r05" 1 --render-path $rp --steps 20 --warmup 5 | cut -c1-40
  done
done
