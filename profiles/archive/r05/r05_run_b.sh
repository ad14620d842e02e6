#!/bin/bash
# round 5, second call: new tests; drop-in path sync-free on / off on one box (host enqueue time beside the wall time); A8 pipeline variants; host profile of the node-default loop
cd $GRAFT_REPO_ROOT
T=r05b
mkdir -p gpurun_out/$T
timeout 900 python -m pytest tests/test_gs_hip.py tests/test_mesh_hip.py tests/test_zz_ref_consumers.py -m gpu -q -k "sync_free or halves or clipped or end_to_end or replayed" 2>&1 | tail -5
for SF in on off on off; do
  for P in boundary fused; do
    timeout 300 python bench.py --render-path $P --sync-free $SF --steps 20 --warmup 5 --cpu-baseline off --targets off 2>/dev/null | tail -1 > gpurun_out/$T/bench_${P}_$SF.json
    echo "[$P sync-free $SF]"; python - <<PY
import json
d=json.load(open("gpurun_out/$T/bench_${P}_$SF.json"))
print(d["value"], d["ms_per_step"], "host enqueue ms/step", d["config"]["host_enqueue_ms_per_step"])
PY
  done
done
i=0
for F in "" "-DGS_A8_PIPE" "-DGS_A8_PIPE -DGS_A8_MINB=3" "-DGS_A8_MINB=3" ""; do
  export C3D_EXTRA_HIPCC_FLAGS="$F"
  if [ $i -eq 1 ]; then timeout 600 python -m pytest tests/test_gs_hip.py -m gpu -x -q -k "fused_multi_view or param_backward_by or trainer_fused or halves" 2>&1 | tail -2; fi
  timeout 300 python bench.py --steps 20 --warmup 3 --targets off --cpu-baseline off 2>/dev/null > gpurun_out/$T/a8_$i.json
  echo "[$F]"; python profiles/benchline.py < gpurun_out/$T/a8_$i.json
  i=$((i+1))
done
unset C3D_EXTRA_HIPCC_FLAGS
timeout 300 python bench.py --workload ref-default --ref-res 512 --steps 700 --warmup 50 --cpu-baseline off 2>/dev/null | tail -1 > gpurun_out/$T/ref_default.json
python - <<PY
import json
d=json.load(open("gpurun_out/$T/ref_default.json")); print(d["value"], d["ms_per_step"], d["config"]["host_enqueue_ms_per_step"], d["config"]["kernel_ms_per_step"])
PY
timeout 300 python -m cProfile -o /tmp/ref.prof bench.py --workload ref-default --ref-res 512 --steps 400 --warmup 50 --cpu-baseline off --timed-prof off > /dev/null 2>&1
python - <<PY > gpurun_out/$T/ref_default_host_profile.txt
import pstats
p = pstats.Stats("/tmp/ref.prof"); p.sort_stats("tottime").print_stats(45)
p.sort_stats("cumulative").print_stats(40)
PY
head -75 gpurun_out/$T/ref_default_host_profile.txt | cut -c1-170
