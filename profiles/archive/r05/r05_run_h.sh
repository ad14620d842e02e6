#!/bin/bash
# lean onesweep (k_onesweep2: keys and values through one LDS buffer one after the other, 6-7 workgroups per CU) against the round-2 kernel, same box
cd $GRAFT_REPO_ROOT
T=r05j
mkdir -p gpurun_out/$T
timeout 600 python -m pytest tests/test_gs_hip.py tests/test_knn.py -m gpu -q -x -k "sort or scan or forward_matches or backward_matches or fused_multi_view or golden or recorded_pair or internal_state or edge_cases or knn or render_views" 2>&1 | tail -3
i=0
for F in "" "-DRS2_FOR=0" "-DRS2_FOR=2" "-DRS2_LOOKBACK=16" "-DRS2_FOR=0" "" "-DRS2_FOR=2"; do
  export C3D_EXTRA_HIPCC_FLAGS="$F"
  timeout 300 python bench.py --steps 30 --warmup 3 --targets off --cpu-baseline off 2>/dev/null > gpurun_out/$T/ab_$i.json
  echo "[$F]"; python profiles/benchline.py < gpurun_out/$T/ab_$i.json
  i=$((i+1))
done
unset C3D_EXTRA_HIPCC_FLAGS
timeout 300 python bench.py --mode fwd --views-per-gpu 64 --steps 10 --warmup 2 --cpu-baseline off --targets off 2>/dev/null | tail -1 > gpurun_out/$T/fwd64.json; python profiles/benchline.py < gpurun_out/$T/fwd64.json
