#!/bin/bash
# A/B of compile-time variants: bash profiles/r04_ab.sh TAG "flags A" "flags B" ...   ("" = default build); prints the fwd+bwd line's per-stage times
cd $GRAFT_REPO_ROOT
T=$1; shift
mkdir -p gpurun_out/$T
i=0
for F in "$@"; do
  export C3D_EXTRA_HIPCC_FLAGS="$F"
  if [ $i -eq 0 ]; then timeout 600 python -m pytest tests/test_gs_hip.py -m gpu -x -q -k "scan or sort or internal_state or forward_matches or backward_matches or fused_multi_view or render_views or edge_cases or golden or bit_reproducible or medium" 2>&1 | tail -2; fi
  timeout 300 python bench.py --steps 20 --warmup 3 --lanes 1 --targets off --cpu-baseline off 2>/dev/null > gpurun_out/$T/fwdbwd_$i.json
  echo "[$F]"; python profiles/benchline.py < gpurun_out/$T/fwdbwd_$i.json
  i=$((i+1))
done
