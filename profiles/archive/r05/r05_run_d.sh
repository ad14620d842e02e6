#!/bin/bash
# round 5, call d: same-box A/B of the record-base scan inside the compositing launch vs as a launch of its own; the fixed sync-free test
cd $GRAFT_REPO_ROOT
T=r05d
mkdir -p gpurun_out/$T
timeout 600 python -m pytest tests/test_gs_hip.py -m gpu -q -k "sync_free or halves" 2>&1 | tail -15
i=0
for F in "" "-DGS_SCAN_A_SEPARATE" "" "-DGS_SCAN_A_SEPARATE"; do
  export C3D_EXTRA_HIPCC_FLAGS="$F"
  timeout 300 python bench.py --steps 30 --warmup 3 --targets off --cpu-baseline off 2>/dev/null > gpurun_out/$T/ab_$i.json
  echo "[$F]"; python profiles/benchline.py < gpurun_out/$T/ab_$i.json
  i=$((i+1))
done
