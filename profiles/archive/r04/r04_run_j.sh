#!/bin/bash
# mesh bench under compile-flag variants: bash profiles/r04_run_j.sh "flags A" "flags B" ...
cd $GRAFT_REPO_ROOT
for F in "$@"; do
  export C3D_EXTRA_HIPCC_FLAGS="$F"
  echo "[$F]"
  timeout 600 python bench.py --workload mesh --steps 20 --warmup 3 --cpu-baseline off 2>/dev/null | tail -1 | python profiles/benchline.py
done
