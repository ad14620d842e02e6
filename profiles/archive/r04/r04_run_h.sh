#!/bin/bash
# forward-only without the record-base scan; full-line record writes A/B
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04h2
for F in "" "-DGS_REC_PAD=1" ""  "-DGS_REC_PAD=1"; do
  export C3D_EXTRA_HIPCC_FLAGS="$F"
  echo "[$F]"
  timeout 300 python bench.py --steps 20 --warmup 3 --lanes 1 --targets off --cpu-baseline off 2>/dev/null | python profiles/benchline.py
  timeout 300 python bench.py --mode fwd --views-per-gpu 64 --group 8 --steps 5 --warmup 2 --targets off --cpu-baseline off 2>/dev/null | python profiles/benchline.py
done
