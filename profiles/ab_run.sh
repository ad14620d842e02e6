#!/bin/bash
# same-box A/B of prebuilt library variants (profiles/ab_prepare.sh): profiles/ab_run.sh <tag> "<variants>" [rounds] [pytest -k expression | -] [bench args]
cd $GRAFT_REPO_ROOT
T=$1; VARS=$2; R=${3:-3}; K=${4:--}; shift 4
P=comfyui-3d-pack_amd; C=$P/csrc
mkdir -p gpurun_out/$T
use() { rm -rf $C; cp -r profiles/_ab/$1/csrc $C; cp profiles/_ab/$1/libc3d_hip.so profiles/_ab/$1/libc3d_hip.digest $P/lib/; export C3D_EXTRA_HIPCC_FLAGS="$(cat profiles/_ab/$1/flags)"; }
if [ "$K" != "-" ]; then for v in $VARS; do use $v; echo "[$v] tests"; timeout 600 python -m pytest tests/test_gs_hip.py -m gpu -q -x -k "$K" 2>&1 | grep -v "^$" | tail -25; done; fi
for i in $(seq 1 $R); do for v in $VARS; do
  use $v
  timeout 300 python bench.py --steps 30 --warmup 3 --targets off --cpu-baseline off "$@" 2>/dev/null > gpurun_out/$T/${v}_$i.json; echo "[$v]"; python profiles/benchline.py < gpurun_out/$T/${v}_$i.json
done; done
