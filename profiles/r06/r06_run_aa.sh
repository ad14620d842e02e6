#!/bin/bash
# round 6, second session: what bounds the projection kernel?  store ablations (results are wrong with them: forward-only runs, kernel table only)
cd $GRAFT_REPO_ROOT
P=comfyui-3d-pack_amd; C=$P/csrc
use() { rm -rf $C; cp -r profiles/_ab/$1/csrc $C; cp profiles/_ab/$1/libc3d_hip.so profiles/_ab/$1/libc3d_hip.digest $P/lib/; export C3D_EXTRA_HIPCC_FLAGS="$(cat profiles/_ab/$1/flags)"; }
mkdir -p gpurun_out/r06aa
for i in 1 2; do for v in aa_base aa_norec aa_nosmall aa_none; do
  use $v
  timeout 200 python bench.py --mode fwd --views-per-gpu 8 --streams 1 --steps 20 --warmup 3 --targets off --cpu-baseline off 2>/dev/null | tail -1 > gpurun_out/r06aa/${v}_$i.json; echo "[$v]"; python profiles/benchline.py < gpurun_out/r06aa/${v}_$i.json | cut -c1-200
done; done
