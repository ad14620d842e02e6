#!/bin/bash
# round 6, second session: k_ras_tri's setup and edge functions in 32-bit integers for triangles within 1024 px (profiles/ab_prepare.sh w_gen work "-DC3D_MESH_NO_NARROW"; w_nar work)
cd $GRAFT_REPO_ROOT
P=comfyui-3d-pack_amd; C=$P/csrc
use() { rm -rf $C; cp -r profiles/_ab/$1/csrc $C; cp profiles/_ab/$1/libc3d_hip.so profiles/_ab/$1/libc3d_hip.digest $P/lib/; export C3D_EXTRA_HIPCC_FLAGS="$(cat profiles/_ab/$1/flags)"; }
mkdir -p gpurun_out/r06w
use w_nar
timeout 1200 python -m pytest tests/test_mesh_hip.py tests/test_zmesh_ext.py tests/test_ref_pin.py tests/test_zz_ref_consumers.py -m gpu -x -q 2>&1 | tail -4
for i in 1 2 3; do for v in w_gen w_nar; do
  use $v
  timeout 300 python bench.py --workload mesh --steps 40 --warmup 5 --cpu-baseline off 2>/dev/null | tail -1 > gpurun_out/r06w/${v}_$i.json; echo "[$v]"; python profiles/benchline.py < gpurun_out/r06w/${v}_$i.json | cut -c1-330
done; done
for v in w_gen w_nar; do use $v; timeout 300 python bench.py --workload mesh --render-path fused --steps 20 --warmup 5 --cpu-baseline off 2>/dev/null | tail -1 > gpurun_out/r06w/${v}_view_api.json; echo "[$v view api]"; python profiles/benchline.py < gpurun_out/r06w/${v}_view_api.json | cut -c1-200; done
