#!/bin/bash
# mesh antialias with per-view silhouette bits: mesh tests, then same-box A/B of the mesh step
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04yb
timeout 1200 python -m pytest tests/test_mesh_hip.py tests/test_zmesh_ext.py tests/test_zz_ref_consumers.py tests/test_zz_replay_gpu.py tests/test_ref_pin.py -m gpu -x -q 2>&1 | tail -3
i=0
for F in "" "-DMESH_NO_SIL_BITS" "" "-DMESH_NO_SIL_BITS"; do
  export C3D_EXTRA_HIPCC_FLAGS="$F"
  timeout 600 python bench.py --workload mesh --steps 40 --warmup 5 --cpu-baseline off 2>/dev/null | tail -1 > gpurun_out/r04yb/mesh_$i.json
  echo "[$F]"; python profiles/benchline.py < gpurun_out/r04yb/mesh_$i.json
  i=$((i+1))
done
