#!/bin/bash
# persistent / pipelined onesweep: sort + binning tests, then same-box A/B (default = pipelined at 3 workgroups per CU; pipelined at 4 (spills); one tile per workgroup)
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04u
timeout 600 python -m pytest tests/test_gs_hip.py -m gpu -x -q -k "scan or sort or internal_state or forward_matches or backward_matches or fused_multi_view or render_views or edge_cases or golden or bit_reproducible or medium" 2>&1 | tail -2
i=0
for F in "" "-DRS_PIPELINE=0 -DRS_MIN_BLOCKS=4" "-DRS_MIN_BLOCKS=4" "" "-DRS_PIPELINE=0 -DRS_MIN_BLOCKS=4"; do
  export C3D_EXTRA_HIPCC_FLAGS="$F"
  timeout 300 python bench.py --steps 20 --warmup 3 --lanes 1 --targets off --cpu-baseline off 2>/dev/null > gpurun_out/r04u/fwdbwd_$i.json
  echo "[$F]"; python profiles/benchline.py < gpurun_out/r04u/fwdbwd_$i.json
  timeout 300 python bench.py --mode fwd --views-per-gpu 64 --steps 10 --warmup 2 --targets off --cpu-baseline off 2>/dev/null > gpurun_out/r04u/fwd64_$i.json
  python profiles/benchline.py < gpurun_out/r04u/fwd64_$i.json
  i=$((i+1))
done
