#!/bin/bash
# round 6, second session: FusedViewRender(streams=S) -- the views of a forward-only call in S parts on HIP streams of their own
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06t
timeout 900 python -m pytest tests/test_gs_hip.py -m gpu -x -q -k "several_streams or render_views_equals or recorded_pair_activity or sort" 2>&1 | tail -4
for s in 1 4 1 4; do
  python bench.py --mode fwd --views-per-gpu 64 --steps 10 --warmup 3 --cpu-baseline off --streams $s 2>/dev/null | tail -1 > gpurun_out/r06t/fwd64_streams$s.json; echo "[streams $s]"; python profiles/benchline.py < gpurun_out/r06t/fwd64_streams$s.json | cut -c1-300
done
echo "--- medium scene: 200k Gaussians (scale 0.01), 1024x1024, 32 views"
timeout 600 python profiles/microbench/two_streams_fwd64.py 16 200000 1024 1024 32 -4.6 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r06t/medium_200k_1024_32.txt
echo "--- small scene: 10k Gaussians (scale 0.03), 512x512, 32 views"
timeout 600 python profiles/microbench/two_streams_fwd64.py 16 10000 512 512 32 -3.5 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r06t/small_10k_512_32.txt
echo "--- BASELINE config 2"
timeout 600 python profiles/microbench/two_streams_fwd64.py 16 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r06t/config2_1m_1080p_64.txt
