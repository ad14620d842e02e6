#!/bin/bash
# round 6, second session: camera matrices / positions read through the constant address space (scalar loads: no s_waitcnt vmcnt(0) behind the previous view's stores)
# (profiles/ab_prepare.sh ab_old work "-DC3D_NO_CONST_LOADS"; ab_new work)
cd $GRAFT_REPO_ROOT
P=comfyui-3d-pack_amd; C=$P/csrc
use() { rm -rf $C; cp -r profiles/_ab/$1/csrc $C; cp profiles/_ab/$1/libc3d_hip.so profiles/_ab/$1/libc3d_hip.digest $P/lib/; export C3D_EXTRA_HIPCC_FLAGS="$(cat profiles/_ab/$1/flags)"; }
use ab_new
timeout 1500 python -m pytest tests/test_gs_hip.py tests/test_zz_replay_gpu.py -m gpu -x -q 2>&1 | tail -3
bash profiles/ab_run.sh r06ab/step "ab_old ab_new" 3 - | cut -c1-400
bash profiles/ab_run.sh r06ab/fwd64 "ab_old ab_new" 2 - --mode fwd --views-per-gpu 64 --steps 10 | cut -c1-300
bash profiles/ab_run.sh r06ab/fwd64s1 "ab_old ab_new" 2 - --mode fwd --views-per-gpu 64 --streams 1 --steps 10 | cut -c1-300
bash profiles/ab_run.sh r06ab/boundary "ab_old ab_new" 2 - --render-path boundary --steps 20 | cut -c1-400
bash profiles/ab_run.sh r06ab/refdefault "ab_old ab_new" 2 - --workload ref-default --ref-res 512 --steps 600 --warmup 50 --timed-prof off | cut -c1-100
