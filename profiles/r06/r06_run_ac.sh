#!/bin/bash
# round 6, second session: the projection kernel's record stores through a wave-private LDS tile (four lanes per 64-byte line instead of 64 requests of 16 bytes per instruction)
# (profiles/ab_prepare.sh ac_dir work "-DC3D_PRE_DIRECT_STORES"; ac_lds work)
cd $GRAFT_REPO_ROOT
P=comfyui-3d-pack_amd; C=$P/csrc
use() { rm -rf $C; cp -r profiles/_ab/$1/csrc $C; cp profiles/_ab/$1/libc3d_hip.so profiles/_ab/$1/libc3d_hip.digest $P/lib/; export C3D_EXTRA_HIPCC_FLAGS="$(cat profiles/_ab/$1/flags)"; }
use ac_lds
timeout 1500 python -m pytest tests/test_gs_hip.py tests/test_zz_replay_gpu.py -m gpu -x -q 2>&1 | tail -3
bash profiles/ab_run.sh r06ac/step "ac_dir ac_lds" 3 - | cut -c1-400
bash profiles/ab_run.sh r06ac/fwd64 "ac_dir ac_lds" 2 - --mode fwd --views-per-gpu 64 --steps 10 | cut -c1-300
bash profiles/ab_run.sh r06ac/fwd64s1 "ac_dir ac_lds" 2 - --mode fwd --views-per-gpu 64 --streams 1 --steps 10 | cut -c1-300
bash profiles/ab_run.sh r06ac/inference "ac_dir ac_lds" 2 - --render-path boundary --mode fwd --inference-mode on --steps 20 | cut -c1-300
bash profiles/ab_run.sh r06ac/refdefault "ac_dir ac_lds" 2 - --workload ref-default --ref-res 512 --steps 600 --warmup 50 --timed-prof off | cut -c1-100
