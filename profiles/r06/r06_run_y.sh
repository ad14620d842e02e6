#!/bin/bash
# round 6, second session: MS-SSIM derivative maps with a row pitch of whole tiles (profiles/ab_prepare.sh y_nop work "-DC3D_MS_NO_PITCH"; y_pit work)
cd $GRAFT_REPO_ROOT
bash profiles/ab_run.sh r06y/train "y_nop y_pit" 3 "msssim or trainer_fused or fused_multi_view or mirror" --mode train --steps 20 | cut -c1-420
bash profiles/ab_run.sh r06y/refdefault "y_nop y_pit" 2 - --workload ref-default --ref-res 512 --steps 600 --warmup 50 --timed-prof off | cut -c1-100
