#!/bin/bash
# round 6, second session: k_emit requests the next chunk's inputs before it writes this chunk's pairs (profiles/ab_prepare.sh ad_old work "-DC3D_EMIT_NO_PREFETCH"; ad_new work)
cd $GRAFT_REPO_ROOT
bash profiles/ab_run.sh r06ad/step "ad_old ad_new" 3 "forward_matches or internal_state or edge or config1 or golden or unequal or wider or exact_when" | cut -c1-400
bash profiles/ab_run.sh r06ad/fwd64s1 "ad_old ad_new" 2 - --mode fwd --views-per-gpu 64 --streams 1 --steps 10 | cut -c1-300
