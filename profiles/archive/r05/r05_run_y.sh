#!/bin/bash
# round 5: k_onesweep's workgroups stay and draw tile after tile (one per residency slot; home view = XCD, stealing at the end) = p; against pinned without stealing, one workgroup per tile of the capacity (vxw), and unpinned (new)
bash profiles/ab_run.sh r05y "p" 0 "sort_pairs or scan or fused_multi_view or golden or render_views or reproducible or unequal or halves or edge or trainer_densify"
bash profiles/ab_run.sh r05y "new vxw p" 3 -
