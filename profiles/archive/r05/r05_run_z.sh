#!/bin/bash
# round 5: staying workgroups with the look next to the draw, one workgroup per tile for launches of at most two rounds (p3); against p (first staying version), vxw (pinned, one workgroup per tile of the capacity), new (unpinned)
bash profiles/ab_run.sh r05zz "p3" 0 "sort_pairs or scan or fused_multi_view or golden or render_views or reproducible or unequal or halves or edge or trainer_densify"
bash profiles/ab_run.sh r05zz "new vxw p p3" 3 -
