#!/bin/bash
# round 5: p4 = two instances: workgroups that stay (launches of more than two rounds: tile sort) / one workgroup per tile (depth sort of <= 8 views); both with home view = XCD
bash profiles/ab_run.sh r05z2 "p4" 0 "sort_pairs or scan or fused_multi_view or golden or render_views or reproducible or unequal or halves or edge or trainer_densify"
bash profiles/ab_run.sh r05z2 "new vxw p p4" 3 -
