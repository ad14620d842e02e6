#!/bin/bash
# round 5: stealing with the view's count requested before the ticket (w2) against stealing with the count behind the barrier (w) and no stealing (vxw), same box
bash profiles/ab_run.sh r05v "vxw w w2" 3 "sort_pairs or fused_multi_view or golden or render_views or reproducible or unequal"
