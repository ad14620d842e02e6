#!/bin/bash
# round 5: the spin of the chained scan out of line (kernel 58 -> 30 KB): pin + that (w), vx + that (vxw), against vx and new on the same box
bash profiles/ab_run.sh r05u "new vx vxw w" 3 "sort_pairs or fused_multi_view or golden or render_views or reproducible"
