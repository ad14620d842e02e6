#!/bin/bash
# round 5: the histogram kernel with 8 / 4 keys per thread (i8 / i4: twice / four times the workgroups) against 16 (base)
bash profiles/ab_run.sh r05h2 "i8" 0 "sort_pairs or fused_multi_view or golden or render_views"
bash profiles/ab_run.sh r05h2 "base i8 i4" 3 -
