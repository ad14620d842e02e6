#!/bin/bash
# round 5: the histogram kernel reads 2 / 4 / 8 tiles per workgroup before it flushes its counters (h2 / h4 / h8) against one (base = committed tree)
bash profiles/ab_run.sh r05h1 "h4" 0 "sort_pairs or fused_multi_view or golden or render_views or reproducible or unequal"
bash profiles/ab_run.sh r05h1 "base h2 h4 h8" 3 -
