#!/bin/bash
# round 5: view = (linear block id) % V with work stealing in the sort passes + the same mapping in the emit-offset scan (pin), + in the histogram (pinh), against bitop3-only (new)
bash profiles/ab_run.sh r05s "new pin pinh" 3 "sort_pairs or scan or fused_multi_view or golden or unequal or render_views or reproducible or halves"
