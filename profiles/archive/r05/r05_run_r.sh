#!/bin/bash
# round 5: view = XCD for the workgroups of the V-view sort passes (vx) / of the emit-offset scan (sx), against the tree without (new)
bash profiles/ab_run.sh r05r "new vx sx" 3 "sort_pairs or scan or fused_multi_view or golden"
