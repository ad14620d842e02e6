#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/tl
timeout 600 rocprofv3 --kernel-trace -d /tmp/tl -o tl -- python $R/bench.py --steps 3 --warmup 1 --cpu-baseline off --timed-prof off "$@" > /tmp/tl.log 2>&1
python $R/profiles/summarize_timeline.py $(find /tmp/tl -name "*.db" | head -1)
