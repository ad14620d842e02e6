#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/r06j
rm -rf /tmp/kr
timeout 600 rocprofv3 --kernel-trace -d /tmp/kr -o kr -- python $R/bench.py --workload ref-default --ref-res 512 --steps 400 --warmup 50 --cpu-baseline off --timed-prof off < /dev/null > /tmp/kr.log 2>&1
tail -1 /tmp/kr.log | cut -c1-300
python $R/profiles/iteration_timeline.py $(find /tmp/kr -name "*.db" | head -1) 20 > $R/gpurun_out/r06j/ref_default_iteration_timeline_untimed.txt
cat $R/gpurun_out/r06j/ref_default_iteration_timeline_untimed.txt
