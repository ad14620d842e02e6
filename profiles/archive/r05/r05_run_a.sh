#!/bin/bash
# round 5, first call: the sync-free drop-in path (tests, bench lines of the three render paths), the float32-oracle column of the 1M parity test, kernel trace of the boundary path
cd $GRAFT_REPO_ROOT
T=r05a
mkdir -p gpurun_out/$T
timeout 900 python -m pytest tests/test_gs_hip.py -m gpu -q -k "sync_free or halves or recorded_pair or fused_activation or raw_parameter or call_patterns or edge_cases or bit_reproducible or render_views" 2>&1 | tail -5
timeout 900 python -m pytest tests/test_zz_baseline_1m.py -m gpu -q -s 2>&1 | grep -E "^\[1M|passed|failed|Error|assert" | cut -c1-420 > gpurun_out/$T/baseline_1m.log; tail -30 gpurun_out/$T/baseline_1m.log
timeout 400 python bench.py --steps 20 --warmup 5 2>gpurun_out/$T/bench.err | tail -1 > gpurun_out/$T/bench_n1.json; python profiles/benchline.py < gpurun_out/$T/bench_n1.json
for P in boundary fused; do
  timeout 300 python bench.py --render-path $P --steps 20 --warmup 5 --cpu-baseline off --targets off 2>gpurun_out/$T/bench_$P.err | tail -1 > gpurun_out/$T/bench_$P.json; python profiles/benchline.py < gpurun_out/$T/bench_$P.json
done
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/kb
timeout 240 rocprofv3 --kernel-trace -d /tmp/kb -o kb -- python $GRAFT_REPO_ROOT/bench.py --render-path boundary --steps 3 --warmup 2 --cpu-baseline off --targets off --timed-prof off < /dev/null > /tmp/kb.log 2>&1
python $GRAFT_REPO_ROOT/profiles/summarize_rocpd.py $(find /tmp/kb -name "*.db" | head -1) > $GRAFT_REPO_ROOT/gpurun_out/$T/boundary_kernel_stats.csv
head -30 $GRAFT_REPO_ROOT/gpurun_out/$T/boundary_kernel_stats.csv | cut -c1-160
