#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06l
timeout 2400 python -m pytest tests -m gpu -q -x --deselect tests/test_zz_replay_gpu.py::test_the_drivers_eight_rank_command_with_the_all_gather_exchange --deselect tests/test_zz_baseline_1m.py::test_render_views_raw_matches_oracle_forward_on_all_64_orbit_cameras 2>&1 | grep -v "^$" | tail -15 > gpurun_out/r06l/pytest.log; tail -15 gpurun_out/r06l/pytest.log
echo "== node defaults: r05 tree --timed-prof off | work"
for i in 1 2 3; do
  bash profiles/ab_tree_run.sh r06l/ref_r05_untimed_$i "r05" 1 --workload ref-default --ref-res 512 --steps 600 --warmup 50 --timed-prof off
  bash profiles/ab_tree_run.sh r06l/ref_work_$i "work" 1 --workload ref-default --ref-res 512 --steps 600 --warmup 50
done
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf /tmp/kr
timeout 600 rocprofv3 --kernel-trace -d /tmp/kr -o kr -- python $R/bench.py --workload ref-default --ref-res 512 --steps 400 --warmup 50 --cpu-baseline off --timed-prof off < /dev/null > /tmp/kr.log 2>&1
python $R/profiles/iteration_timeline.py $(find /tmp/kr -name "*.db" | head -1) 20 > $R/gpurun_out/r06l/ref_default_iteration_timeline.txt
cat $R/gpurun_out/r06l/ref_default_iteration_timeline.txt | cut -c1-120
