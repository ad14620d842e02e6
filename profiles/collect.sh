#!/bin/bash
# Collect the committed evidence of a round on the GPU box: bench lines of every BASELINE config, rocprofv3 kernel stats, PMC traffic (separate
# --pmc passes, never combined with other trace domains), SQ counters.  Every JSON / sidecar carries the code digest of the library it was
# measured on (c3d_hip.code_digest(): csrc/*, include/*.h, flags); bench.py refuses traffic / issue figures of any other digest.
# usage (from the repo root, through gpurun):  bash profiles/collect.sh r03z [quick]
# Every command runs under `timeout` with stdin closed: a tool that falls back to reading stdin must not be able to hang the box (round 3 lost 15 GPU-minutes to that).
set -x
TAG=${1:-rXX}
QUICK=${2:-}
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
# --- PMC + SQ first: the bench lines below then find traffic / issue figures of THIS code under profiles/ (copied there on the box as well)
rm -rf /tmp/pf /tmp/pw /tmp/mf /tmp/mw /tmp/sqa /tmp/sqb
timeout 240 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d /tmp/pf -o pf -- python $R/bench.py --lanes 1 --steps 2 --warmup 1 --cpu-baseline off --timed-prof off < /dev/null > /tmp/pf.log 2>&1
timeout 240 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d /tmp/pw -o pw -- python $R/bench.py --lanes 1 --steps 2 --warmup 1 --cpu-baseline off --timed-prof off < /dev/null > /tmp/pw.log 2>&1
python $R/profiles/summarize_pmc.py $(find /tmp/pf -name "*.db" | head -1) $(find /tmp/pw -name "*.db" | head -1) $OUT/${TAG}_pmc_traffic.json > $OUT/${TAG}_pmc_traffic.csv
timeout 240 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d /tmp/mf -o mf -- python $R/bench.py --workload mesh --steps 2 --warmup 1 --cpu-baseline off --timed-prof off < /dev/null > /tmp/mf.log 2>&1
timeout 240 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d /tmp/mw -o mw -- python $R/bench.py --workload mesh --steps 2 --warmup 1 --cpu-baseline off --timed-prof off < /dev/null > /tmp/mw.log 2>&1
python $R/profiles/summarize_pmc.py $(find /tmp/mf -name "*.db" | head -1) $(find /tmp/mw -name "*.db" | head -1) $OUT/${TAG}_mesh_pmc_traffic.json mesh > $OUT/${TAG}_mesh_pmc_traffic.csv
timeout 240 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_BRANCH SQ_INSTS_SMEM SQ_INSTS_VMEM SQ_WAVES SQ_BUSY_CYCLES -d /tmp/sqa -o sqa -- python $R/bench.py --lanes 1 --steps 2 --warmup 1 --cpu-baseline off --timed-prof off < /dev/null > /tmp/sqa.log 2>&1
python $R/profiles/summarize_sq.py $(find /tmp/sqa -name "*.db" | head -1) $OUT/${TAG}_sq_instruction_mix_lanes1.csv | head -6
timeout 240 rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES -d /tmp/sqb -o sqb -- python $R/bench.py --lanes 1 --steps 2 --warmup 1 --cpu-baseline off --timed-prof off < /dev/null > /tmp/sqb.log 2>&1
python $R/profiles/summarize_sq.py $(find /tmp/sqb -name "*.db" | head -1) $OUT/${TAG}_sq_pipe_activity_lanes1.csv | head -6
cp $OUT/${TAG}_pmc_traffic.json $OUT/${TAG}_mesh_pmc_traffic.json $OUT/${TAG}_sq_instruction_mix_lanes1.csv $OUT/${TAG}_sq_instruction_mix_lanes1.csv.meta.json $OUT/${TAG}_sq_pipe_activity_lanes1.csv $OUT/${TAG}_sq_pipe_activity_lanes1.csv.meta.json $R/profiles/
# --- kernel stats (timeout 240 rocprofv3 --kernel-trace of the default command and of the mesh workload)
rm -rf /tmp/kt /tmp/km /tmp/ktt
# (round 6: the step's kernels and the forward target's in TWO tables -- the default command also renders the 64-camera forward target, trains and runs the mesh step after its timed region)
timeout 240 rocprofv3 --kernel-trace -d /tmp/kt -o kt -- python $R/bench.py --steps 3 --warmup 1 --cpu-baseline off --targets off < /dev/null > /tmp/kt.log 2>&1
python $R/profiles/summarize_rocpd.py $(find /tmp/kt -name "*.db" | head -1) > $OUT/${TAG}_fwdbwd_kernel_stats.csv
rm -rf /tmp/kf
# (--streams 1: the kernels of the table run alone; the bench line of config 2 below runs the product default, four HIP streams)
timeout 240 rocprofv3 --kernel-trace -d /tmp/kf -o kf -- python $R/bench.py --mode fwd --views-per-gpu 64 --streams 1 --steps 3 --warmup 1 --cpu-baseline off --targets off < /dev/null > /tmp/kf.log 2>&1
python $R/profiles/summarize_rocpd.py $(find /tmp/kf -name "*.db" | head -1) > $OUT/${TAG}_fwd64_kernel_stats.csv
timeout 240 rocprofv3 --kernel-trace -d /tmp/ktt -o ktt -- python $R/bench.py --mode train --steps 3 --warmup 1 --cpu-baseline off < /dev/null > /tmp/ktt.log 2>&1
python $R/profiles/summarize_rocpd.py $(find /tmp/ktt -name "*.db" | head -1) > $OUT/${TAG}_train_kernel_stats.csv
timeout 240 rocprofv3 --kernel-trace -d /tmp/km -o km -- python $R/bench.py --workload mesh --steps 2 --warmup 1 --cpu-baseline off < /dev/null > /tmp/km.log 2>&1
python $R/profiles/summarize_rocpd.py $(find /tmp/km -name "*.db" | head -1) > $OUT/${TAG}_mesh_kernel_stats.csv
# --- bench lines, un-profiled: config 4's per-GPU share / the headline (default command), config 2 (64 views, forward), config 3 (training step), config 5 (mesh)
cd $R
timeout 400 python bench.py --steps 20 --warmup 5 < /dev/null 2>/dev/null | tail -1 > $OUT/${TAG}_bench_n1.json
timeout 300 python bench.py --mode fwd --steps 20 --warmup 5 --cpu-baseline off --targets off < /dev/null 2>/dev/null | tail -1 > $OUT/${TAG}_bench_fwd_n1.json
timeout 300 python bench.py --mode fwd --views-per-gpu 64 --steps 20 --warmup 3 --cpu-baseline off --targets off < /dev/null 2>/dev/null | tail -1 > $OUT/${TAG}_bench_fwd64_n1.json
timeout 300 python bench.py --mode fwd --views-per-gpu 64 --streams 1 --steps 20 --warmup 3 --cpu-baseline off --targets off < /dev/null 2>/dev/null | tail -1 > $OUT/${TAG}_bench_fwd64_one_stream_n1.json
timeout 300 python bench.py --mode train --steps 20 --warmup 5 --cpu-baseline off --targets off < /dev/null 2>/dev/null | tail -1 > $OUT/${TAG}_bench_train_n1.json
timeout 400 python bench.py --workload mesh --steps 40 --warmup 5 < /dev/null 2>/dev/null | tail -1 > $OUT/${TAG}_bench_mesh_n1.json
timeout 300 python bench.py --workload mesh --render-path fused --steps 20 --warmup 5 --cpu-baseline off < /dev/null 2>/dev/null | tail -1 > $OUT/${TAG}_bench_mesh_view_api_n1.json
if [ -z "$QUICK" ]; then
  timeout 300 python bench.py --mode train --loss full-torch --steps 5 --warmup 3 --cpu-baseline off --targets off < /dev/null 2>/dev/null | tail -1 > $OUT/${TAG}_bench_train_torchloss_n1.json
  timeout 300 python bench.py --mode train --loss l1alpha --steps 20 --warmup 5 --cpu-baseline off --targets off < /dev/null 2>/dev/null | tail -1 > $OUT/${TAG}_bench_train_l1alpha_n1.json
  timeout 300 python bench.py --render-path fused --steps 20 --warmup 5 --cpu-baseline off --targets off < /dev/null 2>/dev/null | tail -1 > $OUT/${TAG}_bench_renderer_api_n1.json
  timeout 300 python bench.py --render-path boundary --steps 20 --warmup 5 --cpu-baseline off --targets off < /dev/null 2>/dev/null | tail -1 > $OUT/${TAG}_bench_boundary_api_n1.json
  timeout 300 python bench.py --render-path boundary --sync-free off --steps 20 --warmup 5 --cpu-baseline off --targets off < /dev/null 2>/dev/null | tail -1 > $OUT/${TAG}_bench_boundary_api_syncfree_off_n1.json
  timeout 300 python bench.py --render-path boundary --sync-free unverified --steps 20 --warmup 5 --cpu-baseline off --targets off < /dev/null 2>/dev/null | tail -1 > $OUT/${TAG}_bench_boundary_api_unverified_n1.json
  timeout 300 python bench.py --render-path boundary --mode fwd --inference-mode on --steps 20 --warmup 5 --cpu-baseline off --targets off < /dev/null 2>/dev/null | tail -1 > $OUT/${TAG}_bench_boundary_inference_n1.json
  timeout 300 python bench.py --render-path boundary --mode fwd --inference-mode on --forward-only off --steps 20 --warmup 5 --cpu-baseline off --targets off < /dev/null 2>/dev/null | tail -1 > $OUT/${TAG}_bench_boundary_inference_forward_only_off_n1.json
  timeout 300 python bench.py --workload ref-default --ref-res 512 --steps 700 --warmup 50 --cpu-baseline off < /dev/null 2>/dev/null | tail -1 > $OUT/${TAG}_bench_ref_default_512.json
  timeout 120 python profiles/microbench/sort_phases.py < /dev/null > $OUT/${TAG}_sort_phases.txt 2>&1
  (cd /tmp && rm -rf /tmp/kr && timeout 600 rocprofv3 --kernel-trace -d /tmp/kr -o kr -- python $R/bench.py --workload ref-default --ref-res 512 --steps 400 --warmup 50 --cpu-baseline off --timed-prof off < /dev/null > /tmp/kr.log 2>&1; python $R/profiles/iteration_timeline.py $(find /tmp/kr -name "*.db" | head -1) 20 > $OUT/${TAG}_ref_default_iteration_timeline.txt)
fi
if [ -z "$QUICK" ]; then timeout 1800 python -m pytest tests -m gpu -q -s < /dev/null > $OUT/${TAG}_gpu_pytest.log 2>&1; tail -3 $OUT/${TAG}_gpu_pytest.log; fi
head -12 $OUT/${TAG}_fwdbwd_kernel_stats.csv
head -12 $OUT/${TAG}_pmc_traffic.csv
for f in $OUT/${TAG}_bench_*.json; do echo $f; timeout 20 python profiles/benchline.py < $f; done
