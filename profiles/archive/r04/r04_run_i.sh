#!/bin/bash
# round 4: mesh step -- owner bits (TriOwned), batched vertex gather, no rast_db; tests, bench, kernel trace, HBM traffic and SQ counters per kernel
R=$GRAFT_REPO_ROOT
T=${1:-r04i2}
OUT=$R/gpurun_out/$T
mkdir -p $OUT
export HSA_ENABLE_IPC_MODE_LEGACY=0
cd $R
timeout 1200 python -m pytest tests/test_mesh_hip.py tests/test_zmesh_ext.py tests/test_zz_ref_consumers.py tests/test_zz_replay_gpu.py -m gpu -x -q > $OUT/pytest_mesh.log 2>&1; echo "pytest rc $?" >> $OUT/pytest_mesh.log
tail -4 $OUT/pytest_mesh.log
for i in 1 2; do timeout 600 python bench.py --workload mesh --steps 20 --warmup 3 --cpu-baseline off 2>/dev/null | tail -1 > $OUT/bench_mesh_$i.json; python profiles/benchline.py < $OUT/bench_mesh_$i.json; done
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/km /tmp/mf /tmp/mw /tmp/sa /tmp/sb
BM="python $R/bench.py --workload mesh --steps 3 --warmup 1 --cpu-baseline off --timed-prof off"
timeout 300 rocprofv3 --kernel-trace -d /tmp/km -o km -- $BM < /dev/null > /tmp/km.log 2>&1
python $R/profiles/summarize_rocpd.py $(find /tmp/km -name "*.db" | head -1) > $OUT/${T}_mesh_kernel_stats.csv
head -14 $OUT/${T}_mesh_kernel_stats.csv
timeout 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d /tmp/mf -o mf -- $BM < /dev/null > /tmp/mf.log 2>&1
timeout 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d /tmp/mw -o mw -- $BM < /dev/null > /tmp/mw.log 2>&1
python $R/profiles/summarize_pmc.py $(find /tmp/mf -name "*.db" | head -1) $(find /tmp/mw -name "*.db" | head -1) $OUT/${T}_mesh_pmc_traffic.json mesh > $OUT/${T}_mesh_pmc_traffic.csv
cat $OUT/${T}_mesh_pmc_traffic.csv | head -20
timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_SMEM -d /tmp/sa -o sa -- $BM < /dev/null > /tmp/sa.log 2>&1
python $R/profiles/summarize_sq.py $(find /tmp/sa -name "*.db" | head -1) $OUT/${T}_mesh_sq_instruction_mix.csv | head -12
timeout 300 rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY -d /tmp/sb -o sb -- $BM < /dev/null > /tmp/sb.log 2>&1
python $R/profiles/summarize_sq.py $(find /tmp/sb -name "*.db" | head -1) $OUT/${T}_mesh_sq_pipe_activity.csv | head -12
