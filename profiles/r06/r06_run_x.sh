#!/bin/bash
# round 6, second session: SQ counters of the mesh step's kernels (what does k_ras_tri wait for?)
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/r06x
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/sqa /tmp/sqb /tmp/sqc
timeout 240 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_SMEM SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES -d /tmp/sqa -o sqa -- python $R/bench.py --workload mesh --steps 2 --warmup 1 --cpu-baseline off --timed-prof off < /dev/null > /tmp/sqa.log 2>&1
python $R/profiles/summarize_sq.py $(find /tmp/sqa -name "*.db" | head -1) $R/gpurun_out/r06x/r06x_mesh_sq_instruction_mix.csv | head -8
timeout 240 rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INST_CYCLES_VMEM -d /tmp/sqb -o sqb -- python $R/bench.py --workload mesh --steps 2 --warmup 1 --cpu-baseline off --timed-prof off < /dev/null > /tmp/sqb.log 2>&1
python $R/profiles/summarize_sq.py $(find /tmp/sqb -name "*.db" | head -1) $R/gpurun_out/r06x/r06x_mesh_sq_pipe_activity.csv | head -8
tail -3 /tmp/sqb.log
timeout 240 rocprofv3 --kernel-trace --pmc TCP_TCC_READ_REQ_sum TCP_TOTAL_ACCESSES_sum TCC_HIT_sum TCC_MISS_sum TCC_ATOMIC_sum TCC_EA0_RDREQ_sum -d /tmp/sqc -o sqc -- python $R/bench.py --workload mesh --steps 2 --warmup 1 --cpu-baseline off --timed-prof off < /dev/null > /tmp/sqc.log 2>&1
python $R/profiles/summarize_sq.py $(find /tmp/sqc -name "*.db" | head -1) $R/gpurun_out/r06x/r06x_mesh_cache.csv | head -8
tail -3 /tmp/sqc.log
