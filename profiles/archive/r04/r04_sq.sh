#!/bin/bash
# SQ counters of the default bench command (lanes 1), two passes
T=${1:-r04sq}
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/$T
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/sqa /tmp/sqb
timeout 240 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_BRANCH SQ_INSTS_SMEM SQ_INSTS_VMEM SQ_WAVES SQ_BUSY_CYCLES -d /tmp/sqa -o sqa -- python $R/bench.py --lanes 1 --steps 2 --warmup 1 --cpu-baseline off --targets off --timed-prof off < /dev/null > /tmp/sqa.log 2>&1
python $R/profiles/summarize_sq.py $(find /tmp/sqa -name "*.db" | head -1) $OUT/${T}_sq_instruction_mix_lanes1.csv | head -12
timeout 240 rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES -d /tmp/sqb -o sqb -- python $R/bench.py --lanes 1 --steps 2 --warmup 1 --cpu-baseline off --targets off --timed-prof off < /dev/null > /tmp/sqb.log 2>&1
python $R/profiles/summarize_sq.py $(find /tmp/sqb -name "*.db" | head -1) $OUT/${T}_sq_pipe_activity_lanes1.csv | head -12
