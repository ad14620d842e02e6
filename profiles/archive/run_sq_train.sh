# SQ counters of the training step's kernels (MS-SSIM included), one view lane.  usage (through gpurun): bash profiles/run_sq_train.sh
set -x
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/sqa /tmp/sqb
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_BRANCH SQ_INSTS_SMEM SQ_INSTS_VMEM SQ_WAVES SQ_BUSY_CYCLES -d /tmp/sqa -o sqa -- python $R/bench.py --mode train --lanes 1 --steps 2 --warmup 1 --cpu-baseline off --timed-prof off > /tmp/sqa.log 2>&1
python $R/profiles/summarize_sq.py $(find /tmp/sqa -name "*.db" | head -1) $R/gpurun_out/sq_train_mix.csv | grep -E "kernel|k_ms|k_loss|adam"
rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES -d /tmp/sqb -o sqb -- python $R/bench.py --mode train --lanes 1 --steps 2 --warmup 1 --cpu-baseline off --timed-prof off > /tmp/sqb.log 2>&1
python $R/profiles/summarize_sq.py $(find /tmp/sqb -name "*.db" | head -1) $R/gpurun_out/sq_train_active.csv | grep -E "kernel|k_ms|k_loss|adam"
