set -x
cd $GRAFT_REPO_ROOT
python bench.py --steps 6 --warmup 2 --cpu-baseline off 2>&1 | tail -1 > gpurun_out/bench_l4.json
python bench.py --steps 6 --warmup 2 --cpu-baseline off --timed-prof off 2>&1 | tail -1 > gpurun_out/bench_l4_noprof.json
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_SALU SQ_INSTS_LDS -d /tmp/sq -o sq -- python $GRAFT_REPO_ROOT/bench.py --lanes 1 --steps 2 --warmup 1 --cpu-baseline off > /tmp/sq.log 2>&1
ls -R /tmp/sq | head
DB=$(find /tmp/sq -name "*.db" | head -1)
python $GRAFT_REPO_ROOT/profiles/summarize_sq.py $DB $GRAFT_REPO_ROOT/gpurun_out/sq_counters.csv | head -30
