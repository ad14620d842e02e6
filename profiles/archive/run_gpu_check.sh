#!/bin/bash
# One GPU call: the whole -m gpu suite (captured output of passing tests kept: the [grad] / [replay] lines), then the bench lines.
# usage (through gpurun, from the repo root):  bash profiles/run_gpu_check.sh <tag> [pytest args...]
TAG=${1:-rXX}; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
timeout 1200 python -m pytest tests -m gpu -q -rP "$@" > $OUT/pytest.log 2>&1
echo "pytest rc=$?" | tee -a $OUT/pytest.log
grep -E "^\[grad\]|^\[replay|^\[mesh|passed|failed|^FAILED|^ERROR" $OUT/pytest.log | tail -150
for mode in fwdbwd fwd train; do
  timeout 400 python bench.py --mode $mode --cpu-baseline off > $OUT/bench_$mode.json 2> $OUT/bench_$mode.err
  echo "bench $mode rc=$?"; python profiles/benchline.py < $OUT/bench_$mode.json 2>/dev/null || tail -3 $OUT/bench_$mode.err
done
