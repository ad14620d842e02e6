#!/bin/bash
# bench sweeps on the GPU box: one condensed line per configuration.  usage: bash profiles/run_sweep.sh <tag> <mode> "<args 1>" "<args 2>" ...
TAG=${1:-rXX}; MODE=${2:-fwdbwd}; shift; shift
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
i=0
for args in "$@"; do
  i=$((i+1))
  echo "== $MODE $args"
  env $(echo "$args" | tr ' ' '\n' | grep '=' | grep -v '^--' | tr '\n' ' ') timeout 300 python bench.py --mode $MODE --cpu-baseline off $(echo "$args" | tr ' ' '\n' | grep -v '^[A-Z0-9_]*=' | tr '\n' ' ') 2>$OUT/sweep_$i.err | tee $OUT/sweep_${MODE}_$i.json | python profiles/benchline.py
done
