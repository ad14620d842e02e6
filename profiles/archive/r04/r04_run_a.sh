#!/bin/bash
# round 4, first GPU pass of the V-wide chain: GPU test-suite, then the bench lines at several group / lane settings
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r04a
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/r04a/pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/r04a/pytest.log
tail -5 gpurun_out/r04a/pytest.log
for L in 1 2 4; do
  timeout 300 python bench.py --steps 20 --warmup 3 --lanes $L --targets off --cpu-baseline off > gpurun_out/r04a/bench_fwdbwd_l$L.json 2> gpurun_out/r04a/bench_fwdbwd_l$L.err
  python -c "import json,sys; d=json.loads(open('gpurun_out/r04a/bench_fwdbwd_l$L.json').read().strip().splitlines()[-1]); print('fwdbwd lanes $L', d['value'], d['ms_per_step'])"
done
for cfg in "1 8" "2 8" "1 16" "2 16" "2 4"; do
  set -- $cfg
  timeout 300 python bench.py --mode fwd --views-per-gpu 64 --steps 5 --warmup 2 --lanes $1 --group $2 --cpu-baseline off > gpurun_out/r04a/bench_fwd_l$1_g$2.json 2> gpurun_out/r04a/bench_fwd_l$1_g$2.err
  python -c "import json,sys; d=json.loads(open('gpurun_out/r04a/bench_fwd_l$1_g$2.json').read().strip().splitlines()[-1]); print('fwd lanes $1 group $2', d['value'], d['ms_per_step'])"
done
timeout 300 python bench.py --mode train --steps 10 --warmup 3 --targets off --cpu-baseline off > gpurun_out/r04a/bench_train.json 2> gpurun_out/r04a/bench_train.err
python -c "import json,sys; d=json.loads(open('gpurun_out/r04a/bench_train.json').read().strip().splitlines()[-1]); print('train', d['value'], d['ms_per_step'])"
