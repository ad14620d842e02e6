#!/bin/bash
# round 4: quick check of a binning-chain change: the binning / fused-path tests, then the two bench lines with per-kernel tables
cd $GRAFT_REPO_ROOT
T=${1:-r04b}
mkdir -p gpurun_out/$T
timeout 900 python -m pytest tests/test_gs_hip.py -m gpu -x -q -k "scan or sort or internal_state or forward_matches or backward_matches or fused_multi_view or render_views or edge_cases or golden or bit_reproducible or medium or scaling_modifier or halves or pair_activity" > gpurun_out/$T/pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/$T/pytest.log
tail -4 gpurun_out/$T/pytest.log
timeout 300 python bench.py --steps 20 --warmup 3 --lanes 1 --targets off --cpu-baseline off > gpurun_out/$T/bench_fwdbwd_l1.json 2> gpurun_out/$T/bench_fwdbwd_l1.err
timeout 300 python bench.py --mode fwd --views-per-gpu 64 --steps 5 --warmup 2 --lanes 1 --group 8 --cpu-baseline off > gpurun_out/$T/bench_fwd_l1_g8.json 2> gpurun_out/$T/bench_fwd_l1_g8.err
timeout 300 python bench.py --mode fwd --views-per-gpu 64 --steps 5 --warmup 2 --lanes 2 --group 8 --cpu-baseline off > gpurun_out/$T/bench_fwd_l2_g8.json 2> gpurun_out/$T/bench_fwd_l2_g8.err
python - <<PY
import json
for f in ['bench_fwdbwd_l1','bench_fwd_l1_g8','bench_fwd_l2_g8']:
    try:
        d=json.loads(open('gpurun_out/$T/%s.json'%f).read().strip().splitlines()[-1])
        print(f, d['value'], d['ms_per_step'])
        if d.get('kernels'): print('   '+'  '.join('%s %.3f'%(k.replace('gs_',''),v['avg_ms']*v['launches']/d['steps']) for k,v in d['kernels'].items()))
    except Exception as e: print(f, 'failed', e); print(open('gpurun_out/$T/%s.err'%f).read()[-2000:])
PY
