#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06n
timeout 600 python -m pytest tests/test_gs_hip.py -m gpu -q -x -k "sort or scan or drop_in or sync_free or beyond" 2>&1 | tail -4
( time python -c "import __graft_entry__ as g; g.smoke()" ) 2>&1 | tail -5
( time python bench.py > gpurun_out/r06n/bench_default.json 2> gpurun_out/r06n/bench_default.err ) 2>&1 | tail -4
python -c "
import json; d=json.load(open('gpurun_out/r06n/bench_default.json')); print(d['value'], d['ms_per_step'], d['roofline']['traffic'], d['roofline'].get('stale'), {k:v for k,v in d['targets'].items() if k.endswith('Mpx') or k.endswith('ms_per_step')})"
