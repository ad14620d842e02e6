#!/bin/bash
# round 3, GPU call L: full GPU suite on the current code + default line
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r03l; mkdir -p $OUT; cd $R
timeout 1500 python -m pytest tests -m gpu -q -s < /dev/null 2>&1 | grep -v "amdgpu.ids" > $OUT/pytest.log; tail -4 $OUT/pytest.log; grep -n "FAILED\|Error" $OUT/pytest.log | head
timeout 300 python bench.py --steps 30 --warmup 5 --cpu-baseline off < /dev/null 2>/dev/null | tail -1 > $OUT/bench_default.json; head -c 250 $OUT/bench_default.json
