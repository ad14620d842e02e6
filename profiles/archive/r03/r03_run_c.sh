#!/bin/bash
# round 3, GPU call C: GPU suite, PMC calibration (known-byte kernels under rocprofv3), bin-ahead A/B, targets line, the reference's default workload
set -x
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r03c; mkdir -p $OUT; cd $R
timeout 1500 python -m pytest tests -m gpu -q -s 2>&1 | grep -v "amdgpu.ids" > $OUT/pytest.log; tail -5 $OUT/pytest.log
python bench.py --steps 20 --warmup 5 --cpu-baseline off 2>$OUT/bench_new.err | tail -1 > $OUT/bench_new.json
C3D_BIN_AHEAD=0 python bench.py --steps 20 --warmup 5 --cpu-baseline off --targets off 2>/dev/null | tail -1 > $OUT/bench_noahead.json
python bench.py --mode train --steps 20 --warmup 5 --cpu-baseline off --targets off 2>/dev/null | tail -1 > $OUT/bench_train.json
python bench.py --workload ref-default --ref-res 512 --steps 700 --warmup 50 2>$OUT/ref512.err | tail -1 > $OUT/bench_ref_default_512.json
python bench.py --workload ref-default --ref-res 1024 --steps 700 --warmup 50 2>$OUT/ref1024.err | tail -1 > $OUT/bench_ref_default_1024.json
cd /tmp && export TMPDIR=/tmp
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o /tmp/pmc_calib $R/profiles/microbench/pmc_calib.hip 2>/dev/null
rm -rf /tmp/cf /tmp/cw
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d /tmp/cf -o cf -- /tmp/pmc_calib > /tmp/calib_known.json 2>/tmp/cf.log
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d /tmp/cw -o cw -- /tmp/pmc_calib > /dev/null 2>/tmp/cw.log
python $R/profiles/summarize_calib.py $(find /tmp/cf -name "*.db" | head -1) $(find /tmp/cw -name "*.db" | head -1) /tmp/calib_known.json $OUT/r03_pmc_calibration.json
cd $R
for f in $OUT/bench_*.json; do echo $f; head -c 600 $f; echo; done
tail -3 $OUT/*.err
grep -n "\[1M\|passed\|failed" $OUT/pytest.log | cut -c1-330
