#!/bin/bash
# round 4: mesh parity tests after the V-wide conversion + a kernel trace of the mesh bench command
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r04g
mkdir -p $OUT
export HSA_ENABLE_IPC_MODE_LEGACY=0
cd $R
timeout 1500 python -m pytest tests/test_mesh_hip.py tests/test_zmesh_ext.py tests/test_zz_ref_consumers.py tests/test_zz_replay_gpu.py -m gpu -x -q > $OUT/pytest_mesh.log 2>&1; echo "pytest rc $?" >> $OUT/pytest_mesh.log
tail -5 $OUT/pytest_mesh.log
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/km
timeout 300 rocprofv3 --kernel-trace -d /tmp/km -o km -- python $R/bench.py --workload mesh --steps 5 --warmup 2 --cpu-baseline off --timed-prof off < /dev/null > /tmp/km.log 2>&1
tail -2 /tmp/km.log | head -c 400
python $R/profiles/summarize_rocpd.py $(find /tmp/km -name "*.db" | head -1) > $OUT/r04g_mesh_kernel_stats.csv
head -40 $OUT/r04g_mesh_kernel_stats.csv
