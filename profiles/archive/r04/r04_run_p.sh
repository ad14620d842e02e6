#!/bin/bash
# MS-SSIM after the tile pipelining: residency 3 vs 4, then the kernel trace of the training step
cd $GRAFT_REPO_ROOT
T=r04p
mkdir -p gpurun_out/$T
i=0
for F in "" "-DMS_MIN_BLOCKS=3" ""; do
  export C3D_EXTRA_HIPCC_FLAGS="$F"
  timeout 300 python bench.py --mode train --steps 10 --warmup 3 --targets off --cpu-baseline off 2>/dev/null | tail -1 > gpurun_out/$T/train_$i.json
  echo "[$F]"; python profiles/benchline.py < gpurun_out/$T/train_$i.json
  i=$((i+1))
done
export C3D_EXTRA_HIPCC_FLAGS=""
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/ktt
timeout 240 rocprofv3 --kernel-trace -d /tmp/ktt -o ktt -- python $R/bench.py --mode train --steps 3 --warmup 1 --cpu-baseline off --targets off --timed-prof off < /dev/null > /tmp/ktt.log 2>&1
python $R/profiles/summarize_rocpd.py $(find /tmp/ktt -name "*.db" | head -1) > $R/gpurun_out/$T/${T}_train_kernel_stats.csv
grep -E 'k_ms_|k_adam|k_composite_bwd<true' $R/gpurun_out/$T/${T}_train_kernel_stats.csv | cut -c1-120
