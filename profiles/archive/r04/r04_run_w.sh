#!/bin/bash
# bench.py launched the way the driver launches N > 1 (torch.distributed.run, backend nccl = RCCL) at world 1: what the exchange code paths cost on one GPU
cd $GRAFT_REPO_ROOT
T=r04w
mkdir -p gpurun_out/$T
for X in allreduce allgather zero1; do
  timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 1 --mode train --exchange $X --steps 20 --warmup 5 --targets off --cpu-baseline off 2>/dev/null | tail -1 > gpurun_out/$T/${T}_train_nccl_world1_$X.json
  python - <<PY
import json
d=json.load(open("gpurun_out/$T/${T}_train_nccl_world1_$X.json"))
print("$X", d["value"], d["ms_per_step"], d["config"].get("exchange"), d["config"].get("dist_backend"))
PY
done
timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29534 bench.py --gpus 1 --steps 20 --warmup 5 --targets off --cpu-baseline off 2>/dev/null | tail -1 > gpurun_out/$T/${T}_fwdbwd_nccl_world1_allreduce.json
python profiles/benchline.py < gpurun_out/$T/${T}_fwdbwd_nccl_world1_allreduce.json
