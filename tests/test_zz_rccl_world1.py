"""RCCL on HIP memory, once: a world-1 "nccl" process group on the box's GPU drives every collective of c3d_hip.parallel -- all-gather and
all-reduce of the flat gradient, the chunked / overlapped all-reduce on row slices, ZeRO-1's all-to-all + in-place all-gather of the parameters and the
all-gather of the moments -- and bench.py's multi-GPU control flow under torchrun with backend nccl.  One rank cannot show scaling (the driver measures that
on an 8-GPU node); it shows that the calls are legal on device buffers, that the stream ordering between the library's launches and RCCL's stream holds
(results are compared with the collective-free path), and that the in-place all-gather is accepted.  Runs in a child process: the process group and the
SKIP_SINGLE_RANK switch must not leak into the other tests."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)

CHILD = r'''
import os, sys, json
import numpy as np, torch
import torch.distributed as dist
sys.path[:0] = [%(root)r, os.path.join(%(root)r, "comfyui-3d-pack_amd")]
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29653")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda:0"))
from c3d_hip import parallel
from c3d_hip.optim import FusedAdam
parallel.SKIP_SINGLE_RANK = False                      # issue every collective although there is one rank
out = {"backend": dist.get_backend()}
g = torch.Generator(device="cpu").manual_seed(3)
N = 20011                                              # not a multiple of anything
shapes = [(N, 3), (N, 1, 3), (N, 15, 3), (N, 1), (N, 3), (N, 4)]
params = [torch.randn(s, generator=g).cuda().requires_grad_(True) for s in shapes]
grads = [torch.randn(s, generator=g).cuda() for s in shapes]

# FlatGrads.exchange, both modes, and the chunked overlapped form: world 1 -> the sum over ranks is the gradient itself (x 1 / world when averaged)
for mode in ("allgather", "allreduce"):
    fg = parallel.FlatGrads(params)
    for v, q in zip(fg.views, grads):
        v.copy_(q)
    fg.exchange(None, mode, average=True)
    torch.cuda.synchronize()
    out["exchange_" + mode] = bool(all(torch.equal(v, q) for v, q in zip(fg.views, grads)))
fg = parallel.FlatGrads(params)
for v, q in zip(fg.views, grads):
    v.copy_(q)
bounds = [0, 4096, 12288, N]
for a, b in zip(bounds[:-1], bounds[1:]):
    fg.views[2][a:b].mul_(2.0)                          # "produce" the rows on the current stream, then start their all-reduce underneath the next range
    fg.exchange_rows(a, b)
fg.exchange_finish()
torch.cuda.synchronize()
want = [q.clone() for q in grads]; want[2] = want[2] * 2.0
out["exchange_rows"] = bool(all(torch.equal(v, q) for v, q in zip(fg.views, want)))

# ZeRO-1 against the replicated step: all_to_all_single + in-place all_gather_into_tensor of the parameters, then all-gather of the moments
lrs = [1.6e-4, 2.5e-3, 1.25e-4, 0.05, 5e-3, 1e-3]
ref_p = [p.detach().clone().requires_grad_(True) for p in params]
ref_opt = FusedAdam([{"params": [p], "lr": lr} for p, lr in zip(ref_p, lrs)], lr=0.0, eps=1e-15)
opt = FusedAdam([{"params": [p], "lr": lr} for p, lr in zip(params, lrs)], lr=0.0, eps=1e-15)
z = parallel.ZeroOneAdam(opt, params, None, average=True)
out["zero_collective"] = bool(z.collective)
for it in range(3):
    for gq, q, rp in zip(z.grads, grads, ref_p):
        gq.copy_(q * (it + 1)); rp.grad = q * (it + 1)
    z.step(); ref_opt.step()
torch.cuda.synchronize()
out["zero1_params"] = bool(all(torch.equal(p.data, rp.data) for p, rp in zip(params, ref_p)))
z.unshard()
out["zero1_moments"] = bool(all(torch.equal(opt.state[p]["exp_avg"], ref_opt.state[rp]["exp_avg"]) and torch.equal(opt.state[p]["exp_avg_sq"], ref_opt.state[rp]["exp_avg_sq"])
                                for p, rp in zip(params, ref_p)))
# the generic path (torch.cat of .grad, all-gather, rank-ordered sum)
for p, q in zip(params, grads):
    p.grad = q.clone()
parallel.exchange_gradients(params, None, "allgather", average=False)
torch.cuda.synchronize()
out["exchange_gradients"] = bool(all(torch.equal(p.grad, q) for p, q in zip(params, grads)))
dist.barrier(); dist.destroy_process_group()
print("RESULT " + json.dumps(out))
'''


def test_rccl_world1_drives_every_collective_on_hip_memory():
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", MASTER_ADDR="127.0.0.1", MASTER_PORT="29653")
    r = subprocess.run([sys.executable, "-c", CHILD % {"root": ROOT}], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("RESULT ")][-1]
    out = json.loads(line[len("RESULT "):])
    print(out)
    assert out.pop("backend") == "nccl"
    assert all(out.values()), out


@pytest.mark.parametrize("mode,exchange", [("train", "zero1"), ("fwdbwd", "allreduce")])
def test_bench_under_torchrun_with_nccl(mode, exchange):
    """bench.py's N > 1 control flow (torch.distributed.run rendezvous, nccl process group bound to the device, barriers, the exchange after the fused step) with
    the one rank a one-GPU box can hold: --gpus 1 under torchrun, C3D_BENCH_FORCE_DIST=1 makes bench.py create the group and issue the collectives anyway."""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", C3D_BENCH_FORCE_DIST="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1", "--master-port", "29654",
           os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1", "--gaussians", "100000", "--width", "640", "--height", "360",
           "--views-per-gpu", "2", "--mode", mode, "--exchange", exchange, "--cpu-baseline", "off", "--targets", "off"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 1 and line["value"] > 0
    assert line["config"].get("dist_backend") == "nccl", line["config"]
