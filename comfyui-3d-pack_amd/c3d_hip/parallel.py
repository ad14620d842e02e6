"""View-parallel data parallelism for the shared-Gaussian training loop (new -- the reference is single GPU).

One process per GPU (`torch.distributed`, backend "nccl" = RCCL over xGMI on ROCm; "gloo" in the CPU tests).  Views
are independent units: every rank renders its shard of the step's views against the full, replicated parameter set
and the only exchange per step is the dense parameter gradient.  Two exchange modes:
  "allgather": every rank receives every rank's gradient and sums them in RANK ORDER (the north star's wording;
               fixed summation order -> replicas stay bit-identical, and the backward kernels are deterministic);
  "allreduce": one RCCL all-reduce (sum order chosen by the library).
xGMI is point-to-point (7 links per GPU), so the all-gather sends one gradient copy per link concurrently.
"""
import torch
import torch.distributed as dist


def shard_views(indices, rank, world):
    """views of this rank: a strided slice, so every rank gets the same count when len(indices) % world == 0"""
    return list(indices[rank::world])


def flatten_grads(params):
    """[N, sum(cols)] dense gradient of tensors that all have N rows (59 columns for SH degree 3)"""
    n = params[0].shape[0]
    return torch.cat([(p.grad if p.grad is not None else torch.zeros_like(p)).reshape(n, -1) for p in params], dim=1)


def unflatten_into_grads(flat, params):
    off = 0
    for p in params:
        w = p[0].numel() if p.shape[0] else 0
        g = flat[:, off:off + w].reshape(p.shape)
        if p.grad is None:
            p.grad = g.clone()
        else:
            p.grad.copy_(g)
        off += w


def exchange_gradients(params, group=None, mode="allgather", average=False):
    """sum (or mean) the gradients of `params` over the ranks of `group`; every rank ends with identical bits"""
    if not dist.is_available() or not dist.is_initialized():
        return
    world = dist.get_world_size(group)
    if world == 1:
        return
    flat = flatten_grads(params).contiguous()
    if mode == "allgather":
        buf = torch.empty((world * flat.shape[0],) + tuple(flat.shape[1:]), dtype=flat.dtype, device=flat.device)
        dist.all_gather_into_tensor(buf, flat, group=group)      # concatenation along dim 0 (works on RCCL and gloo)
        buf = buf.view((world,) + tuple(flat.shape))
        total = buf[0].clone()
        for r in range(1, world):      # fixed rank order
            total += buf[r]
    elif mode == "allreduce":
        total = flat
        dist.all_reduce(total, group=group)
    else:
        raise ValueError("mode must be 'allgather' or 'allreduce'")
    if average:
        total /= world
    unflatten_into_grads(total, params)


def broadcast_parameters(params, src=0, group=None):
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        for p in params:
            dist.broadcast(p.data, src=src, group=group)
