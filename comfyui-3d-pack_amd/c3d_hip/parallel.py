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


class FlatGrads:
    """One flat fp32 buffer holding the gradients of all parameter tensors (16-byte aligned segments); `views[i]` has the shape of
    params[i] and aliases the buffer, so the kernels that produce the gradients write straight into what the collective sends:
    no torch.cat before the exchange and no copy back after it."""

    def __init__(self, params):
        sizes = [(p.numel() + 3) // 4 * 4 for p in params]
        self.flat = torch.zeros((sum(sizes),), dtype=torch.float32, device=params[0].device)
        self.views, off = [], 0
        for p, sz in zip(params, sizes):
            self.views.append(self.flat[off:off + p.numel()].view(p.shape))
            off += sz
        self._gathered = None
        # chunked, overlapped exchange (exchange_rows / exchange_finish): the row-sliceable "big" tensors -- those holding most of the bytes (f_rest: 45 of the
        # 59 floats per Gaussian) -- go range by range; what is left forms a few contiguous spans of the flat buffer, reduced once at the end
        total = float(sum(p.numel() for p in params)) or 1.0
        self._big = [i for i, p in enumerate(params) if p.numel() / total >= 0.25]
        self._rest_spans, off, start = [], 0, None
        for i, (p, sz) in enumerate(zip(params, sizes)):
            if i in self._big:
                if start is not None:
                    self._rest_spans.append((start, off)); start = None
            elif start is None:
                start = off
            off += sz
        if start is not None:
            self._rest_spans.append((start, off))
        self._works = []

    def exchange(self, group=None, mode="allgather", average=False):
        """sum (or mean) over the ranks of `group`, in place; identical bits on every rank"""
        if not dist.is_available() or not dist.is_initialized():
            return
        world = dist.get_world_size(group)
        if world == 1:
            return
        flat, n = self.flat, self.flat.numel()
        scale = 1.0 / world if average else 1.0
        if mode == "allgather":
            if self._gathered is None or self._gathered.numel() != world * n:
                self._gathered = torch.empty((world * n,), dtype=flat.dtype, device=flat.device)
            dist.all_gather_into_tensor(self._gathered, flat, group=group)
            if flat.is_cuda:        # one streaming pass, ranks added in rank order
                import c3d_hip as _h
                with torch.cuda.device(flat.device):
                    _h.check(_h.lib().c3d_reduce_ranks_f32(_h.ptr(flat), _h.ptr(self._gathered), world, n, scale, _h.stream(flat.device)), "c3d_reduce_ranks_f32")
            else:                   # gloo / CPU tensors (host-logic tests): same order, torch ops
                g = self._gathered.view(world, n)
                total = g[0].clone()
                for r in range(1, world):
                    total += g[r]
                flat.copy_(total * scale if average else total)
        elif mode == "allreduce":
            dist.all_reduce(flat, group=group)
            if average:
                flat.mul_(scale)
        else:
            raise ValueError("mode must be 'allgather' or 'allreduce'")


    # ---- chunked exchange, overlapped with the kernels that produce the gradients (all-reduce mode) ------------------------------------------
    def exchange_rows(self, g0, g1, group=None):
        """Rows [g0, g1) of the big gradient tensors are final on the current stream: start their all-reduce now (async: the backend's stream waits
        for what the current stream has enqueued so far, then runs underneath whatever is enqueued next -- the next Gaussian range of the
        per-Gaussian backward pass).  Pair with exchange_finish()."""
        if not dist.is_available() or not dist.is_initialized() or dist.get_world_size(group) == 1 or g1 <= g0:
            return
        for i in self._big:
            self._works.append(dist.all_reduce(self.views[i][g0:g1], group=group, async_op=True))

    def exchange_finish(self, group=None, average=False):
        """all-reduce everything exchange_rows() does not cover (contiguous spans of the flat buffer), then make the current stream wait for all of it"""
        if not dist.is_available() or not dist.is_initialized():
            return
        world = dist.get_world_size(group)
        if world == 1:
            return
        for a, b in self._rest_spans:
            self._works.append(dist.all_reduce(self.flat[a:b], group=group, async_op=True))
        for w in self._works:
            w.wait()
        self._works = []
        if average:
            self.flat.mul_(1.0 / world)


def broadcast_parameters(params, src=0, group=None):
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        for p in params:
            dist.broadcast(p.data, src=src, group=group)
