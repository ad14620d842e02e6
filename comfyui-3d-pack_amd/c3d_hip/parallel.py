"""View-parallel data parallelism for the shared-Gaussian training loop (new -- the reference is single GPU).

One process per GPU (`torch.distributed`, backend "nccl" = RCCL over xGMI on ROCm; "gloo" in the CPU tests).  Views
are independent units: every rank renders its shard of the step's views against the full, replicated parameter set
and the only exchange per step is the dense parameter gradient.  Two exchange modes:
  "allgather": every rank receives every rank's gradient and sums them in RANK ORDER (the north star's wording;
               fixed summation order -> replicas stay bit-identical, and the backward kernels are deterministic);
  "allreduce": one RCCL all-reduce (sum order chosen by the library).
xGMI is point-to-point (7 links per GPU), so the all-gather sends one gradient copy per link concurrently.
  "zero1":     ZeRO stage 1 (class ZeroOneAdam below): every rank owns 1/n of the flat parameter vector.  The gradient is
               reduce-scattered as ONE all-to-all (each rank sends slice j of its gradient straight to rank j: seven concurrent single-hop
               transfers on the seven xGMI links instead of a ring) followed by a fixed-rank-order sum of the n received copies of the owned
               slice; Adam runs on the owned slice only (moments: 1/n of the memory, 1/n of the 1.65 GB the replicated step streams); one
               all-gather returns the updated parameters.  Same bytes on the wire as the ring all-reduce (2 (n-1)/n x 236 MB per rank), same
               bits as the replicated "allgather" step.
"""
import math

import torch
import torch.distributed as dist

# A process group of ONE rank has nothing to exchange and every function below returns early for it.  Set to False to issue the collectives anyway: a
# world-1 "nccl" group on a single GPU then drives every RCCL call of this module on HIP memory (tests/test_zz_rccl_world1.py) -- the only RCCL the
# one-GPU boxes of the build loop can run.
SKIP_SINGLE_RANK = True

# Bytes per rank and step at N Gaussians (F = 59 N floats = 236 MB at N = 1e6, SH degree 3) over n ranks, and what they cost on xGMI (7 links x ~153 GB/s per GPU,
# point to point; MI355X_MICROARCH / DESIGN 6) -- the model the exchange modes were chosen with:
#   allreduce  ring: sends 2 (n - 1) / n x 4F bytes, per-link bound: n = 8 -> 413 MB / ~153 GB/s per link direction ~ 2.7 ms if one ring, ~0.4 ms over 7 rings
#   allgather  receives (n - 1) x 4F bytes (1.65 GB at n = 8) over 7 links in parallel: ~1.5 ms, + a 1.9 GB streaming sum (~0.4 ms): fixed summation order
#   zero1      all-to-all: sends (n - 1) / n x 4F (seven single-hop transfers of F / 8 each: ~0.2 ms), Adam on 1 / n, all-gather of (n - 1) / n x 4F back: ~0.2 ms
def exchange_bytes(n_floats, world, mode):
    """bytes one rank SENDS per step for a gradient of n_floats float32 values"""
    f = 4.0 * n_floats
    if world <= 1:
        return 0.0
    return {"allreduce": 2.0 * (world - 1) / world * f, "allgather": (world - 1) * f, "zero1": 2.0 * (world - 1) / world * f}[mode]


def _single(world):
    return world == 1 and SKIP_SINGLE_RANK


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
    if _single(world):
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
        if _single(world):
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
        if not dist.is_available() or not dist.is_initialized() or _single(dist.get_world_size(group)) or g1 <= g0:
            return
        for i in self._big:
            self._works.append(dist.all_reduce(self.views[i][g0:g1], group=group, async_op=True))

    def exchange_finish(self, group=None, average=False):
        """all-reduce everything exchange_rows() does not cover (contiguous spans of the flat buffer), then make the current stream wait for all of it"""
        if not dist.is_available() or not dist.is_initialized():
            return
        world = dist.get_world_size(group)
        if _single(world):
            return
        for a, b in self._rest_spans:
            self._works.append(dist.all_reduce(self.flat[a:b], group=group, async_op=True))
        for w in self._works:
            w.wait()
        self._works = []
        if average:
            self.flat.mul_(1.0 / world)


def status_max(group=None):
    """-> the `status_sync` of a FusedViewStep whose step ends in a collective: replaces a small int32 device tensor by its maximum over the ranks of `group` (None for one
    rank: nothing to agree on).  4-byte words, one all-reduce per step, in stream order behind the step's kernels; with it an overflow of the pair capacity on ONE rank makes
    EVERY rank regrow and redo the step, where round 5 raised on that rank and left the others waiting in a collective."""
    if not dist.is_available() or not dist.is_initialized() or _single(dist.get_world_size(group)):
        return None
    return lambda words: dist.all_reduce(words, op=dist.ReduceOp.MAX, group=group)


def adam_reference_(p, g, m, v, lr, b1, b2, eps, t):
    """torch.optim.Adam's single-tensor update (no weight decay, no amsgrad), statement by statement, on CPU tensors or slices of them:
    what ZeroOneAdam runs on its owned slice where there is no HIP device (the gloo tests); bit-identical to torch.optim.Adam(foreach=False)"""
    m.mul_(b1).add_(g, alpha=1 - b1)
    v.mul_(b2).addcmul_(g, g, value=1 - b2)
    bc1, bc2 = 1 - b1 ** t, 1 - b2 ** t
    denom = (v.sqrt() / math.sqrt(bc2)).add_(eps)
    p.addcdiv_(m, denom, value=-(lr / bc1))


class ZeroOneAdam:
    """ZeRO-1 step of the shared-Gaussian loop: reduce-scatter(gradient) -> Adam on the owned 1/n -> all-gather(parameters).

    `optimizer` supplies the hyper-parameters (param_groups: lr / betas / eps per tensor, as GaussianModel.training_setup builds them and the LR
    schedule updates them); its `state` is NOT used while this object lives: the moments exist only for the owned slice.  `params` are re-pointed
    into one flat buffer (`p.data` becomes a view), the gradients live in a second one (`grads[i]`, shaped like params[i]: the kernels / autograd
    write straight into what the all-to-all sends).  Both buffers are padded so that every rank owns the same number of elements, a multiple of 4
    (16-byte aligned slices for the float4 kernels).

    step():  all_to_all_single  ->  fixed-order sum of the n copies of the owned slice (c3d_reduce_ranks_f32 on a HIP device)  ->  Adam on the
    intersections of the owned slice with each parameter tensor (c3d_adam_step on a HIP device, adam_reference_ on the CPU)  ->  all_gather_into_tensor
    of the parameter buffer.  Every element is summed in rank order and updated by the same elementwise rule as in the replicated step, so the
    parameters are bit-identical to mode "allgather" + a full Adam step (tests/test_host_logic.py, worlds 2 and 4).

    Densification / opacity reset / checkpoints need whole moments: unshard() all-gathers them into optimizer.state (torch.optim layout); a new
    ZeroOneAdam built afterwards adopts that state and drops it again."""

    def __init__(self, optimizer, params, group=None, average=True):
        self.opt, self.params, self.group, self.average = optimizer, list(params), group, average
        on = dist.is_available() and dist.is_initialized()
        self.world = dist.get_world_size(group) if on else 1
        self.rank = dist.get_rank(group) if on else 0
        dev = self.params[0].device
        sizes = [(p.numel() + 3) // 4 * 4 for p in self.params]
        q = 4 * self.world
        total = (sum(sizes) + q - 1) // q * q
        self.total, self.chunk = total, total // self.world
        self.lo, self.hi = self.rank * self.chunk, (self.rank + 1) * self.chunk
        self.flat_p = torch.zeros((total,), dtype=torch.float32, device=dev)
        self.flat_g = torch.zeros((total,), dtype=torch.float32, device=dev)
        self.grads, self.offsets, off = [], [], 0
        for p, sz in zip(self.params, sizes):
            self.flat_p[off:off + p.numel()].copy_(p.data.reshape(-1))
            p.data = self.flat_p[off:off + p.numel()].view(p.shape)
            self.grads.append(self.flat_g[off:off + p.numel()].view(p.shape))
            self.offsets.append(off)
            off += sz
        self.m = torch.zeros((self.chunk,), dtype=torch.float32, device=dev)
        self.v = torch.zeros((self.chunk,), dtype=torch.float32, device=dev)
        self.gsum = torch.zeros((self.chunk,), dtype=torch.float32, device=dev)
        self.collective = on and not _single(self.world)
        self.recv = torch.empty((total,), dtype=torch.float32, device=dev) if self.collective else None
        self.t = 0
        self.group_of = {}
        for gr in optimizer.param_groups:
            for p in gr["params"]:
                self.group_of[id(p)] = gr
        self._adopt()

    def _segments(self):
        """(param index, a, b): the element range [a, b) of the flat buffers where params[i] meets the owned slice"""
        for i, (p, off) in enumerate(zip(self.params, self.offsets)):
            a, b = max(off, self.lo), min(off + p.numel(), self.hi)
            if a < b:
                yield i, a, b

    def _adopt(self):
        for i, a, b in self._segments():
            st = self.opt.state.get(self.params[i])
            if st and "exp_avg" in st:
                off = self.offsets[i]
                self.m[a - self.lo:b - self.lo].copy_(st["exp_avg"].reshape(-1)[a - off:b - off])
                self.v[a - self.lo:b - self.lo].copy_(st["exp_avg_sq"].reshape(-1)[a - off:b - off])
        steps = [int(st["step"]) for st in (self.opt.state.get(p) for p in self.params) if st and "step" in st]
        self.t = max(steps) if steps else 0
        for p in self.params:
            if p in self.opt.state:
                del self.opt.state[p]

    def zero_grad(self):
        self.flat_g.zero_()

    def step(self):
        scale = 1.0 / self.world if self.average else 1.0
        if self.collective:
            dist.all_to_all_single(self.recv, self.flat_g, group=self.group)       # recv[r * chunk : (r + 1) * chunk] = rank r's copy of MY slice
            if self.flat_g.is_cuda:
                import c3d_hip as _h
                with torch.cuda.device(self.flat_g.device):
                    _h.check(_h.lib().c3d_reduce_ranks_f32(_h.ptr(self.gsum), _h.ptr(self.recv), self.world, self.chunk, scale, _h.stream(self.flat_g.device)), "c3d_reduce_ranks_f32")
            else:
                r = self.recv.view(self.world, self.chunk)
                total = r[0].clone()
                for k in range(1, self.world):
                    total += r[k]
                self.gsum.copy_(total * scale if self.average else total)
        else:
            self.gsum.copy_(self.flat_g[self.lo:self.hi])
        self.t += 1
        for i, a, b in self._segments():
            gr = self.group_of[id(self.params[i])]
            b1, b2 = gr["betas"]
            ps, gs, ms, vs = self.flat_p[a:b], self.gsum[a - self.lo:b - self.lo], self.m[a - self.lo:b - self.lo], self.v[a - self.lo:b - self.lo]
            if ps.is_cuda:
                import c3d_hip as _h
                with torch.cuda.device(ps.device):
                    _h.check(_h.lib().c3d_adam_step(_h.ptr(ps), _h.ptr(gs), _h.ptr(ms), _h.ptr(vs), b - a, float(gr["lr"]), float(b1), float(b2), float(gr["eps"]),
                                                    int(self.t), _h.stream(ps.device)), "c3d_adam_step")
            else:
                adam_reference_(ps, gs, ms, vs, float(gr["lr"]), float(b1), float(b2), float(gr["eps"]), self.t)
        if self.collective:
            # in place on a HIP device (RCCL / NCCL define the in-place all-gather: the input IS the rank's slot of the output) -- no 1 / n-sized copy per step;
            # gloo (the CPU tests) gets a separate input tensor
            mine = self.flat_p[self.lo:self.hi]
            dist.all_gather_into_tensor(self.flat_p, mine if self.flat_p.is_cuda else mine.clone(), group=self.group)

    def unshard(self):
        """all-gather the moments and hand them to the wrapped optimizer in torch.optim.Adam's layout (densification surgery, checkpoints)"""
        fm, fv = self.m, self.v
        if self.collective:
            fm, fv = torch.empty_like(self.flat_p), torch.empty_like(self.flat_p)
            dist.all_gather_into_tensor(fm, self.m, group=self.group)
            dist.all_gather_into_tensor(fv, self.v, group=self.group)
        # torch.optim.Adam keeps `step` as a float32 tensor and rejects anything else ("state_steps must contain singleton tensors"); the fused optimizer
        # (c3d_hip.optim.FusedAdam) counts with a Python int
        as_tensor = isinstance(self.opt, torch.optim.Adam)
        for p, off in zip(self.params, self.offsets):
            self.opt.state[p] = {"step": torch.tensor(float(self.t)) if as_tensor else self.t,
                                 "exp_avg": fm[off:off + p.numel()].view(p.shape).clone(), "exp_avg_sq": fv[off:off + p.numel()].view(p.shape).clone()}


def broadcast_parameters(params, src=0, group=None):
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        for p in params:
            dist.broadcast(p.data, src=src, group=group)
