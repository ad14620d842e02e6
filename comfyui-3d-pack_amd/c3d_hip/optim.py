"""torch.optim front-end of the fused HIP Adam step (include/c3d_optim.h).

Drop-in for the `torch.optim.Adam(l, lr=0.0, eps=1e-15)` the reference builds at
MVs_Algorithms/GaussianSplatting/main_3DGS_renderer.py:449: same param_groups (incl. the "name" keys the LR scheduler and the
densifier look up), same state keys ("step", "exp_avg", "exp_avg_sq"), same update rule.  ONE kernel launch for all tensors of a device
(c3d_adam_step_multi; round 2 issued one per tensor: six launches of ~7 us at the reference's default scene size)."""
import torch

import c3d_hip as _h


class FusedAdam(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8):
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps))

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        lib = _h.lib()
        per_device = {}          # device -> ([c3d_adam_tensor], [tensors kept alive until the launch])
        for group in self.param_groups:
            b1, b2 = group["betas"]
            for p in group["params"]:
                if p.grad is None:
                    continue
                if not p.is_cuda:
                    raise RuntimeError("c3d FusedAdam: parameters must live on a HIP device (no CPU fallback)")
                st = self.state[p]
                if len(st) == 0:
                    st["step"] = 0
                    st["exp_avg"] = torch.zeros_like(p, memory_format=torch.contiguous_format)
                    st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.contiguous_format)
                st["step"] += 1
                if p.numel() == 0:
                    continue
                g = p.grad if p.grad.is_contiguous() else p.grad.contiguous()
                recs, keep = per_device.setdefault(p.device, ([], []))
                recs.append(_h.AdamTensor(_h.ptr(p.data), _h.ptr(g), _h.ptr(st["exp_avg"]), _h.ptr(st["exp_avg_sq"]), p.numel(), int(st["step"]),
                                          float(group["lr"]), float(b1), float(b2), float(group["eps"])))
                keep.append(g)
        for dev, (recs, keep) in per_device.items():
            arr = (_h.AdamTensor * len(recs))(*recs)
            with torch.cuda.device(dev):
                _h.check(lib.c3d_adam_step_multi(arr, len(recs), _h.stream(dev)), "c3d_adam_step_multi")
        return loss
