"""`from simple_knn._C import distCUDA2` (reference: MVs_Algorithms/GaussianSplatting/main_3DGS_renderer.py:408,420).

distCUDA2(points [N,3] float32, HIP device) -> [N] float32: mean of the squared distances to the 3 nearest other points.
HIP kernel behind include/c3d_knn.h (uniform grid + radix sort, exact); no CPU fallback."""
import ctypes as C

import torch

import c3d_hip as _h


def distCUDA2(points):
    if not torch.is_tensor(points) or not points.is_cuda:
        raise RuntimeError("distCUDA2: points must be a tensor on a HIP device (this build has no CPU path)")
    p = _h.f32c(points.detach()).reshape(-1, 3)
    n = p.shape[0]
    out = torch.zeros((n,), dtype=torch.float32, device=p.device)
    if n == 0:
        return out
    lo, hi = p.min(dim=0).values.tolist(), p.max(dim=0).values.tolist()
    lib = _h.lib()
    with torch.cuda.device(p.device):
        scratch = torch.empty((lib.c3d_knn_scratch_bytes(n),), dtype=torch.uint8, device=p.device)
        _h.check(lib.c3d_knn3_mean_dist2(_h.ptr(p), n, (C.c_float * 3)(*lo), (C.c_float * 3)(*hi), _h.ptr(scratch), _h.ptr(out), _h.stream(p.device)),
                 "c3d_knn3_mean_dist2")
    return out
