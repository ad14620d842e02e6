"""Fused glue of DiffRastRenderer.render (include/c3d_mesh.h, "renderer glue"): the vertex transform and the final shade of
MVs_Algorithms/DiffRastMesh/diff_mesh_renderer.py as one HIP kernel each way instead of ~25 torch elementwise / GEMM launches per view.
Same values as the torch chain (tests/test_mesh_hip.py compares them); HIP tensors only."""
import torch

import c3d_hip as _h


class _Transform(torch.autograd.Function):
    @staticmethod
    def forward(ctx, v, M):
        v_c, M_c = _h.f32c(v), _h.f32c(M)
        V = v_c.shape[0]
        out = torch.empty((V, 4), dtype=torch.float32, device=v_c.device)
        with torch.cuda.device(v_c.device):
            _h.check(_h.lib().c3d_mesh_transform_fwd(_h.ptr(v_c), _h.ptr(M_c), V, _h.ptr(out), _h.stream(v_c.device)), "c3d_mesh_transform_fwd")
        ctx.save_for_backward(M_c)
        return out

    @staticmethod
    def backward(ctx, dout):
        (M,) = ctx.saved_tensors
        d = _h.f32c(dout)
        V = d.shape[0]
        dv = torch.empty((V, 3), dtype=torch.float32, device=d.device)
        with torch.cuda.device(d.device):
            _h.check(_h.lib().c3d_mesh_transform_bwd(_h.ptr(M), _h.ptr(d), V, _h.ptr(dv), _h.stream(d.device)), "c3d_mesh_transform_bwd")
        return dv, None


def transform_vertices(v, M):
    """[V,3], 4x4 (device tensor, no gradient) -> [V,4] = [v, 1] @ M^T"""
    return _Transform.apply(v, M)


class _Shade(torch.autograd.Function):
    @staticmethod
    def forward(ctx, albedo, alpha, bg):
        a_c, al_c, bg_c = _h.f32c(albedo), _h.f32c(alpha), _h.f32c(bg)
        P = al_c.numel()
        image, alpha_out = torch.empty_like(a_c), torch.empty_like(al_c)
        with torch.cuda.device(a_c.device):
            _h.check(_h.lib().c3d_mesh_shade_fwd(_h.ptr(a_c), _h.ptr(al_c), _h.ptr(bg_c), P, _h.ptr(image), _h.ptr(alpha_out), _h.stream(a_c.device)), "c3d_mesh_shade_fwd")
        ctx.save_for_backward(a_c, al_c, bg_c)
        return image, alpha_out

    @staticmethod
    def backward(ctx, dimage, dalpha_out):
        a_c, al_c, bg_c = ctx.saved_tensors
        P = al_c.numel()
        dalbedo, dalpha = torch.empty_like(a_c), torch.empty_like(al_c)
        di = _h.f32c(dimage) if dimage is not None else None
        da = _h.f32c(dalpha_out) if dalpha_out is not None else None
        with torch.cuda.device(a_c.device):
            _h.check(_h.lib().c3d_mesh_shade_bwd(_h.ptr(a_c), _h.ptr(al_c), _h.ptr(bg_c), P, _h.ptr(di) if di is not None else None,
                                                 _h.ptr(da) if da is not None else None, _h.ptr(dalbedo), _h.ptr(dalpha), _h.stream(a_c.device)), "c3d_mesh_shade_bwd")
        return dalbedo, dalpha, None


def shade(albedo, alpha, bg):
    """albedo [H,W,3], alpha [H,W,1] (before its clamp), bg [3] -> (clamp(a*albedo + (1-a)*bg, 0, 1), a = clamp(alpha, 0, 1))"""
    return _Shade.apply(albedo, alpha, bg)
