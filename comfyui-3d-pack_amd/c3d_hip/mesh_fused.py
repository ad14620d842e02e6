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


# ---------------------------------------------------------------------------------------------------- one view, one call each way (round 2)
class _RenderView(torch.autograd.Function):
    """DiffRastRenderer.render for ssaa = 1 as c3d_mesh_view_fwd / c3d_mesh_view_bwd (include/c3d_mesh.h): the whole op sequence of a view
    enqueued from C.  Inputs that carry gradients: v_offsets (or None) and raw_albedo; everything else is a constant of the view."""

    @staticmethod
    def forward(ctx, v, v_offsets, raw_albedo, f, vt, ft, clip_from_world, bg, H, W, glctx, aa_table, hold):
        import ctypes as C
        from c3d_hip.mesh_sigs import MeshView
        lib = _h.lib()
        dev = v.device
        V, T, Vt = int(v.shape[0]), int(f.shape[0]), int(vt.shape[0])
        Ht, Wt = int(raw_albedo.shape[0]), int(raw_albedo.shape[1])
        d = MeshView(V, T, Vt, int(H), int(W), Ht, Wt, (C.c_float * 16)(*[float(x) for x in clip_from_world.reshape(-1)]), (C.c_float * 3)(*[float(x) for x in bg]))
        state = torch.empty((lib.c3d_mesh_view_state_bytes(V, T, H, W),), dtype=torch.uint8, device=dev)
        image = torch.empty((H, W, 3), dtype=torch.float32, device=dev)
        alpha = torch.empty((H, W, 1), dtype=torch.float32, device=dev)
        scratch = glctx.scratch(lib.c3d_mesh_raster_scratch_bytes(1, H, W, T), dev)
        v_c, ra_c = _h.f32c(v), _h.f32c(raw_albedo)
        vo_c = _h.f32c(v_offsets) if v_offsets is not None else None
        with torch.cuda.device(dev):
            _h.check(lib.c3d_mesh_view_fwd(C.byref(d), _h.ptr(v_c), _h.ptr(vo_c), _h.ptr(f), _h.ptr(vt), _h.ptr(ft), _h.ptr(ra_c), _h.ptr(aa_table), _h.ptr(scratch),
                                           _h.ptr(state), _h.ptr(image), _h.ptr(alpha), _h.stream(dev)), "c3d_mesh_view_fwd")
        ctx.d, ctx.glctx, ctx.aa_table, ctx.geo = d, glctx, aa_table, v_offsets is not None and v_offsets.requires_grad
        ctx.save_for_backward(v_c, vo_c if vo_c is not None else v_c.new_empty(0), ra_c, f, vt, ft, state)
        hold["state"], hold["V"] = state, V            # the renderer reads rast / v_clip out of it for the outputs it produces on demand
        return image, alpha

    @staticmethod
    def backward(ctx, dimage, dalpha):
        import ctypes as C
        lib = _h.lib()
        v_c, vo_c, ra_c, f, vt, ft, state = ctx.saved_tensors
        d, dev = ctx.d, v_c.device
        d_ra = torch.empty_like(ra_c)
        d_v = torch.empty_like(v_c) if ctx.geo else None
        topo = ctx.glctx.vertex_topology(f, d.V) if ctx.geo else None
        scratch = torch.empty((lib.c3d_mesh_view_bwd_scratch_bytes(d.V, d.T, d.H, d.W, d.Ht, d.Wt),), dtype=torch.uint8, device=dev)
        di = _h.f32c(dimage) if dimage is not None else None
        da = _h.f32c(dalpha) if dalpha is not None else None
        with torch.cuda.device(dev):
            _h.check(lib.c3d_mesh_view_bwd(C.byref(d), _h.ptr(v_c), _h.ptr(vo_c) if vo_c.numel() else None, _h.ptr(f), _h.ptr(vt), _h.ptr(ft), _h.ptr(ra_c), _h.ptr(ctx.aa_table),
                                           _h.ptr(topo), _h.ptr(scratch), _h.ptr(state), _h.ptr(di), _h.ptr(da), _h.ptr(d_ra), _h.ptr(d_v), _h.stream(dev)), "c3d_mesh_view_bwd")
        return None, d_v, d_ra, None, None, None, None, None, None, None, None, None, None


def render_view(v, v_offsets, raw_albedo, f, vt, ft, clip_from_world, bg, H, W, glctx, aa_table):
    """-> image [H,W,3], alpha [H,W,1], hold (dict with the saved state: see view_state_tensors)"""
    hold = {}
    image, alpha = _RenderView.apply(v, v_offsets, raw_albedo, f, vt, ft, clip_from_world, bg, H, W, glctx, aa_table, hold)
    return image, alpha, hold


def view_state_tensors(hold, H, W):
    """rast [1,H,W,4] and v_clip [1,V,4] as tensor views of a fused view's saved state (layout: include/c3d_mesh.h, c3d_mesh_view_state_bytes)"""
    state, V = hold["state"], hold["V"]
    a = lambda n: (n + 255) // 256 * 256
    o_rast = a(16 * max(V, 1))
    vclip = state[:16 * V].view(torch.float32).view(1, V, 4)
    rast = state[o_rast:o_rast + 16 * H * W].view(torch.float32).view(1, H, W, 4)
    return rast, vclip
