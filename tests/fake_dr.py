"""TEST INFRASTRUCTURE: a CPU stand-in for the module the reference imports as `nvdiffrast.torch`, backed by oracle/mesh_oracle.c
(float64 arithmetic, float32 tensors out): each op is an autograd function whose backward is the oracle's backward of that op.

It exists so that the *glue* around the four ops -- the reference's DiffRastRenderer.render and this repo's mirror of it -- can be run
on the CPU over the very same op implementation and compared (tests/golden/make_golden_ref_render.py, tests/test_ref_render_glue.py).
It records every call (`CALLS`) so that the two op sequences can be compared as well.  Never imported by the product.
"""
import numpy as np
import torch

from oracle import mesh_oracle as M

CALLS = []


def _np(t):
    return t.detach().cpu().numpy() if torch.is_tensor(t) else np.asarray(t)


def _t(a):
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32))


class RasterizeCudaContext:
    def __init__(self, device=None):
        self.device = device


class RasterizeGLContext(RasterizeCudaContext):
    def __init__(self, output_db=True, mode='automatic', device=None):
        super().__init__(device)


def _g(a):
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32))


class _Rasterize(torch.autograd.Function):
    @staticmethod
    def forward(ctx, pos, tri, resolution):
        rast, db = M.rasterize(_np(pos), _np(tri), resolution, dtype=np.float64)
        ctx.saved = (_np(pos), _np(tri), rast)
        db_t = _t(db)
        ctx.mark_non_differentiable(db_t)
        return _t(rast), db_t

    @staticmethod
    def backward(ctx, dy, ddb):
        pos, tri, rast = ctx.saved
        return _g(M.rasterize_bwd(pos, tri, rast, _np(dy), dtype=np.float64)), None, None


class _Interpolate(torch.autograd.Function):
    @staticmethod
    def forward(ctx, attr, rast, tri, rast_db, diff_attrs):
        out, da = M.interpolate(_np(attr), _np(rast), _np(tri), None if rast_db is None else _np(rast_db), diff_attrs, dtype=np.float64)
        ctx.saved = (_np(attr), _np(rast), _np(tri))
        da_t = _t(da)
        ctx.mark_non_differentiable(da_t)
        return _t(out), da_t

    @staticmethod
    def backward(ctx, dy, dda):
        attr, rast, tri = ctx.saved
        dattr, drast = M.interpolate_bwd(attr, rast, tri, _np(dy), dtype=np.float64)
        return _g(dattr), _g(drast), None, None, None


class _Texture(torch.autograd.Function):
    @staticmethod
    def forward(ctx, tex, uv, filter_mode, boundary_mode):
        ctx.saved = (_np(tex), _np(uv), filter_mode, boundary_mode)
        return _t(M.texture(_np(tex), _np(uv), filter_mode, boundary_mode, dtype=np.float64))

    @staticmethod
    def backward(ctx, dy):
        tex, uv, fm, bm = ctx.saved
        dtex, duv = M.texture_bwd(tex, uv, _np(dy), fm, bm, dtype=np.float64)
        return _g(dtex), _g(duv), None, None


class _Antialias(torch.autograd.Function):
    @staticmethod
    def forward(ctx, color, rast, pos, tri):
        ctx.saved = (_np(color), _np(rast), _np(pos), _np(tri))
        return _t(M.antialias(*ctx.saved, dtype=np.float64))

    @staticmethod
    def backward(ctx, dy):
        color, rast, pos, tri = ctx.saved
        dcolor, dpos = M.antialias_bwd(color, rast, pos, tri, _np(dy), dtype=np.float64)
        return _g(dcolor), None, _g(dpos), None


def rasterize(glctx, pos, tri, resolution, ranges=None, grad_db=True):
    assert ranges is None
    CALLS.append(("rasterize", tuple(pos.shape), tuple(tri.shape), tuple(int(r) for r in resolution)))
    return _Rasterize.apply(pos, tri, (int(resolution[0]), int(resolution[1])))


def interpolate(attr, rast, tri, rast_db=None, diff_attrs=None):
    CALLS.append(("interpolate", tuple(attr.shape), tuple(tri.shape), rast_db is not None, diff_attrs if isinstance(diff_attrs, str) or diff_attrs is None
                  else tuple(diff_attrs)))
    return _Interpolate.apply(attr, rast, tri, rast_db, diff_attrs)


def texture(tex, uv, uv_da=None, mip_level_bias=None, mip=None, filter_mode='auto', boundary_mode='wrap', max_mip_level=None):
    CALLS.append(("texture", tuple(tex.shape), tuple(uv.shape), uv_da is not None, filter_mode, boundary_mode))
    if filter_mode == 'auto':
        filter_mode = 'linear' if (uv_da is None and mip_level_bias is None) else 'linear-mipmap-linear'
    if filter_mode in ('nearest', 'linear'):
        return _Texture.apply(tex, uv, filter_mode, boundary_mode)
    return _t(M.texture_mip(_np(tex), _np(uv), None if uv_da is None else _np(uv_da), None if mip_level_bias is None else _np(mip_level_bias),
                            filter_mode=filter_mode, boundary_mode=boundary_mode, max_mip_level=max_mip_level, dtype=np.float64))   # forward only


def antialias(color, rast, pos, tri, topology_hash=None, pos_gradient_boost=1.0):
    CALLS.append(("antialias", tuple(color.shape), tuple(pos.shape), tuple(tri.shape)))
    return _Antialias.apply(color, rast, pos, tri)
