"""TEST INFRASTRUCTURE: a CPU stand-in for the module the reference imports as `nvdiffrast.torch`, backed by oracle/mesh_oracle.c
(float64 arithmetic, float32 tensors out), forward only.

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


def rasterize(glctx, pos, tri, resolution, ranges=None, grad_db=True):
    assert ranges is None
    CALLS.append(("rasterize", tuple(pos.shape), tuple(tri.shape), tuple(int(r) for r in resolution)))
    rast, db = M.rasterize(_np(pos), _np(tri), (int(resolution[0]), int(resolution[1])), dtype=np.float64)
    return _t(rast), _t(db)


def interpolate(attr, rast, tri, rast_db=None, diff_attrs=None):
    CALLS.append(("interpolate", tuple(attr.shape), tuple(tri.shape), rast_db is not None, diff_attrs if isinstance(diff_attrs, str) or diff_attrs is None
                  else tuple(diff_attrs)))
    out, da = M.interpolate(_np(attr), _np(rast), _np(tri), None if rast_db is None else _np(rast_db), diff_attrs, dtype=np.float64)
    return _t(out), _t(da)


def texture(tex, uv, uv_da=None, mip_level_bias=None, mip=None, filter_mode='auto', boundary_mode='wrap', max_mip_level=None):
    CALLS.append(("texture", tuple(tex.shape), tuple(uv.shape), uv_da is not None, filter_mode, boundary_mode))
    if filter_mode == 'auto':
        filter_mode = 'linear' if (uv_da is None and mip_level_bias is None) else 'linear-mipmap-linear'
    if filter_mode in ('nearest', 'linear'):
        return _t(M.texture(_np(tex), _np(uv), filter_mode, boundary_mode, dtype=np.float64))
    return _t(M.texture_mip(_np(tex), _np(uv), None if uv_da is None else _np(uv_da), None if mip_level_bias is None else _np(mip_level_bias),
                            filter_mode=filter_mode, boundary_mode=boundary_mode, max_mip_level=max_mip_level, dtype=np.float64))


def antialias(color, rast, pos, tri, topology_hash=None, pos_gradient_boost=1.0):
    CALLS.append(("antialias", tuple(color.shape), tuple(pos.shape), tuple(tri.shape)))
    return _t(M.antialias(_np(color), _np(rast), _np(pos), _np(tri), dtype=np.float64))
