"""TEST INFRASTRUCTURE: a CPU stand-in for the module the reference imports as `diff_gaussian_rasterization`, backed by
oracle/gs_oracle.c (float32 build): oracle forward, oracle backward, wired as one autograd function.  It records the settings and tensors of every rasterizer call (`CALLS`), so that the
glue in front of the rasterizer -- the reference's GaussianSplattingRenderer.render and this repo's mirror -- can be run on the CPU
over the same stand-in and compared (tests/golden/make_golden_ref_gs_render.py, tests/test_ref_render_glue.py).  Never imported by
the product."""
from typing import NamedTuple

import numpy as np
import torch

from oracle import gs_oracle as O

CALLS = []


class GaussianRasterizationSettings(NamedTuple):
    image_height: int
    image_width: int
    tanfovx: float
    tanfovy: float
    bg: torch.Tensor
    scale_modifier: float
    viewmatrix: torch.Tensor
    projmatrix: torch.Tensor
    sh_degree: int
    campos: torch.Tensor
    prefiltered: bool
    debug: bool


def _np(t):
    return None if t is None else t.detach().cpu().numpy().astype(np.float32)


RECORD = True        # the training-loop fixtures switch the call log off (hundreds of calls)


class _Rasterize(torch.autograd.Function):
    """oracle forward / oracle backward as one differentiable op (the argument order of the dependency's own autograd function)"""

    @staticmethod
    def forward(ctx, means3D, means2D, shs, colors_precomp, opacities, scales, rotations, cov3D_precomp, st):
        color, radii, depth, alpha, state = O.forward(_np(means3D), _np(opacities), st, shs=_np(shs), colors_precomp=_np(colors_precomp),
                                                      scales=_np(scales), rotations=_np(rotations), cov3D_precomp=_np(cov3D_precomp), dtype=np.float32)
        ctx.state = state
        radii_t = torch.from_numpy(radii)
        ctx.mark_non_differentiable(radii_t)
        return torch.from_numpy(color), radii_t, torch.from_numpy(depth), torch.from_numpy(alpha)

    @staticmethod
    def backward(ctx, dcolor, dradii, ddepth, dalpha):
        g = O.backward(ctx.state, _np(dcolor), None if ddepth is None else _np(ddepth)[0], None if dalpha is None else _np(dalpha)[0])
        inp = ctx.state.inputs
        T = lambda k, present: torch.from_numpy(np.ascontiguousarray(g[k], dtype=np.float32)) if present else None
        return (T("means3D", True), T("means2D", True), T("shs", inp["shs"] is not None), T("colors", inp["colors_precomp"] is not None),
                T("opacities", True), T("scales", inp["scales"] is not None), T("rotations", inp["rotations"] is not None),
                T("cov3D", inp["cov3D_precomp"] is not None), None)


class GaussianRasterizer:
    def __init__(self, raster_settings):
        self.raster_settings = raster_settings

    def __call__(self, means3D, means2D, opacities, shs=None, colors_precomp=None, scales=None, rotations=None, cov3D_precomp=None):
        rs = self.raster_settings
        st = {k: (_np(v) if torch.is_tensor(v) else v) for k, v in rs._asdict().items()}
        if RECORD:
            CALLS.append({"settings": st, "means3D": _np(means3D), "means2D": _np(means2D), "opacities": _np(opacities), "shs": _np(shs),
                          "colors_precomp": _np(colors_precomp), "scales": _np(scales), "rotations": _np(rotations), "cov3D_precomp": _np(cov3D_precomp)})
        return _Rasterize.apply(means3D, means2D, shs, colors_precomp, opacities, scales, rotations, cov3D_precomp, st)
