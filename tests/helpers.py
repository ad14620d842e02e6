"""Shared helpers for the parity tests: scene -> oracle call, scene -> HIP call."""
import numpy as np

from oracle import gs_oracle as O

GS_KEYS = ("means3D", "opacities", "shs", "scales", "rotations")


def oracle_forward(scene, st, dtype=np.float32, nthreads=1, **over):
    kw = dict(shs=scene.get("shs"), colors_precomp=scene.get("colors_precomp"), scales=scene.get("scales"),
              rotations=scene.get("rotations"), cov3D_precomp=scene.get("cov3D_precomp"))
    kw.update(over)
    return O.forward(scene["means3D"], scene["opacities"], st, dtype=dtype, nthreads=nthreads, **kw)


def hip_settings(st, device):
    import torch
    from diff_gaussian_rasterization import GaussianRasterizationSettings
    t = lambda a: torch.as_tensor(np.asarray(a, dtype=np.float32)).to(device)
    return GaussianRasterizationSettings(
        image_height=st["image_height"], image_width=st["image_width"], tanfovx=st["tanfovx"], tanfovy=st["tanfovy"],
        bg=t(st["bg"]), scale_modifier=st.get("scale_modifier", 1.0), viewmatrix=t(st["viewmatrix"]).reshape(4, 4),
        projmatrix=t(st["projmatrix"]).reshape(4, 4), sh_degree=st["sh_degree"], campos=t(st["campos"]),
        prefiltered=False, debug=False)


def hip_forward(scene, st, device="cuda", requires_grad=False):
    """-> (color, radii, depth, alpha, inputs dict of torch tensors, means2D)"""
    import torch
    from diff_gaussian_rasterization import GaussianRasterizer
    inp = {}
    for k in ("means3D", "opacities", "shs", "colors_precomp", "scales", "rotations", "cov3D_precomp"):
        if scene.get(k) is not None:
            inp[k] = torch.tensor(scene[k], dtype=torch.float32, device=device, requires_grad=requires_grad)
    means2D = torch.zeros_like(inp["means3D"], requires_grad=requires_grad)
    rast = GaussianRasterizer(hip_settings(st, device))
    color, radii, depth, alpha = rast(means3D=inp["means3D"], means2D=means2D, opacities=inp["opacities"],
                                      shs=inp.get("shs"), colors_precomp=inp.get("colors_precomp"), scales=inp.get("scales"),
                                      rotations=inp.get("rotations"), cov3D_precomp=inp.get("cov3D_precomp"))
    return color, radii, depth, alpha, inp, means2D


def rel_err(got, ref):
    """max |got-ref| / max |ref|  (the north star's 'relative' for gradients)"""
    ref = np.asarray(ref, dtype=np.float64); got = np.asarray(got, dtype=np.float64)
    d = np.abs(ref).max()
    return float(np.abs(got - ref).max() / (d if d > 0 else 1.0))
