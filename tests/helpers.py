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


def grad_report(got, ref):
    """-> dict(rel_l2, frac_viol, worst, max_norm): per-tensor relative L2, and the element-wise test |got - ref| <= 1e-3 |ref| + 1e-6 max|ref|
    (share of elements that violate it, worst ratio error / tolerance)"""
    ref = np.asarray(ref, dtype=np.float64); got = np.asarray(got, dtype=np.float64)
    if ref.size == 0:
        return dict(rel_l2=0.0, frac_viol=0.0, worst=0.0, max_norm=0.0)
    mx = np.abs(ref).max()
    err = np.abs(got - ref)
    tol = 1e-3 * np.abs(ref) + 1e-6 * mx
    nrm = np.linalg.norm(ref.ravel())
    return dict(rel_l2=float(np.linalg.norm(err.ravel()) / (nrm if nrm > 0 else 1.0)), frac_viol=float((err > tol).mean()) if mx > 0 else float((err > 0).mean()),
                worst=float((err / np.maximum(tol, 1e-300)).max()) if mx > 0 else 0.0, max_norm=float(err.max() / (mx if mx > 0 else 1.0)))


def assert_grad_close(got, ref, name="", rel_l2=1e-3, max_frac=0.0, hard=10.0):
    """The round-2 gradient metric: a float32 kernel against the FLOAT64 oracle must satisfy
      * per-tensor relative L2 <= 1e-3,
      * element-wise |got - ref| <= 1e-3 |ref| + 1e-6 max|ref| on all but a share `max_frac` of the elements (0 = every element; a
        float32 sum of cancelling terms can exceed this on isolated small entries, so large scenes pass a small share explicitly),
      * and no element off by more than `hard` times that tolerance."""
    r = grad_report(got, ref)
    msg = "%s: relL2 %.2e, elementwise violations %.2e of elements (worst %.2f x tol), max-norm %.2e" % (name, r["rel_l2"], r["frac_viol"], r["worst"], r["max_norm"])
    print("[grad]", msg)
    assert r["rel_l2"] <= rel_l2, msg
    assert r["frac_viol"] <= max_frac, msg
    assert r["worst"] <= hard, msg
    return r
