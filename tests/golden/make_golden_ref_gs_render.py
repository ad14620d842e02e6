"""Regenerates tests/golden/ref_gs_render.npz by RUNNING THE REFERENCE'S OWN GaussianSplattingRenderer.render
(/root/reference/MVs_Algorithms/GaussianSplatting/main_3DGS_renderer.py:830-957) on the CPU in this container, on a GaussianModel of the
reference (its activations :226-235, accessors :294-321) seen through the reference's MiniCam.

`diff_gaussian_rasterization` is an un-vendored CUDA wheel, so the module is replaced by tests/fake_dgr.py (the CPU 3DGS oracle behind
the same names, recording what it is handed).  The fixture therefore pins the glue in front of the rasterizer: the settings tuple built
from the camera, which activated tensors are passed under which keyword (sigmoid / exp / normalize, the [N,K,3] feature layout, the
gaussain_idx subset, override_color, the python covariance path), the zero screen-space tensor, the clamp and the result dict.
One function of the third-party `kiui` package that the model's constructor binds is restated: inverse_sigmoid(x) = log(x / (1 - x)).

  python tests/golden/make_golden_ref_gs_render.py [--check]
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), HERE]
import make_golden_ref_py as G  # noqa: E402

OUT = os.path.join(HERE, "ref_gs_render.npz")

CASES = [  # name, kwargs of render(); tensors named by key of the scene dict
    ("default", dict()),
    ("modifier_bg", dict(scaling_modifier=0.7, bg_color="bg")),
    ("subset", dict(gaussain_idx="idx")),
    ("override", dict(override_color="colors")),
    ("cov_python", dict(scaling_modifier=0.8, compute_cov3D_python=True)),
]


def scene():
    rng = np.random.default_rng(31)
    N = 90
    d = {}
    d["xyz"] = (rng.normal(size=(N, 3)) * 0.35).astype(np.float32)
    d["f_dc"] = rng.normal(size=(N, 1, 3)).astype(np.float32)
    d["f_rest"] = (rng.normal(size=(N, 15, 3)) * 0.2).astype(np.float32)
    d["scaling"] = rng.normal(np.log(0.05), 0.4, (N, 3)).astype(np.float32)
    d["rotation"] = rng.normal(size=(N, 4)).astype(np.float32)
    d["opacity"] = rng.normal(0.5, 1.5, (N, 1)).astype(np.float32)
    d["bg"] = np.array([0.2, 0.5, 0.9], np.float32)
    d["idx"] = np.sort(rng.choice(N, 40, replace=False)).astype(np.int64)
    d["colors"] = rng.uniform(0, 1, (N, 3)).astype(np.float32)
    return d


def reference_renderer():
    G._install_stubs()
    G._cpu_redirect()
    sys.modules["kiui.op"].inverse_sigmoid = lambda x: torch.log(x / (1 - x))
    import fake_dgr
    sys.modules["diff_gaussian_rasterization"] = fake_dgr
    _, cam, ren = G.reference_modules()
    return ren, cam, fake_dgr


def flatten_call(prefix, call):
    out = {}
    for k, v in call["settings"].items():
        out[prefix + "set_" + k] = np.asarray(v)
    for k, v in call.items():
        if k != "settings":
            out[prefix + "arg_" + k] = np.zeros((0,), np.float32) if v is None else v
    return out


def generate():
    ren, cam, fake = reference_renderer()
    sc = scene()
    out = {"scene_" + k: v for k, v in sc.items()}
    W, H = 80, 56
    oc = cam.OrbitCamera(W, H, r=2.2, fovy=49.1)
    oc.orbit(700.0, -350.0)
    fovx = cam.calculate_fovX(H, W, oc.fovy)
    pose = oc.pose
    out["pose"], out["fovy"], out["fovx"], out["size"] = pose, np.float64(oc.fovy), np.float64(fovx), np.asarray([W, H])
    P = lambda a: torch.nn.Parameter(torch.from_numpy(a.copy()))
    for name, kw in CASES:
        r = ren.GaussianSplattingRenderer(sh_degree=3, white_background=True, radius=1)
        g = r.gaussians
        g._xyz, g._features_dc, g._features_rest = P(sc["xyz"]), P(sc["f_dc"]), P(sc["f_rest"])
        g._scaling, g._rotation, g._opacity = P(sc["scaling"]), P(sc["rotation"]), P(sc["opacity"])
        g.active_sh_degree = 2                                        # as mid-training: fewer active bands than stored
        mc = cam.MiniCam(pose.copy(), W, H, oc.fovy, fovx, oc.near, oc.far)
        kw = {k: (torch.from_numpy(sc[v].copy()) if isinstance(v, str) else v) for k, v in kw.items()}
        del fake.CALLS[:]
        res = r.render(mc, **kw)
        assert len(fake.CALLS) == 1
        out.update(flatten_call(name + "_", fake.CALLS[0]))
        out[name + "_keys"] = np.asarray(sorted(res.keys()))
        for k, v in res.items():
            out[name + "_out_" + k] = v.detach().numpy()
        out[name + "_viewspace_requires_grad"] = np.asarray(bool(res["viewspace_points"].requires_grad))
    out.update(orbit_case(ren, cam, sc, fake))
    return {k: np.asarray(v) for k, v in out.items()}


ORBIT_POSES = [(2.2, -15.0, 30.0, 0.0, 0.0, 0.0), (2.0, 20.0, 150.0, 0.05, 0.0, 0.0), (2.4, 0.0, -80.0, 0.0, -0.05, 0.1)]


def orbit_case(ren, cam, sc, fake):
    """the orbit loop the renderer nodes use: GaussianSplattingCameraController(...).render_all_pose (main_3DGS.py:76-82 on top of
    BaseCameraController.render_at_pose / render_all_pose, camera_utils.py:240-274), with kiui.cam.orbit_camera restated as in
    make_golden_ref_gs_train.py and the background drawn from numpy's global generator (seed 4)"""
    import make_golden_ref_gs_train as GT
    import types as _t
    sys.modules["kiui.cam"].orbit_camera = GT.orbit_camera
    sys.modules["comfy.utils"].ProgressBar = lambda *a, **k: None
    sys.modules["pytorch_msssim"].MS_SSIM = lambda *a, **k: None
    sys.modules["pytorch_msssim"].SSIM = lambda *a, **k: None
    for name in ("torchvision", "torchvision.transforms", "torchvision.transforms.functional", "PIL", "PIL.Image"):
        if name not in sys.modules:
            try:
                __import__(name)
            except Exception:
                m = G._Stub(name); m.__path__ = []
                sys.modules[name] = m
    G._load("shared_utils.image_utils", "shared_utils/image_utils.py")
    cam.orbit_camera = GT.orbit_camera                                # camera_utils bound the stub's name at import
    pkg = _t.ModuleType("ref_gs_pkg2"); pkg.__path__ = [os.path.join(G.REF, "MVs_Algorithms", "GaussianSplatting")]
    sys.modules["ref_gs_pkg2"] = pkg
    sys.modules["ref_gs_pkg2.main_3DGS_renderer"] = ren
    tr = G._load("ref_gs_pkg2.main_3DGS", "MVs_Algorithms/GaussianSplatting/main_3DGS.py")
    P = lambda a: torch.nn.Parameter(torch.from_numpy(a.copy()))
    r = ren.GaussianSplattingRenderer(sh_degree=3, white_background=True, radius=1)
    g = r.gaussians
    g._xyz, g._features_dc, g._features_rest = P(sc["xyz"]), P(sc["f_dc"]), P(sc["f_rest"])
    g._scaling, g._rotation, g._opacity = P(sc["scaling"]), P(sc["rotation"]), P(sc["opacity"])
    g.active_sh_degree = 3
    ctl = tr.GaussianSplattingCameraController(r, 72, 48, 49.1, 0.5, None, torch.device("cpu"))
    np.random.seed(4)
    del fake.CALLS[:]
    with torch.no_grad():
        images, masks, extra = ctl.render_all_pose(ORBIT_POSES)
    out = {"orbit_images": images.numpy(), "orbit_masks": masks.numpy(), "orbit_extra_keys": np.asarray(sorted(extra.keys()))}
    for k, v in extra.items():
        out["orbit_extra_" + k] = v.numpy()
    out["orbit_bg_per_view"] = np.stack([c["settings"]["bg"] for c in fake.CALLS])
    out["orbit_viewmatrix_per_view"] = np.stack([c["settings"]["viewmatrix"] for c in fake.CALLS])
    return out


def main():
    out = generate()
    if "--check" in sys.argv:
        ref = np.load(OUT)
        bad = [k for k in out if k not in ref.files or out[k].shape != ref[k].shape or not np.array_equal(out[k], ref[k])]
        bad += [k for k in ref.files if k not in out]
        if bad:
            print("MISMATCH:", bad)
            sys.exit(1)
        print("ok: %d arrays identical to the committed fixture" % len(out))
        return
    np.savez_compressed(OUT, **out)
    print("wrote", OUT, "with", len(out), "arrays")


if __name__ == "__main__":
    main()
