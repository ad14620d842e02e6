"""Regenerates tests/golden/ref_py_conventions.npz by RUNNING THE REFERENCE'S OWN PYTHON in this container: the pure helper
functions either side of the 3DGS rasterizer that fix the conventions its inputs follow --

  shared_utils/sh_utils.py        eval_sh (:57-112), RGB2SH / SH2RGB (:114-118)                 SH basis, coefficient order, signs
  shared_utils/camera_utils.py    calculate_fovX (:171), get_projection_matrix (:174-186), MiniCam (:188-214),
                                  OrbitCamera.pose / view / perspective / intrinsics / orbit (:88-169), look_at (:45-62),
                                  compose_orbit_camposes (:276-289)                               camera matrices as the rasterizer receives them
  MVs_Algorithms/GaussianSplatting/main_3DGS_renderer.py
                                  get_expon_lr_func (:21-44), strip_symmetric (:46-58), build_rotation (:81-102),
                                  build_scaling_rotation (:104-113), GaussianModel.covariance_activation (:219-224)
                                                                                                quaternion layout, covariance 6-vector order, lr schedule

The modules are imported from where they lie under /root/reference.  Third-party packages they import at module level and this image
lacks (kiui, plyfile, kornia, torchtyping, pymeshlab, trimesh, cv2, pytorch_msssim) are replaced by inert stubs -- none of the functions
listed above calls into them -- and their hard-coded `device='cuda'` / `.cuda()` are redirected to the CPU.  The outputs are committed;
tests/test_ref_conventions.py holds this repo's mirrors AND the CPU oracle (oracle/gs_oracle.c) to them.

  python tests/golden/make_golden_ref_py.py            # rewrite the fixture
  python tests/golden/make_golden_ref_py.py --check    # regenerate in memory and compare bit for bit with the committed file
"""
import importlib
import importlib.util
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference"
OUT = os.path.join(HERE, "ref_py_conventions.npz")


class _Inert:
    """Stands for anything a stubbed third-party module exports: callable, subscriptable, attribute-able, never evaluated."""

    def __call__(self, *a, **k):
        raise RuntimeError("stubbed third-party symbol was called: the golden generator must only run reference code")

    def __getattr__(self, name):
        return _Inert()

    def __getitem__(self, item):
        return _Inert()

    def __mro_entries__(self, bases):
        return (object,)


class _Stub(types.ModuleType):
    __all__ = []

    def __getattr__(self, name):
        if name.startswith("__"):
            raise AttributeError(name)
        return _Inert()


def _install_stubs():
    for name in ["kiui", "kiui.cam", "kiui.op", "kiui.typing", "kiui.mesh", "plyfile", "kornia", "kornia.geometry", "kornia.geometry.conversions",
                 "torchtyping", "pymeshlab", "trimesh", "cv2", "pytorch_msssim", "comfy", "comfy.utils", "open3d", "xatlas", "pygltflib",
                 "mesh_processer", "mesh_processer.mesh", "mesh_processer.mesh_utils", "diff_gaussian_rasterization", "simple_knn", "simple_knn._C"]:
        if name not in sys.modules:
            try:
                importlib.import_module(name)
            except Exception:
                m = _Stub(name)
                m.__path__ = []
                sys.modules[name] = m
        if "." in name:                                               # `import a.b` then `a.b.x`: the parent must expose the child
            parent, child = name.rsplit(".", 1)
            if isinstance(sys.modules.get(parent), _Stub):
                setattr(sys.modules[parent], child, sys.modules[name])


def _cpu_redirect():
    """the reference hard-codes CUDA placement; run the same arithmetic on the CPU"""
    torch.Tensor.cuda = lambda self, *a, **k: self
    for fn in ("zeros", "ones", "empty", "tensor", "eye", "full", "zeros_like", "ones_like", "empty_like", "arange", "rand", "randn"):
        orig = getattr(torch, fn)

        def wrapped(*a, _orig=orig, **k):
            if str(k.get("device", "")).startswith("cuda"):
                k["device"] = "cpu"
            return _orig(*a, **k)
        setattr(torch, fn, wrapped)


def _load(modname, relpath):
    spec = importlib.util.spec_from_file_location(modname, os.path.join(REF, relpath))
    mod = importlib.util.module_from_spec(spec)
    sys.modules[modname] = mod
    spec.loader.exec_module(mod)
    return mod


def reference_modules():
    assert os.path.isdir(REF), "needs /root/reference"
    _install_stubs()
    _cpu_redirect()
    pkg = types.ModuleType("shared_utils"); pkg.__path__ = [os.path.join(REF, "shared_utils")]
    sys.modules["shared_utils"] = pkg
    sh = _load("shared_utils.sh_utils", "shared_utils/sh_utils.py")
    cam = _load("shared_utils.camera_utils", "shared_utils/camera_utils.py")
    ren = _load("ref_main_3DGS_renderer", "MVs_Algorithms/GaussianSplatting/main_3DGS_renderer.py")
    return sh, cam, ren


def generate():
    sh, cam, ren = reference_modules()
    rng = np.random.default_rng(2024)
    out = {}
    # ---- spherical harmonics: eval_sh(deg, sh [N,3,K], dirs [N,3]) for every degree the rasterizer supports
    N = 64
    coef = rng.normal(size=(N, 3, 16)).astype(np.float32)
    dirs = rng.normal(size=(N, 3)).astype(np.float32)
    dirs /= np.linalg.norm(dirs, axis=1, keepdims=True)
    out["sh_coef"], out["sh_dirs"] = coef, dirs
    for deg in range(4):
        K = (deg + 1) ** 2
        out["sh_eval_deg%d" % deg] = sh.eval_sh(deg, torch.from_numpy(coef[..., :K]), torch.from_numpy(dirs)).numpy()
    rgb = rng.uniform(0, 1, size=(N, 3)).astype(np.float32)
    out["rgb"], out["rgb2sh"], out["sh2rgb"] = rgb, sh.RGB2SH(torch.from_numpy(rgb)).numpy(), sh.SH2RGB(torch.from_numpy(rgb)).numpy()
    # ---- cameras
    cases = [(640, 360, 49.1, 0.01, 100.0), (128, 96, 60.0, 0.1, 50.0), (97, 131, 35.0, 0.5, 10.0)]
    out["cam_cases"] = np.asarray(cases, np.float64)
    for i, (W, H, fovy_deg, zn, zf) in enumerate(cases):
        W, H = int(W), int(H)
        oc = cam.OrbitCamera(W, H, r=2.5, fovy=fovy_deg, near=zn, far=zf)
        oc.orbit(300.0 * (i + 1), -170.0 * (i + 1))
        oc.pan(40.0, -25.0, 10.0)
        oc.scale(0.7)
        pose = oc.pose
        fovx = cam.calculate_fovX(H, W, oc.fovy)
        mc = cam.MiniCam(pose.copy(), W, H, oc.fovy, fovx, zn, zf)
        pre = "cam%d_" % i
        out[pre + "pose"], out[pre + "view"], out[pre + "perspective"] = pose, oc.view, oc.perspective
        out[pre + "mvp"], out[pre + "intrinsics"], out[pre + "campos"] = oc.mvp, oc.intrinsics, oc.campos
        out[pre + "fovx_orbit"], out[pre + "fovx"] = np.float64(oc.fovx), np.float64(fovx)
        out[pre + "projection"] = cam.get_projection_matrix(zn, zf, fovx, oc.fovy).numpy()
        out[pre + "world_view_transform"] = mc.world_view_transform.numpy()
        out[pre + "projection_matrix"] = mc.projection_matrix.numpy()
        out[pre + "full_proj_transform"] = mc.full_proj_transform.numpy()
        out[pre + "camera_center"] = mc.camera_center.numpy()
    # ---- points inside camera 0's frustum (sampled in its view space, carried to the world with the reference's own matrix), the view
    #      directions from the reference's camera_center, and the reference's SH colours for them: what stage A7 of the rasterizer must give
    wv = torch.from_numpy(out["cam0_world_view_transform"]).double()
    pv = np.concatenate([rng.uniform(-0.45, 0.45, (N, 2)), rng.uniform(1.5, 4.0, (N, 1)), np.ones((N, 1))], 1)
    pv[:, :2] *= pv[:, 2:3] * 0.6
    world = (torch.from_numpy(pv) @ torch.linalg.inv(wv))[:, :3].float()
    vdir = world - torch.from_numpy(out["cam0_camera_center"]).float()[None]
    vdir = vdir / vdir.norm(dim=1, keepdim=True)
    out["frustum_points"], out["frustum_dirs"] = world.numpy(), vdir.numpy()
    for deg in range(4):
        K = (deg + 1) ** 2
        out["frustum_sh_eval_deg%d" % deg] = sh.eval_sh(deg, torch.from_numpy(coef[..., :K]), vdir).numpy()
    campos = torch.tensor([[1.0, 2.0, -3.0], [0.2, -0.5, 2.0]]); target = torch.tensor([[0.1, 0.0, 0.3], [0.0, 0.0, 0.0]])
    try:
        out["look_at_opengl"] = np.stack([np.asarray(cam.look_at(campos[k].numpy(), target[k].numpy(), True)) for k in range(2)])
        out["look_at_colmap"] = np.stack([np.asarray(cam.look_at(campos[k].numpy(), target[k].numpy(), False)) for k in range(2)])
    except Exception as e:        # signature differences are not worth failing the whole fixture
        print("look_at skipped:", e)
    out["orbit_camposes"] = np.asarray(cam.compose_orbit_camposes([2.0, 1.5, 3.0], [-100.0, 10.0, 95.0], [-200.0, 30.0, 190.0], [0.0, 0.1, 0.2],
                                                                  [0.0, -0.1, 0.3], [0.5, 0.0, -0.5]), np.float64)
    # ---- Gaussian model helpers
    M = 48
    q = rng.normal(size=(M, 4)).astype(np.float32)                    # deliberately not normalised: build_rotation normalises
    s = np.exp(rng.normal(-2.5, 0.6, size=(M, 3))).astype(np.float32)
    out["quat"], out["scale"] = q, s
    out["rotation"] = ren.build_rotation(torch.from_numpy(q)).numpy()
    out["scaling_rotation"] = ren.build_scaling_rotation(torch.from_numpy(s), torch.from_numpy(q)).numpy()
    gm = ren.GaussianModel.__new__(ren.GaussianModel)
    gm.setup_functions()
    for mod in (1.0, 0.6):
        out["covariance_mod%g" % mod] = gm.covariance_activation(torch.from_numpy(s), mod, torch.from_numpy(q)).numpy()
    steps = np.array([-1, 0, 1, 10, 100, 499, 500, 1000, 5000, 30000, 40000], np.int64)
    out["lr_steps"] = steps
    f1 = ren.get_expon_lr_func(lr_init=1.6e-4, lr_final=1.6e-6, lr_delay_mult=0.01, max_steps=30000)
    f2 = ren.get_expon_lr_func(lr_init=1e-3, lr_final=1e-5, lr_delay_steps=500, lr_delay_mult=0.1, max_steps=5000)
    f3 = ren.get_expon_lr_func(lr_init=2e-3, lr_final=2e-3)
    out["lr_default"] = np.array([f1(int(t)) for t in steps], np.float64)
    out["lr_delayed"] = np.array([f2(int(t)) for t in steps], np.float64)
    out["lr_constant"] = np.array([f3(int(t)) for t in steps], np.float64)
    out.update(densify_case(ren, rng))
    # ---- degree 4 of eval_sh (the reference's host function goes to 4, sh_utils.py:101-111; the rasterizer stops at 3).  Its own generator: the entries above keep
    # their values
    rng4 = np.random.default_rng(2025)
    coef4 = rng4.normal(size=(48, 3, 25)).astype(np.float32)
    dirs4 = rng4.normal(size=(48, 3)).astype(np.float32)
    dirs4 /= np.linalg.norm(dirs4, axis=1, keepdims=True)
    out["sh4_coef"], out["sh4_dirs"] = coef4, dirs4
    out["sh_eval_deg4"] = sh.eval_sh(4, torch.from_numpy(coef4), torch.from_numpy(dirs4)).numpy()
    return {k: np.asarray(v) for k, v in out.items()}


DENSIFY_ARGS = dict(max_grad=2e-4, min_opacity=0.02, extent=2.0, max_screen_size=20)


def densify_inputs(rng, N=300):
    """a small cloud in the state a trainer would hand to densify_and_prune: mixed scales either side of percent_dense * extent,
    some transparent points, some oversized ones, gradient statistics with never-seen points (denominator 0)"""
    d = {}
    d["xyz"] = rng.uniform(-1, 1, (N, 3)).astype(np.float32)
    d["f_dc"] = rng.normal(size=(N, 1, 3)).astype(np.float32)
    d["f_rest"] = (rng.normal(size=(N, 15, 3)) * 0.1).astype(np.float32)
    d["scaling"] = rng.normal(np.log(0.02), 0.9, (N, 3)).astype(np.float32)          # raw = log
    d["rotation"] = rng.normal(size=(N, 4)).astype(np.float32)
    d["opacity"] = rng.normal(0.0, 2.5, (N, 1)).astype(np.float32)                   # raw = logit
    d["max_radii2D"] = rng.integers(0, 60, N).astype(np.float32)
    d["denom"] = rng.integers(0, 4, (N, 1)).astype(np.float32)
    d["xyz_gradient_accum"] = (rng.exponential(2.5e-4, (N, 1)) * d["denom"]).astype(np.float32)
    d["step_grads"] = [rng.normal(size=d[k].shape).astype(np.float32) for k in ("xyz", "f_dc", "f_rest", "opacity", "scaling", "rotation")]
    return d


def densify_case(ren, rng):
    """GaussianModel.densify_and_prune (:759-761 = clone :670-687, split :641-668, prune :771-781 + the optimizer surgery :558-614) of
    the reference itself, on the CPU, with torch.manual_seed(0) for the split samples"""
    from types import SimpleNamespace
    d = densify_inputs(rng)
    gm = ren.GaussianModel(3)
    P = lambda a: torch.nn.Parameter(torch.from_numpy(a.copy()).requires_grad_(True))
    gm._xyz, gm._features_dc, gm._features_rest = P(d["xyz"]), P(d["f_dc"]), P(d["f_rest"])
    gm._scaling, gm._rotation, gm._opacity = P(d["scaling"]), P(d["rotation"]), P(d["opacity"])
    gm.init_xyz = torch.from_numpy(d["xyz"].copy())
    gm.spatial_lr_scale = 1.0
    gm.training_setup(SimpleNamespace(percent_dense=0.01, position_lr_init=1.6e-4, position_lr_final=1.6e-6, position_lr_delay_mult=0.01,
                                      position_lr_max_steps=30000, feature_lr=2.5e-3, opacity_lr=0.05, scaling_lr=5e-3, rotation_lr=1e-3))
    for prm, gr in zip((gm._xyz, gm._features_dc, gm._features_rest, gm._opacity, gm._scaling, gm._rotation), d["step_grads"]):
        prm.grad = torch.from_numpy(gr.copy())
    gm.optimizer.step()                                                              # Adam moments exist, as mid-training
    gm.optimizer.zero_grad(set_to_none=True)
    gm.max_radii2D = torch.from_numpy(d["max_radii2D"].copy())
    gm.xyz_gradient_accum = torch.from_numpy(d["xyz_gradient_accum"].copy())
    gm.denom = torch.from_numpy(d["denom"].copy())
    before = {k: getattr(gm, "_" + k).detach().numpy().copy() for k in ("xyz", "features_dc", "features_rest", "scaling", "rotation", "opacity")}
    torch.manual_seed(0)
    gm.densify_and_prune(**DENSIFY_ARGS)
    out = {"dens_in_" + k: v for k, v in d.items() if k != "step_grads"}
    for i, gr in enumerate(d["step_grads"]):
        out["dens_in_step_grad%d" % i] = gr
    for k, v in before.items():
        out["dens_stepped_" + k] = v
    for k in ("xyz", "features_dc", "features_rest", "scaling", "rotation", "opacity"):
        out["dens_out_" + k] = getattr(gm, "_" + k).detach().numpy()
    out["dens_out_init_xyz"] = gm.init_xyz.numpy()
    out["dens_out_max_radii2D"] = gm.max_radii2D.numpy()
    out["dens_out_xyz_gradient_accum"], out["dens_out_denom"] = gm.xyz_gradient_accum.numpy(), gm.denom.numpy()
    for grp in gm.optimizer.param_groups:
        st = gm.optimizer.state[grp["params"][0]]
        out["dens_out_exp_avg_" + grp["name"]] = st["exp_avg"].numpy()
        out["dens_out_exp_avg_sq_" + grp["name"]] = st["exp_avg_sq"].numpy()
        assert grp["params"][0] is getattr(gm, {"xyz": "_xyz", "f_dc": "_features_dc", "f_rest": "_features_rest", "opacity": "_opacity",
                                                 "scaling": "_scaling", "rotation": "_rotation"}[grp["name"]])
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
