"""Holds this repo's host-side mirrors AND the CPU oracle of the 3DGS rasterizer to outputs of the REFERENCE'S OWN PYTHON.

tests/golden/ref_py_conventions.npz was produced by importing the reference's modules in the build container and running their pure
helper functions (tests/golden/make_golden_ref_py.py lists them with file:line).  They fix every convention the rasterizer's inputs
follow: SH basis / coefficient order (eval_sh), the camera matrices exactly as the rasterizer receives them (MiniCam: transposed,
row-vector, "rectified" w2c, camera_center), quaternion layout and the 6-vector covariance order (build_rotation,
covariance_activation), the learning-rate schedule -- and one full run of the reference's GaussianModel.densify_and_prune with its optimizer
surgery, which the one-pass formulation of this repo must reproduce point for point.  What stays unpinned is the un-vendored CUDA kernel itself: culling constants, the
EWA projection, tile binning and the blending loop (oracle/gs_oracle.c header).
"""
import math
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from oracle import gs_oracle as O

GOLD_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.fixture(scope="module")
def g():
    z = np.load(os.path.join(GOLD_DIR, "ref_py_conventions.npz"))
    return {k: z[k] for k in z.files}


def test_fixture_is_what_the_reference_produces_now():
    if not os.path.isdir("/root/reference/shared_utils"):
        pytest.skip("/root/reference is not mounted here")
    r = subprocess.run([sys.executable, os.path.join(GOLD_DIR, "make_golden_ref_py.py"), "--check"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr


# ------------------------------------------------------------------------------------------------ the mirrors
def test_sh_helpers_match_reference(g):
    from shared_utils import sh_utils as SH
    coef, dirs = torch.from_numpy(g["sh_coef"]), torch.from_numpy(g["sh_dirs"])
    for deg in range(4):
        K = (deg + 1) ** 2
        got = SH.eval_sh(deg, coef[..., :K], dirs).numpy()
        assert np.abs(got - g["sh_eval_deg%d" % deg]).max() <= 2e-6, deg
        assert np.abs(SH.eval_sh(deg, coef, dirs).numpy() - g["sh_eval_deg%d" % deg]).max() <= 2e-6          # surplus coefficients are ignored
    # degree 4: the reference's host function goes that far (sh_utils.py:101-111); the rasterizer kernels stop at 3
    got4 = SH.eval_sh(4, torch.from_numpy(g["sh4_coef"]), torch.from_numpy(g["sh4_dirs"])).numpy()
    assert np.abs(got4 - g["sh_eval_deg4"]).max() <= 5e-6
    np.testing.assert_allclose(SH.RGB2SH(torch.from_numpy(g["rgb"])).numpy(), g["rgb2sh"], atol=1e-6)
    np.testing.assert_allclose(SH.SH2RGB(torch.from_numpy(g["rgb"])).numpy(), g["sh2rgb"], atol=1e-6)
    with pytest.raises(ValueError):
        SH.eval_sh(2, coef[..., :4], dirs)


def test_cameras_match_reference(g):
    from shared_utils import camera_utils as CU
    for i, (W, H, fovy_deg, zn, zf) in enumerate(g["cam_cases"]):
        W, H = int(W), int(H)
        pre = "cam%d_" % i
        oc = CU.OrbitCamera(W, H, r=2.5, fovy=fovy_deg, near=zn, far=zf)
        oc.orbit(300.0 * (i + 1), -170.0 * (i + 1))
        oc.pan(40.0, -25.0, 10.0)
        oc.scale(0.7)
        np.testing.assert_allclose(oc.pose, g[pre + "pose"], atol=2e-6)
        np.testing.assert_allclose(oc.view, g[pre + "view"], atol=5e-6)
        np.testing.assert_allclose(oc.perspective, g[pre + "perspective"], rtol=1e-6, atol=1e-7)
        np.testing.assert_allclose(oc.mvp, g[pre + "mvp"], rtol=1e-5, atol=1e-5)
        np.testing.assert_allclose(oc.intrinsics, g[pre + "intrinsics"], rtol=1e-6)
        np.testing.assert_allclose(oc.campos, g[pre + "campos"], atol=2e-6)
        assert abs(oc.fovx - float(g[pre + "fovx_orbit"])) <= 1e-12
        fovx = CU.calculate_fovX(H, W, oc.fovy)
        assert abs(fovx - float(g[pre + "fovx"])) <= 1e-12
        np.testing.assert_allclose(CU.get_projection_matrix(zn, zf, fovx, oc.fovy).numpy(), g[pre + "projection"], rtol=1e-6, atol=1e-7)
        mc = CU.MiniCam(g[pre + "pose"].copy(), W, H, oc.fovy, fovx, zn, zf, device="cpu")             # the reference's pose in: isolates MiniCam
        np.testing.assert_allclose(mc.world_view_transform.numpy(), g[pre + "world_view_transform"], atol=1e-6)
        np.testing.assert_allclose(mc.projection_matrix.numpy(), g[pre + "projection_matrix"], rtol=1e-6, atol=1e-7)
        np.testing.assert_allclose(mc.full_proj_transform.numpy(), g[pre + "full_proj_transform"], rtol=1e-5, atol=1e-6)
        np.testing.assert_allclose(mc.camera_center.numpy(), g[pre + "camera_center"], atol=1e-7)
    campos = np.array([[1.0, 2.0, -3.0], [0.2, -0.5, 2.0]], np.float32); target = np.array([[0.1, 0.0, 0.3], [0.0, 0.0, 0.0]], np.float32)
    for k in range(2):
        np.testing.assert_allclose(CU.look_at(campos[k], target[k], True), g["look_at_opengl"][k], atol=1e-6)
        np.testing.assert_allclose(CU.look_at(campos[k], target[k], False), g["look_at_colmap"][k], atol=1e-6)
    got = CU.compose_orbit_camposes([2.0, 1.5, 3.0], [-100.0, 10.0, 95.0], [-200.0, 30.0, 190.0], [0.0, 0.1, 0.2], [0.0, -0.1, 0.3], [0.5, 0.0, -0.5])
    np.testing.assert_allclose(np.asarray(got, np.float64), g["orbit_camposes"], atol=1e-12)


def test_gaussian_model_helpers_match_reference(g):
    from MVs_Algorithms.GaussianSplatting import main_3DGS_renderer as R
    q, s = torch.from_numpy(g["quat"]), torch.from_numpy(g["scale"])
    np.testing.assert_allclose(R.build_rotation(q).numpy(), g["rotation"], atol=1e-6)
    np.testing.assert_allclose(R.build_scaling_rotation(s, q).numpy(), g["scaling_rotation"], atol=1e-7)
    for mod in (1.0, 0.6):
        cov = R.build_covariance_from_scaling_rotation(s, mod, q).numpy()
        np.testing.assert_allclose(cov, g["covariance_mod%g" % mod], rtol=1e-5, atol=1e-9)
    gm = R.GaussianModel(3, device="cpu")
    gm._scaling, gm._rotation = torch.log(s), q
    np.testing.assert_allclose(gm.get_covariance(0.6).numpy(), g["covariance_mod0.6"], rtol=1e-5, atol=1e-9)
    np.testing.assert_allclose(gm.get_covariance(1, torch.arange(5)).numpy(), g["covariance_mod1"][:5], rtol=1e-5, atol=1e-9)
    steps = g["lr_steps"]
    f1 = R.get_expon_lr_func(lr_init=1.6e-4, lr_final=1.6e-6, lr_delay_mult=0.01, max_steps=30000)
    f2 = R.get_expon_lr_func(lr_init=1e-3, lr_final=1e-5, lr_delay_steps=500, lr_delay_mult=0.1, max_steps=5000)
    f3 = R.get_expon_lr_func(lr_init=2e-3, lr_final=2e-3)
    np.testing.assert_allclose([f1(int(t)) for t in steps], g["lr_default"], rtol=1e-12)
    np.testing.assert_allclose([f2(int(t)) for t in steps], g["lr_delayed"], rtol=1e-12)
    np.testing.assert_allclose([f3(int(t)) for t in steps], g["lr_constant"], rtol=1e-12)


# ------------------------------------------------------------------------------------------------ the oracle
def _settings(g, deg=3, mod=1.0):
    W, H, fovy_deg, zn, zf = g["cam_cases"][0]
    fovy = math.radians(fovy_deg)
    return {"image_height": int(H), "image_width": int(W), "tanfovx": math.tan(float(g["cam0_fovx"]) * 0.5), "tanfovy": math.tan(fovy * 0.5),
            "bg": np.zeros(3, np.float32), "scale_modifier": mod, "viewmatrix": g["cam0_world_view_transform"].astype(np.float32),
            "projmatrix": g["cam0_full_proj_transform"].astype(np.float32), "sh_degree": deg, "campos": g["cam0_camera_center"].astype(np.float32),
            "prefiltered": False, "debug": False}


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_oracle_stage_outputs_follow_the_reference_conventions(g, oracle_built, dtype):
    """Per-Gaussian outputs of the oracle's preprocess stage against values computed from the reference's own functions: SH colour
    (eval_sh with the direction from the reference's camera_center), view depth and pixel position through the reference's matrices."""
    pts = g["frustum_points"]
    N = pts.shape[0]
    shs = np.ascontiguousarray(np.transpose(g["sh_coef"], (0, 2, 1)))           # reference [N,3,K] -> rasterizer [N,K,3]
    scales = np.full((N, 3), 0.01, np.float32); rots = np.tile(np.array([1, 0, 0, 0], np.float32), (N, 1)); opac = np.full((N, 1), 0.5, np.float32)
    tol = 5e-6 if dtype == np.float32 else 2e-6                                 # the reference values themselves are float32
    for deg in range(4):
        st = _settings(g, deg)
        _, radii, _, _, state = O.forward(pts, opac, st, shs=shs, scales=scales, rotations=rots, dtype=dtype)
        assert (radii > 0).all()                                                # every sampled point is inside the frustum
        geo = state.geometry()
        want = np.maximum(g["frustum_sh_eval_deg%d" % deg] + 0.5, 0.0)
        assert np.abs(geo["rgb"] - want).max() <= tol, deg
    ph = np.concatenate([pts.astype(np.float64), np.ones((N, 1))], 1)
    pview = ph @ g["cam0_world_view_transform"].astype(np.float64)
    np.testing.assert_allclose(geo["depths"], pview[:, 2], rtol=2e-6)
    pp = ph @ g["cam0_full_proj_transform"].astype(np.float64)
    ndc = pp[:, :2] / (pp[:, 3:4] + 1e-7)
    W, H = st["image_width"], st["image_height"]
    np.testing.assert_allclose(geo["xy"], ((ndc + 1.0) * np.array([W, H]) - 1.0) * 0.5, rtol=1e-5, atol=2e-4)


@pytest.mark.parametrize("mod", [1.0, 0.6])
def test_oracle_covariance_follows_the_reference_quaternion_and_6vector_layout(g, oracle_built, mod):
    """scales + rotations through the oracle's own covariance stage == the reference's covariance_activation output fed as cov3D_precomp"""
    M = g["quat"].shape[0]
    pts = g["frustum_points"][:M]
    q = g["quat"] / np.linalg.norm(g["quat"], axis=1, keepdims=True)            # the renderer hands the rasterizer normalised quaternions
    colors = np.full((M, 3), 0.5, np.float32); opac = np.full((M, 1), 0.7, np.float32)
    st = _settings(g, 0, mod)
    d = np.float64
    _, ra, _, _, sa = O.forward(pts, opac, st, colors_precomp=colors, scales=g["scale"], rotations=q, dtype=d)
    st1 = dict(st, scale_modifier=1.0)                                          # a precomputed covariance already carries the modifier
    _, rb, _, _, sb = O.forward(pts, opac, st1, colors_precomp=colors, cov3D_precomp=g["covariance_mod%g" % mod], dtype=d)
    ga, gb = sa.geometry(), sb.geometry()
    assert (ra > 0).all()
    np.testing.assert_allclose(ga["conic_opacity"], gb["conic_opacity"], rtol=2e-4)
    np.testing.assert_allclose(ga["xy"], gb["xy"], atol=1e-9)
    assert np.abs(ra - rb).max() <= 1


# ------------------------------------------------------------------------------------------------ densify / prune
def test_densify_and_prune_matches_a_run_of_the_reference(g):
    """The fixture holds a run of the reference's OWN GaussianModel.densify_and_prune (clone -> split -> prune with its optimizer
    surgery, main_3DGS_renderer.py:558-781) on the CPU.  The one-pass formulation here must reproduce it: same points in the same order,
    same Adam moments, the same split samples (one normal draw of shape [2S,3] from a generator seeded like torch.manual_seed(0))."""
    sys.path.insert(0, GOLD_DIR)
    from make_golden_ref_py import DENSIFY_ARGS
    from MVs_Algorithms.GaussianSplatting.main_3DGS_renderer import GaussianModel
    from MVs_Algorithms.GaussianSplatting.main_3DGS import GSParams
    T = lambda k: torch.from_numpy(g[k].copy())
    gm = GaussianModel(3, device="cpu")
    gm.create_from_tensors(T("dens_in_xyz"), torch.cat((T("dens_in_f_dc"), T("dens_in_f_rest")), dim=1), T("dens_in_scaling"), T("dens_in_rotation"),
                           T("dens_in_opacity"), spatial_lr_scale=1.0)
    gm.training_setup(GSParams())
    P = gm._param_dict()
    for i, name in enumerate(("xyz", "f_dc", "f_rest", "opacity", "scaling", "rotation")):
        P[name].grad = T("dens_in_step_grad%d" % i)
    gm.optimizer.step()                                               # same Adam configuration -> same parameters after one step
    long = {"xyz": "xyz", "f_dc": "features_dc", "f_rest": "features_rest", "opacity": "opacity", "scaling": "scaling", "rotation": "rotation"}
    for name, t in gm._param_dict().items():
        np.testing.assert_allclose(t.detach().numpy(), g["dens_stepped_" + long[name]], rtol=1e-6, atol=1e-7, err_msg=name)
        t.grad = None
    gm.max_radii2D = T("dens_in_max_radii2D")
    gm.xyz_gradient_accum, gm.denom = T("dens_in_xyz_gradient_accum"), T("dens_in_denom")
    info = gm.densify_and_prune(generator=torch.Generator().manual_seed(0), **DENSIFY_ARGS)
    n = g["dens_out_xyz"].shape[0]
    assert info["points"] == n and info["cloned"] > 10 and info["split"] > 10 and info["pruned"] > 10
    for name, t in gm._param_dict().items():
        assert t.shape[0] == n and t.requires_grad and t.is_leaf
        np.testing.assert_allclose(t.detach().numpy(), g["dens_out_" + long[name]], rtol=1e-6, atol=1e-7, err_msg=name)
    for grp in gm.optimizer.param_groups:
        st = gm.optimizer.state[grp["params"][0]]
        np.testing.assert_allclose(st["exp_avg"].numpy(), g["dens_out_exp_avg_" + grp["name"]], rtol=1e-6, atol=1e-12, err_msg=grp["name"])
        np.testing.assert_allclose(st["exp_avg_sq"].numpy(), g["dens_out_exp_avg_sq_" + grp["name"]], rtol=1e-6, atol=1e-15, err_msg=grp["name"])
    np.testing.assert_allclose(gm.init_xyz.numpy(), g["dens_out_init_xyz"], atol=0)
    assert np.array_equal(gm.max_radii2D.numpy(), g["dens_out_max_radii2D"])          # all zero: the reference's screen-size criterion never fires
    assert np.array_equal(gm.xyz_gradient_accum.numpy(), g["dens_out_xyz_gradient_accum"]) and np.array_equal(gm.denom.numpy(), g["dens_out_denom"])
