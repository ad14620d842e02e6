"""CPU tests that pin the oracle (oracle/gs_oracle.c): analytic known answers, invariants, agreement with
the independent torch-autograd restatement, and the committed golden vectors.  No GPU."""
import math
import os

import numpy as np
import pytest
import torch

from c3d_hip import synthetic as S
from oracle import gs_oracle as O, gs_torch_ref as R
from helpers import oracle_forward

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def _identity_cam(W, H, tanfov=0.5, bg=(0, 0, 0), deg=0):
    """camera at the origin looking down +z (view = identity), simple pinhole projection"""
    view = np.eye(4, dtype=np.float32)
    P = np.zeros((4, 4), dtype=np.float32)
    P[0, 0] = 1 / (tanfov * W / H); P[1, 1] = 1 / tanfov; P[2, 2] = 1.0; P[2, 3] = -0.01; P[3, 2] = 1.0
    return {"image_height": H, "image_width": W, "tanfovx": tanfov * W / H, "tanfovy": tanfov, "bg": np.array(bg, np.float32),
            "scale_modifier": 1.0, "viewmatrix": view.T.copy(), "projmatrix": (view.T @ P.T).copy(), "sh_degree": deg,
            "campos": np.zeros(3, np.float32)}


def test_single_isotropic_gaussian_closed_form(oracle_built):
    """One isotropic Gaussian on the optical axis: alpha(centre pixel) and the falloff have closed forms."""
    W = H = 33
    st = _identity_cam(W, H)
    z, s, o = 2.0, 0.0625, 0.75   # exactly representable in float32
    col = np.array([[0.25, 0.5, 0.875]], np.float32)
    for dt, tol in ((np.float32, 2e-6), (np.float64, 1e-12)):
        color, radii, depth, alpha, state = O.forward(np.array([[0, 0, z]], np.float32), np.array([[o]], np.float32), st,
                                                      colors_precomp=col, scales=np.full((1, 3), s, np.float32),
                                                      rotations=np.array([[1, 0, 0, 0]], np.float32), dtype=dt)
        f = H / (2 * 0.5)
        var = (f * s / z) ** 2 + 0.3   # EWA: J Sigma J^T + 0.3 px^2
        cx = ((0 + 1) * W - 1) / 2     # ndc 0 -> pixel 16.0
        assert radii[0] == math.ceil(3 * math.sqrt(var + math.sqrt(0.1)))   # lambda_max = mid + sqrt(max(0.1, mid^2 - det))
        for (px, py) in [(16, 16), (18, 16), (16, 13), (20, 21)]:
            d2 = (px - cx) ** 2 + (py - cx) ** 2
            a = min(0.99, o * math.exp(-0.5 * d2 / var))
            a = a if a >= 1 / 255 else 0.0
            assert abs(alpha[0, py, px] - a) < tol * 10
            assert abs(depth[0, py, px] - a * z) < tol * 10
            np.testing.assert_allclose(color[:, py, px], a * col[0], atol=tol * 10)


def test_two_overlapping_splats_closed_form(oracle_built):
    W = H = 17
    st = _identity_cam(W, H, bg=(1, 1, 1))
    m = np.array([[0, 0, 3.0], [0, 0, 2.0]], np.float32)       # second one is nearer
    o = np.array([[0.625], [0.5]], np.float32)
    col = np.array([[1, 0, 0], [0, 1, 0]], np.float32)
    color, radii, depth, alpha, _ = O.forward(m, o, st, colors_precomp=col, scales=np.full((2, 3), 0.2, np.float32),
                                              rotations=np.tile(np.array([[1, 0, 0, 0]], np.float32), (2, 1)), dtype=np.float64)
    a_near, a_far = 0.5, 0.625    # at the exact centre pixel (8,8): exp(0) = 1
    T = (1 - a_near) * (1 - a_far)
    np.testing.assert_allclose(color[:, 8, 8], [a_far * (1 - a_near) + T, a_near + T, T], atol=1e-12)
    assert abs(alpha[0, 8, 8] - (1 - T)) < 1e-12
    assert abs(depth[0, 8, 8] - (a_near * 2.0 + (1 - a_near) * a_far * 3.0)) < 1e-12


@pytest.mark.parametrize("seed", [1, 2])
def test_invariants(oracle_built, seed):
    sc = S.make_small_scene(N=64, seed=seed)
    st = S.camera_settings(40, 24, 49.1, 15, 70, 2.0, bg=(0.2, 0.4, 0.6))
    color, radii, depth, alpha, state = oracle_forward(sc, st, dtype=np.float64)
    img = state.image_state()
    # alpha = 1 - T_final
    np.testing.assert_allclose(alpha.reshape(-1), 1 - img["final_T"], atol=1e-12)
    # image is affine in bg
    st0 = dict(st, bg=np.zeros(3, np.float32))
    color0, *_ = oracle_forward(sc, st0, dtype=np.float64)
    np.testing.assert_allclose(color - color0, img["final_T"].reshape(1, 24, 40) * np.asarray(st["bg"], np.float64).reshape(3, 1, 1), atol=1e-12)
    # radii > 0 <=> tiles_touched > 0
    assert ((radii > 0) == (state.geometry()["tiles_touched"] > 0)).all()
    # permutation of the inputs changes nothing (depth ties do not occur in this scene)
    perm = np.random.default_rng(0).permutation(64)
    scp = {k: v[perm] for k, v in sc.items()}
    colorp, radiip, depthp, alphap, _ = oracle_forward(scp, st, dtype=np.float64)
    np.testing.assert_allclose(colorp, color, atol=1e-12)
    assert (radiip == radii[perm]).all()
    # per-tile lists are sorted by (depth, id)
    b, g = state.binning(), state.geometry()
    for (s, e) in b["ranges"]:
        ids = b["point_list"][s:e]
        d = g["depths"][ids]
        assert (np.diff(d) >= 0).all()


def test_empty_and_culled(oracle_built):
    st = S.camera_settings(20, 12, 49.1, 0, 0, 2.0, bg=(0.1, 0.2, 0.3))
    # N = 0
    color, radii, depth, alpha, state = O.forward(np.zeros((0, 3), np.float32), np.zeros((0, 1), np.float32), st,
                                                  shs=np.zeros((0, 16, 3), np.float32), scales=np.zeros((0, 3), np.float32),
                                                  rotations=np.zeros((0, 4), np.float32))
    np.testing.assert_allclose(color, np.broadcast_to(np.array([0.1, 0.2, 0.3], np.float32).reshape(3, 1, 1), color.shape))
    assert state.num_rendered == 0 and alpha.max() == 0
    # everything behind the camera
    sc = S.make_small_scene(N=10)
    sc["means3D"] = sc["means3D"] + np.array([0, 0, 10], np.float32)   # camera sits at z=+2 looking at -z
    color, radii, *_ = oracle_forward(sc, st)
    assert (radii == 0).all()
    # exactly-one-of rules keep the dependency's messages
    with pytest.raises(Exception, match="excatly one of either SHs"):
        O.forward(sc["means3D"], sc["opacities"], st, scales=sc["scales"], rotations=sc["rotations"])
    with pytest.raises(Exception, match="scale/rotation pair or precomputed 3D covariance"):
        O.forward(sc["means3D"], sc["opacities"], st, shs=sc["shs"])


def test_f32_vs_f64(oracle_built):
    sc = S.make_small_scene(N=200, seed=5, scale=0.05)
    st = S.camera_settings(64, 48, 49.1, -10, 120, 2.0)
    c32, r32, d32, a32, _ = oracle_forward(sc, st, dtype=np.float32)
    c64, r64, d64, a64, _ = oracle_forward(sc, st, dtype=np.float64)
    assert np.abs(c32 - c64).mean() < 1e-5
    assert (r32 != r64).sum() <= 1


@pytest.mark.parametrize("case", [(40, 48, 32, -20, 30, 2.0, 7, 3), (60, 40, 40, 35, 200, 1.6, 3, 2), (30, 33, 17, 0, 0, 1.2, 11, 0)])
def test_matches_torch_autograd(oracle_built, case):
    """Forward values and every gradient of the C oracle (float64) against the independent torch restatement."""
    N, W, H, el, az, rad, seed, deg = case
    sc = S.make_small_scene(N=N, seed=seed, sh_degree=3)
    st = S.camera_settings(W, H, 49.1, el, az, rad, bg=(0.3, 0.7, 0.1), sh_degree=deg)
    rng = np.random.default_rng(5)
    gC, gD, gA = rng.normal(size=(3, H, W)), rng.normal(size=(1, H, W)), rng.normal(size=(1, H, W))
    color, radii, depth, alpha, state = oracle_forward(sc, st, dtype=np.float64)
    g = O.backward(state, gC, gD, gA)
    td = lambda a: torch.tensor(a, dtype=torch.float64, requires_grad=True)
    m, o, sh, s, r = (td(sc[k]) for k in ("means3D", "opacities", "shs", "scales", "rotations"))
    c2, r2, d2, a2 = R.render(m, o, st, shs=sh, scales=s, rotations=r)
    np.testing.assert_allclose(color, c2.detach().numpy(), atol=1e-12)
    np.testing.assert_allclose(depth, d2.detach().numpy(), atol=1e-12)
    np.testing.assert_allclose(alpha, a2.detach().numpy(), atol=1e-12)
    assert (radii == r2.numpy()).all()
    ((c2 * torch.tensor(gC)).sum() + (d2 * torch.tensor(gD)).sum() + (a2 * torch.tensor(gA)).sum()).backward()
    for name, t in (("means3D", m), ("opacities", o), ("shs", sh), ("scales", s), ("rotations", r)):
        ref = t.grad.numpy()
        err = np.abs(ref - g[name]).max() / max(np.abs(ref).max(), 1e-30)
        assert err < 1e-6, (name, err)    # residual = the dependency's 1e-7 epsilon in d(conic)/d(cov2D)
    if deg < 3:   # coefficients above the active degree get no gradient
        assert np.abs(g["shs"][:, (deg + 1) ** 2:]).max() == 0


def test_scale_modifier_gradient_conventions(oracle_built):
    """scale_modifier != 1: the oracle's default dL/dscale is what the dependency's backward returns -- the derivative w.r.t.
    (modifier * scale), i.e. the exact one divided by the modifier -- and set_exact_dscale(True) gives the exact derivative, which is what
    torch autograd computes for the independent restatement.  Every other gradient is the same under both conventions."""
    sc = S.make_small_scene(N=40, seed=7, sh_degree=3)
    st = S.camera_settings(48, 32, 49.1, -20, 30, 2.0, bg=(0.3, 0.7, 0.1), sh_degree=3)
    st["scale_modifier"] = 0.7
    gC = np.random.default_rng(5).normal(size=(3, 32, 48))
    td = lambda a: torch.tensor(a, dtype=torch.float64, requires_grad=True)
    m, o, sh, s, r = (td(sc[k]) for k in ("means3D", "opacities", "shs", "scales", "rotations"))
    c2, *_ = R.render(m, o, st, shs=sh, scales=s, rotations=r)
    (c2 * torch.tensor(gC)).sum().backward()
    assert O.set_exact_dscale(False) is False                    # the default is the dependency's convention
    try:
        color, radii, depth, alpha, state = oracle_forward(sc, st, dtype=np.float64)
        np.testing.assert_allclose(color, c2.detach().numpy(), atol=1e-12)
        g_wheel = O.backward(state, gC)
        O.set_exact_dscale(True)
        g_exact = O.backward(state, gC)
    finally:
        O.set_exact_dscale(False)
    rel = lambda a, b: np.abs(a - b).max() / max(np.abs(b).max(), 1e-30)
    assert rel(g_exact["scales"], s.grad.numpy()) < 1e-6
    assert rel(g_wheel["scales"] * 0.7, s.grad.numpy()) < 1e-6
    for name, t in (("means3D", m), ("opacities", o), ("shs", sh), ("rotations", r)):
        assert rel(g_wheel[name], t.grad.numpy()) < 1e-6 and np.array_equal(g_wheel[name], g_exact[name]), name


def test_precomputed_colour_and_cov_paths(oracle_built):
    sc = S.make_small_scene(N=32, seed=9)
    st = S.camera_settings(32, 32, 49.1, 10, -40, 1.8)
    color, radii, depth, alpha, state = oracle_forward(sc, st, dtype=np.float64)
    geo = state.geometry()
    # cov3D_precomp equal to R S S R^T and colours equal to the SH evaluation reproduce the image
    q = torch.tensor(sc["rotations"], dtype=torch.float64)
    Rm = R.quat_to_rot(q).numpy()
    Mm = Rm * sc["scales"].astype(np.float64)[:, None, :]
    Sg = Mm @ Mm.transpose(0, 2, 1)
    cov6 = np.stack([Sg[:, 0, 0], Sg[:, 0, 1], Sg[:, 0, 2], Sg[:, 1, 1], Sg[:, 1, 2], Sg[:, 2, 2]], axis=1)
    c2, r2, *_ = O.forward(sc["means3D"], sc["opacities"], st, colors_precomp=geo["rgb"], cov3D_precomp=cov6, dtype=np.float64)
    np.testing.assert_allclose(c2, color, atol=1e-10)
    assert (r2 == radii).all()


def test_golden_vectors(oracle_built):
    """Frozen outputs (tests/golden/make_golden.py) -- guards the oracle itself against drift."""
    path = os.path.join(GOLD, "gs_small.npz")
    z = np.load(path)
    sc = {k: z[k] for k in ("means3D", "opacities", "shs", "scales", "rotations")}
    st = {k: (z["st_" + k].item() if z["st_" + k].ndim == 0 else z["st_" + k]) for k in
          ("image_height", "image_width", "tanfovx", "tanfovy", "bg", "scale_modifier", "viewmatrix", "projmatrix", "sh_degree", "campos")}
    color, radii, depth, alpha, state = oracle_forward(sc, st, dtype=np.float32)
    np.testing.assert_allclose(color, z["color"], atol=2e-6)
    np.testing.assert_allclose(depth, z["depth"], atol=2e-6)
    np.testing.assert_allclose(alpha, z["alpha"], atol=2e-6)
    assert (radii == z["radii"]).all()
    g = O.backward(state, z["gC"], z["gD"], z["gA"])
    for k in ("means3D", "opacities", "shs", "scales", "rotations", "means2D"):
        ref = z["grad_" + k]
        assert np.abs(g[k] - ref).max() <= 2e-4 * max(np.abs(ref).max(), 1e-12), k
