"""The mesh renderer's glue against a run of the REFERENCE'S OWN DiffRastRenderer.render.

tests/golden/ref_mesh_render.npz holds what /root/reference/MVs_Algorithms/DiffRastMesh/diff_mesh_renderer.py:72-159 returned in the
build container when its `nvdiffrast.torch` was tests/fake_dr.py (the CPU mesh oracle behind the same names;
tests/golden/make_golden_ref_render.py).  Here the mirror (comfyui-3d-pack_amd/MVs_Algorithms/DiffRastMesh/diff_mesh_renderer.py) runs on
the CPU over the same stand-in -- its torch chain; the fused HIP glue is held to that chain by tests/test_mesh_hip.py -- and must return
the same images from the same op calls.  Pinned: vertex transform, op order / arguments, sigmoid, depth / normal / viewcos, the
vertex-normal rebuild under train_geo, compositing, SSAA resize, clamps, result keys.  Not pinned by this: the four ops themselves."""
import os
import subprocess
import sys
from types import SimpleNamespace

import numpy as np
import pytest
import torch

GOLD_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
sys.path.insert(0, GOLD_DIR)


@pytest.fixture(scope="module")
def g():
    z = np.load(os.path.join(GOLD_DIR, "ref_mesh_render.npz"))
    return {k: z[k] for k in z.files}


def test_fixture_is_what_the_reference_produces_now():
    if not os.path.isdir("/root/reference/MVs_Algorithms"):
        pytest.skip("/root/reference is not mounted here")
    r = subprocess.run([sys.executable, os.path.join(GOLD_DIR, "make_golden_ref_render.py"), "--check"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr


def _close(a, b, what):
    """images from two float32 pipelines: equal up to rounding, except that a 1e-7 difference in a clip coordinate may move a snapped
    vertex and flip a handful of edge pixels"""
    d = np.abs(a.astype(np.float64) - b.astype(np.float64))
    assert a.shape == b.shape, what
    assert np.median(d) <= 1e-6 and (d > 1e-4).mean() <= 0.004, (what, float(d.mean()), float((d > 1e-4).mean()))


@pytest.mark.parametrize("case", ["plain", "geo", "ssaa", "image_only"])
def test_mirror_render_equals_the_reference_render(g, case, monkeypatch):
    import fake_dr
    from make_golden_ref_render import CASES
    from MVs_Algorithms.DiffRastMesh import diff_mesh_renderer as MR
    monkeypatch.setattr(MR, "dr", fake_dr)
    name, dxdy, h0, w0, ssaa, bg, train_geo, types_ = next(c for c in CASES if c[0] == case)
    T = lambda k: torch.from_numpy(g["scene_" + k].copy())
    mesh = SimpleNamespace(v=T("v"), f=T("f"), vt=T("vt"), ft=T("f"), vn=T("vn"), fn=T("f"), albedo=T("albedo"))
    r = MR.DiffRastRenderer(mesh, True)
    r.get_params(0.01, train_geo, 0.001)
    with torch.no_grad():
        r.v_offsets.copy_(T("v_offsets"))
        r.raw_albedo.add_(T("raw_albedo_delta"))
    del fake_dr.CALLS[:]
    with torch.no_grad():
        res = r.render(g[case + "_pose"], g[case + "_proj"], h0, w0, ssaa=ssaa, bg_color=bg, optional_render_types=list(types_))
        eager_calls = [repr(c) for c in fake_dr.CALLS]
        assert sorted(res.keys()) == list(g[case + "_keys"])
        for k in sorted(res.keys()):
            _close(res[k].numpy(), g[case + "_" + k], (case, k))
    ref_calls = list(g[case + "_calls"])
    all_calls = [repr(c) for c in fake_dr.CALLS]
    assert sorted(all_calls) == sorted(ref_calls)                    # the same op calls with the same shapes and modes ...
    # ... in the reference's order, except that depth / normal are produced on first access (after the image) instead of eagerly
    lazy = [c for c in ref_calls if c.startswith("('interpolate'") and "True" not in c]
    assert eager_calls == [c for c in ref_calls if c not in lazy]
    assert all_calls == eager_calls + lazy


# ------------------------------------------------------------------------------------------------ 3DGS: the glue in front of the rasterizer
@pytest.fixture(scope="module")
def gg():
    z = np.load(os.path.join(GOLD_DIR, "ref_gs_render.npz"))
    return {k: z[k] for k in z.files}


def test_gs_fixture_is_what_the_reference_produces_now():
    if not os.path.isdir("/root/reference/MVs_Algorithms"):
        pytest.skip("/root/reference is not mounted here")
    r = subprocess.run([sys.executable, os.path.join(GOLD_DIR, "make_golden_ref_gs_render.py"), "--check"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr


@pytest.mark.parametrize("case", ["default", "modifier_bg", "subset", "override", "cov_python"])
def test_gs_mirror_hands_the_rasterizer_what_the_reference_does(gg, case, monkeypatch):
    """GaussianSplattingRenderer.render of the mirror over tests/fake_dgr.py against the recorded call of the reference's render
    (main_3DGS_renderer.py:830-957): settings tuple, activated tensors per keyword, outputs, result dict."""
    import fake_dgr
    from make_golden_ref_gs_render import CASES
    monkeypatch.setitem(sys.modules, "diff_gaussian_rasterization", fake_dgr)
    from MVs_Algorithms.GaussianSplatting.main_3DGS_renderer import GaussianSplattingRenderer
    from shared_utils.camera_utils import MiniCam
    T = lambda k: torch.from_numpy(gg["scene_" + k].copy())
    r = GaussianSplattingRenderer(sh_degree=3, white_background=True, radius=1, device="cpu")
    g = r.gaussians
    g.create_from_tensors(T("xyz"), torch.cat((T("f_dc"), T("f_rest")), dim=1), T("scaling"), T("rotation"), T("opacity"))
    g.active_sh_degree = 2
    W, H = (int(v) for v in gg["size"])
    mc = MiniCam(gg["pose"].copy(), W, H, float(gg["fovy"]), float(gg["fovx"]), 0.01, 100, device="cpu")
    kw = {k: (T(v) if isinstance(v, str) else v) for k, v in dict(next(c for c in CASES if c[0] == case)[1]).items()}
    del fake_dgr.CALLS[:]
    res = r.render(mc, **kw)
    assert len(fake_dgr.CALLS) == 1
    call = fake_dgr.CALLS[0]
    pre = case + "_"
    for k, v in call["settings"].items():
        want = gg[pre + "set_" + k]
        if isinstance(v, np.ndarray):
            np.testing.assert_allclose(v, want, rtol=1e-6, atol=1e-7, err_msg=k)
        else:
            assert v == want.item(), (k, v, want)
    for k, v in call.items():
        if k == "settings":
            continue
        want = gg[pre + "arg_" + k]
        if v is None:
            assert want.size == 0, k                                  # the reference passed None here as well
        else:
            assert want.size > 0 and v.shape == want.shape, (k, v.shape, want.shape)
            np.testing.assert_allclose(v, want, rtol=1e-6, atol=1e-7, err_msg=k)
    assert sorted(res.keys()) == list(gg[pre + "keys"])
    for k in res:
        got = res[k].detach().numpy()
        want = gg[pre + "out_" + k]
        assert got.shape == want.shape and got.dtype == want.dtype, (k, got.dtype, want.dtype)
        np.testing.assert_allclose(got.astype(np.float64), want.astype(np.float64), rtol=0, atol=2e-6, err_msg=k)
    assert bool(res["viewspace_points"].requires_grad) == bool(gg[pre + "viewspace_requires_grad"])
    assert float(res["image"].detach().max()) <= 1.0 and float(res["image"].detach().min()) >= 0.0


def test_gs_orbit_loop_matches_the_reference_controller(gg, monkeypatch):
    """GaussianSplattingCameraController.render_all_pose (the renderer nodes' orbit loop) of the mirror over tests/fake_dgr.py against the
    reference's controller: orbit poses -> MiniCam, the random black / white background per view, stacked images / masks / extras."""
    import fake_dgr
    from make_golden_ref_gs_render import ORBIT_POSES
    monkeypatch.setitem(sys.modules, "diff_gaussian_rasterization", fake_dgr)
    monkeypatch.setattr(fake_dgr, "RECORD", True)
    from MVs_Algorithms.GaussianSplatting.main_3DGS import GaussianSplattingCameraController
    from MVs_Algorithms.GaussianSplatting.main_3DGS_renderer import GaussianSplattingRenderer
    T = lambda k: torch.from_numpy(gg["scene_" + k].copy())
    r = GaussianSplattingRenderer(sh_degree=3, white_background=True, radius=1, device="cpu")
    r.gaussians.create_from_tensors(T("xyz"), torch.cat((T("f_dc"), T("f_rest")), dim=1), T("scaling"), T("rotation"), T("opacity"))
    ctl = GaussianSplattingCameraController(r, 72, 48, 49.1, 0.5, None, "cpu")
    np.random.seed(4)
    del fake_dgr.CALLS[:]
    with torch.no_grad():
        images, masks, extra = ctl.render_all_pose(ORBIT_POSES)
    np.testing.assert_allclose(np.stack([c["settings"]["bg"] for c in fake_dgr.CALLS]), gg["orbit_bg_per_view"], atol=0)
    np.testing.assert_allclose(np.stack([c["settings"]["viewmatrix"] for c in fake_dgr.CALLS]), gg["orbit_viewmatrix_per_view"], atol=1e-6)
    assert images.shape == gg["orbit_images"].shape and masks.shape == gg["orbit_masks"].shape
    np.testing.assert_allclose(images.numpy(), gg["orbit_images"], atol=2e-6)
    np.testing.assert_allclose(masks.numpy(), gg["orbit_masks"], atol=2e-6)
    assert sorted(extra.keys()) == list(gg["orbit_extra_keys"])
    for k in extra:
        np.testing.assert_allclose(extra[k].numpy().astype(np.float64), gg["orbit_extra_" + k].astype(np.float64), atol=2e-6, err_msg=k)
