"""Model construction and the 3DGS PLY wire format against runs of the REFERENCE'S OWN code.

tests/golden/ref_gs_io.npz: GaussianModel.create_from_pcd / to_ply / create_from_ply / get_points_cloud and the mesh_utils PLY helpers of
the reference, executed on the CPU in the build container (tests/golden/make_golden_ref_gs_io.py: what was replaced and why).  The
mirrors must produce the same tensors, the same property list and the same bytes per property."""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

GOLD_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
sys.path.insert(0, GOLD_DIR)
NAMES = {"xyz": "xyz", "features_dc": "f_dc", "features_rest": "f_rest", "scaling": "scaling", "rotation": "rotation", "opacity": "opacity"}


@pytest.fixture(scope="module")
def z():
    f = np.load(os.path.join(GOLD_DIR, "ref_gs_io.npz"))
    return {k: f[k] for k in f.files}


def test_fixture_is_what_the_reference_produces_now():
    if not os.path.isdir("/root/reference/MVs_Algorithms"):
        pytest.skip("/root/reference is not mounted here")
    r = subprocess.run([sys.executable, os.path.join(GOLD_DIR, "make_golden_ref_gs_io.py"), "--check"], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]


def _ply_matrix(plydata):
    el = plydata.elements[0]
    names = [p.name for p in el.properties]
    return names, np.stack([np.asarray(el[n], dtype=np.float32) for n in names], axis=1)


@pytest.mark.parametrize("deg", [3, 1])
def test_model_construction_and_ply_round_trip_match_the_reference(z, deg, monkeypatch):
    import simple_knn._C as KNN
    from oracle import knn_oracle
    from MVs_Algorithms.GaussianSplatting.main_3DGS_renderer import GaussianModel, PointCloud
    from mesh_processer import mesh_utils as MU
    # the HIP kNN kernel needs a device; its statistic is checked against this same oracle in tests/test_knn.py
    monkeypatch.setattr(KNN, "distCUDA2", lambda pts: torch.from_numpy(knn_oracle.dist2_mean3(pts.detach().cpu().numpy())))
    pre = "deg%d_" % deg
    g = GaussianModel(deg, device="cpu")
    g.create_from_pcd(PointCloud(points=z["pcd_points"], colors=z["pcd_colors"], normals=np.zeros_like(z["pcd_points"])), 7.5)
    P = g._param_dict()
    for k, name in NAMES.items():
        np.testing.assert_allclose(P[name].detach().numpy(), z[pre + "pcd_" + k], rtol=1e-6, atol=1e-7, err_msg=k)
        assert P[name].requires_grad
    np.testing.assert_allclose(g.init_xyz.numpy(), z[pre + "pcd_init_xyz"], atol=0)
    assert np.array_equal(g.max_radii2D.numpy(), z[pre + "pcd_max_radii2D"]) and g.spatial_lr_scale == float(z[pre + "pcd_spatial_lr_scale"])
    # a trained-looking model -> PLY: same property list, same values per property
    feats = torch.cat((torch.from_numpy(z[pre + "model_features_dc"]), torch.from_numpy(z[pre + "model_features_rest"])), dim=1)
    g.create_from_tensors(torch.from_numpy(z[pre + "model_xyz"]), feats, torch.from_numpy(z[pre + "model_scaling"]),
                          torch.from_numpy(z[pre + "model_rotation"]), torch.from_numpy(z[pre + "model_opacity"]))
    pd = g.to_ply()
    names, mat = _ply_matrix(pd)
    assert names == list(z[pre + "ply_names"])
    assert np.array_equal(mat, z[pre + "ply_data"])
    assert all(p.dtype == "f4" for p in pd.elements[0].properties)
    assert MU.calculate_max_sh_degree_from_gs_ply(pd)[0] == int(z[pre + "ply_max_sh_degree"]) == deg
    for k, v in zip(("xyz", "features_dc", "features_extra", "opacities", "scales", "rots"), MU.read_gs_ply(pd)):
        assert v.shape == z[pre + "read_" + k].shape and np.array_equal(np.asarray(v, np.float64), z[pre + "read_" + k]), k
    g2 = GaussianModel(deg, device="cpu")
    g2.create_from_ply(pd)
    for k, name in NAMES.items():
        assert np.array_equal(g2._param_dict()[name].detach().numpy(), z[pre + "fromply_" + k]), k
    assert g2.active_sh_degree == int(z[pre + "fromply_active_sh_degree"])
    pc = g.get_points_cloud()
    np.testing.assert_allclose(np.asarray(pc.points), z[pre + "cloud_points"], atol=0)
    np.testing.assert_allclose(np.asarray(pc.colors), z[pre + "cloud_colors"], rtol=1e-6, atol=1e-7)
    # axis switches of the file
    for tag, axis, scale, inv in (("swapA", [2, 0, 1], [1.0, 1.0, 1.0], 0), ("swapB", [0, 2, 1], [1.0, -1.0, 2.0], 1)):
        n2, m2 = _ply_matrix(MU.switch_ply_axis_and_scale(pd, axis, scale, inv))
        assert n2 == list(z[pre + tag + "_names"])
        np.testing.assert_allclose(m2, z[pre + tag + "_data"], rtol=1e-5, atol=1e-6, err_msg=tag)


def test_mesh_container_methods_match_the_reference():
    """Mesh.auto_normal / aabb / auto_size / set_new_albedo against a run of the reference's class (tests/golden/make_golden_ref_mesh_class.py)"""
    from mesh_processer.mesh import Mesh
    if os.path.isdir("/root/reference/mesh_processer"):
        r = subprocess.run([sys.executable, os.path.join(GOLD_DIR, "make_golden_ref_mesh_class.py"), "--check"], capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    z = np.load(os.path.join(GOLD_DIR, "ref_mesh_class.npz"))
    T = lambda k: torch.from_numpy(z[k].copy())
    m = Mesh(v=T("v"), f=T("f"), device="cpu")
    m.auto_normal()
    np.testing.assert_allclose(m.vn.numpy(), z["vn"], rtol=1e-6, atol=1e-7)
    assert np.array_equal(m.fn.numpy(), z["fn"])
    lo, hi = m.aabb()
    assert np.array_equal(lo.numpy(), z["aabb_min"]) and np.array_equal(hi.numpy(), z["aabb_max"])
    m.auto_size(bound=0.9)
    np.testing.assert_allclose(m.v.numpy(), z["sized_v"], rtol=1e-6, atol=1e-7)
    np.testing.assert_allclose(np.asarray(m.ori_center), z["ori_center"], atol=1e-7)
    assert abs(m.ori_scale - float(z["ori_scale"])) <= 1e-6 * float(z["ori_scale"])
    m.set_new_albedo(6, 10)
    assert m.albedo.shape == (6, 10, 3) and np.array_equal(m.albedo.numpy(), z["gray_albedo"])
    m2 = Mesh(v=T("v"), f=T("f"), albedo=T("tex"), device="cpu")
    m2.set_new_albedo(12, 10)
    np.testing.assert_allclose(m2.albedo.numpy(), z["resized_albedo"], rtol=1e-6, atol=1e-7)
