"""simple_knn.distCUDA2 replacement (include/c3d_knn.h): the oracle against brute force on CPU, the HIP kernel against the oracle on the GPU."""
import numpy as np
import pytest
import torch

from oracle import knn_oracle as K


def _brute(p):
    p = np.asarray(p, np.float32).astype(np.float64)
    d2 = ((p[:, None, :] - p[None, :, :]) ** 2).sum(-1)
    np.fill_diagonal(d2, np.inf)
    return (np.sort(d2, axis=1)[:, :3].sum(1) / 3.0).astype(np.float32)


def _cases():
    rng = np.random.default_rng(0)
    ball = rng.normal(size=(3000, 3)); ball = ball / np.linalg.norm(ball, axis=1, keepdims=True) * np.cbrt(rng.uniform(size=(3000, 1))) * 0.5
    flat = np.concatenate([rng.uniform(-1, 1, size=(2000, 2)), np.zeros((2000, 1))], 1)                  # degenerate extent along z
    clustered = np.concatenate([rng.normal(scale=1e-3, size=(1500, 3)), rng.normal(scale=1.0, size=(500, 3)) + 5.0])   # very uneven density
    dup = np.repeat(rng.uniform(size=(300, 3)), 5, axis=0)                                                # every position five times
    line = np.stack([np.linspace(0, 1, 700), np.zeros(700), np.zeros(700)], 1)
    return {"ball": ball, "flat": flat, "clustered": clustered, "duplicates": dup, "line": line, "tiny": rng.uniform(size=(4, 3)), "five": rng.uniform(size=(5, 3))}


@pytest.mark.parametrize("name", list(_cases()))
def test_knn_oracle_equals_brute_force(name):
    p = _cases()[name].astype(np.float32)
    a, b = K.dist2_mean3(p), _brute(p)
    assert np.allclose(a, b, rtol=1e-6, atol=1e-12), name


@pytest.mark.gpu
@pytest.mark.parametrize("name", list(_cases()) + ["big"])
def test_knn_hip_matches_oracle(name):
    if not torch.cuda.is_available():
        pytest.fail("-m gpu tests need a HIP device (no CPU fallback exists)")
    from simple_knn._C import distCUDA2
    p = (np.random.default_rng(3).uniform(-2, 3, size=(200000, 3)) if name == "big" else _cases()[name]).astype(np.float32)
    got = distCUDA2(torch.tensor(p, device="cuda")).cpu().numpy()
    ref = K.dist2_mean3(p)
    # squared distances in float32 on the GPU, float64 in the oracle; near-ties between the 3rd and 4th neighbour resolve to the same sum
    assert got.shape == ref.shape and np.allclose(got, ref, rtol=2e-5, atol=1e-10), (name, np.abs(got - ref).max())
    with pytest.raises(RuntimeError):
        distCUDA2(torch.tensor(p[:10]))                           # CPU tensor: no fallback


@pytest.mark.gpu
def test_create_from_pcd_and_mesh_follow_the_reference_recipe():
    """GaussianModel.create_from_pcd (reference :407-433) with the HIP distCUDA2; renderer.initialize() on None / PointCloud / Mesh inputs."""
    from MVs_Algorithms.GaussianSplatting.main_3DGS_renderer import GaussianSplattingRenderer, PointCloud, inverse_sigmoid
    from shared_utils.sh_utils import RGB2SH
    from c3d_hip import synthetic as S
    rng = np.random.default_rng(1)
    pts, cols = rng.normal(size=(5000, 3)).astype(np.float32) * 0.3, rng.uniform(size=(5000, 3)).astype(np.float32)
    r = GaussianSplattingRenderer(sh_degree=3, device="cuda")
    r.initialize(PointCloud(points=pts, colors=cols, normals=np.zeros_like(pts)))
    g = r.gaussians
    assert g.spatial_lr_scale == 1 and g._xyz.shape == (5000, 3) and g._features_rest.shape == (5000, 15, 3)
    ref_scale = np.log(np.sqrt(np.maximum(K.dist2_mean3(pts), 1e-7)))
    assert np.allclose(g._scaling.detach().cpu().numpy(), np.repeat(ref_scale[:, None], 3, 1), rtol=1e-4, atol=1e-5)
    assert torch.allclose(g._features_dc[:, 0].detach().cpu(), RGB2SH(torch.tensor(cols)), atol=1e-6) and float(g._features_rest.detach().abs().sum()) == 0.0
    assert torch.allclose(g._opacity.detach(), inverse_sigmoid(torch.full_like(g._opacity, 0.1))) and torch.equal(g._rotation[:, 0].detach(), torch.ones(5000, device="cuda"))
    r2 = GaussianSplattingRenderer(sh_degree=3, device="cuda")
    r2.initialize(None, num_pts=2000, radius=0.5)                 # the reference's random ball, lr scale 10
    assert r2.gaussians.spatial_lr_scale == 10 and r2.gaussians._xyz.shape == (2000, 3) and float(r2.gaussians._xyz.detach().norm(dim=1).max()) <= 0.5 + 1e-6
    v, f, vt, vn = S.make_uv_sphere(12, 16, radius=0.6, displacement=0.0)
    mesh = type("M", (), {"v": torch.tensor(v), "f": torch.tensor(f)})()
    r3 = GaussianSplattingRenderer(sh_degree=3, device="cuda")
    r3.initialize(mesh, num_pts=3000)
    x = r3.gaussians._xyz.detach().cpu().numpy()
    assert x.shape[0] == int(np.ceil(3000 / f.shape[0])) * f.shape[0]
    assert np.abs(np.linalg.norm(x, axis=1) - 0.6).max() <= 0.6 * (1 - np.cos(np.pi / 12)) + 1e-3      # samples lie on the faces of the sphere
