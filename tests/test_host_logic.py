"""CPU tests of the host-side mirror: camera conventions, parameter store / optimizer groups, LR schedule, MS-SSIM,
and the world_size-2 gradient exchange over gloo."""
import math
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

PKG = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "comfyui-3d-pack_amd")


def test_orbit_camera_and_minicam_conventions():
    from shared_utils.camera_utils import MiniCam, OrbitCamera, orbit_camera
    c2w = orbit_camera(0, 0, 2.0)
    np.testing.assert_allclose(c2w[:3, 3], [0, 0, 2], atol=1e-6)                 # azimuth 0: camera on +z
    np.testing.assert_allclose(orbit_camera(-90 + 1e-3, 0, 1.0)[:3, 3], [0, 1, 0], atol=1e-4)   # negative elevation = above
    np.testing.assert_allclose(orbit_camera(0, 90, 1.0)[:3, 3], [1, 0, 0], atol=1e-6)
    cam = OrbitCamera(64, 48, fovy=49.1)
    mc = MiniCam(c2w, 64, 48, cam.fovy, cam.fovx, 0.01, 100, device="cpu")
    # world origin is 2 units in front of the camera (view z = +2), and the reference's sign quirks hold
    p = torch.tensor([0.0, 0, 0, 1]) @ mc.world_view_transform
    assert abs(p[2].item() - 2.0) < 1e-6
    np.testing.assert_allclose(mc.camera_center.numpy(), -c2w[:3, 3])            # camera_center = -c2w[:3,3]
    ph = torch.tensor([0.0, 0, 0, 1]) @ mc.full_proj_transform
    assert abs(ph[3].item() - 2.0) < 1e-6 and abs(ph[0].item()) < 1e-6           # w = view depth
    P = cam.perspective
    assert P[1, 1] < 0 and P[3, 2] == -1                                         # y flipped OpenGL projection


def test_gaussian_model_groups_and_lr_schedule():
    from MVs_Algorithms.GaussianSplatting.main_3DGS import GSParams
    from MVs_Algorithms.GaussianSplatting.main_3DGS_renderer import GaussianSplattingRenderer, get_expon_lr_func
    p = GSParams(num_pts=500)
    assert (p.training_iterations, p.batch_size, p.lambda_ssim, p.lambda_alpha, p.feature_lr, p.sh_degree) == (30000, 1, 0.2, 3, 0.0025, 3)
    np.random.seed(0)
    r = GaussianSplattingRenderer(sh_degree=3, device="cpu")
    r.initialize(None, num_pts=500)
    g = r.gaussians
    assert g.get_features.shape == (500, 16, 3) and g.get_scaling.shape == (500, 3) and g.get_rotation.shape == (500, 4)
    assert torch.allclose(g.get_opacity, torch.full((500, 1), 0.1), atol=1e-6)
    assert torch.allclose(g.get_rotation.norm(dim=1), torch.ones(500))
    assert (g.get_xyz.norm(dim=1) <= 0.5 + 1e-6).all()
    g.training_setup(p)
    names = [grp["name"] for grp in g.optimizer.param_groups]
    assert names == ["xyz", "f_dc", "f_rest", "opacity", "scaling", "rotation"]
    lrs = {grp["name"]: grp["lr"] for grp in g.optimizer.param_groups}
    assert math.isclose(lrs["xyz"], 0.00016 * 10) and math.isclose(lrs["f_rest"], 0.0025 / 20) and g.optimizer.defaults["eps"] == 1e-15
    f = get_expon_lr_func(1.6e-3, 1.6e-5, lr_delay_mult=0.01, max_steps=30000)
    assert math.isclose(f(0), 1.6e-3) and math.isclose(f(30000), 1.6e-5) and math.isclose(f(15000), math.sqrt(1.6e-3 * 1.6e-5), rel_tol=1e-9)
    assert math.isclose(g.update_learning_rate(15000), math.sqrt(1.6e-3 * 1.6e-5), rel_tol=1e-6)


def test_raw_storage_guard_sends_unsupported_sh_storage_down_the_per_view_path():
    """render() / render_views / render_all_pose take the raw-parameter kernels only for SH storage they can read (degree 0-3, K in 1/4/9/16, K >= (active + 1)^2)"""
    from MVs_Algorithms.GaussianSplatting.main_3DGS_renderer import GaussianSplattingRenderer
    np.random.seed(0)
    for deg, ok in ((0, True), (1, True), (3, True), (4, False)):
        r = GaussianSplattingRenderer(sh_degree=deg, device="cpu")
        r.initialize(None, num_pts=50)
        assert r.raw_storage_ok() is ok, deg
    r = GaussianSplattingRenderer(sh_degree=3, device="cpu")
    r.initialize(None, num_pts=50)
    g = r.gaussians
    g._features_rest = torch.nn.Parameter(g._features_rest.data[:, :5].contiguous())      # six coefficients per channel: no kernel instance reads that
    assert not r.raw_storage_ok()
    g._features_rest = torch.nn.Parameter(g._features_rest.data[:, :3].contiguous())      # a degree-1 PLY in a degree-3 renderer: fine while the active degree fits
    g.active_sh_degree = 1
    assert r.raw_storage_ok()
    g.active_sh_degree = 2
    assert not r.raw_storage_ok()


def test_ms_ssim_properties():
    from shared_utils.msssim import MS_SSIM
    torch.manual_seed(0)
    m = MS_SSIM(data_range=1, size_average=True, channel=3)
    x = torch.rand(2, 3, 192, 200)
    assert abs(m(x, x).item() - 1.0) < 1e-6
    y = (x + 0.1 * torch.randn_like(x)).clamp(0, 1)
    v = m(x, y).item()
    assert 0 < v < 1 and abs(v - m(y, x).item()) < 1e-6
    assert m(x, (x + 0.3 * torch.randn_like(x)).clamp(0, 1)).item() < v
    y.requires_grad_(True)
    (1 - m(x, y)).backward()
    assert torch.isfinite(y.grad).all()


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, mode, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path[:0] = [root, os.path.join(root, "comfyui-3d-pack_amd")]
    from c3d_hip import parallel
    torch.manual_seed(0)
    N = 257
    params = [torch.nn.Parameter(torch.randn(N, *s)) for s in ((3,), (1, 3), (15, 3), (1,), (3,), (4,))]
    parallel.broadcast_parameters(params, src=0)
    views = list(range(8))
    mine = parallel.shard_views(views, rank, world)
    g = torch.Generator().manual_seed(100 + rank)
    for p in params:   # stand-in for "this rank's views' gradients"
        p.grad = sum(torch.randn(p.shape, generator=torch.Generator().manual_seed(1000 * v + i)) for i, v in enumerate(mine)) * 1.0
    # the same exchange through the flat gradient buffer the fused step writes into (no cat / copy-back)
    fg = parallel.FlatGrads(params)
    for v_, p in zip(fg.views, params):
        v_.copy_(p.grad)
    assert all(v_.data_ptr() % 16 == 0 for v_ in fg.views)
    fg2 = parallel.FlatGrads(params)        # the chunked, overlapped form of the all-reduce: big tensors range by range, the rest at the end
    for v_, p in zip(fg2.views, params):
        v_.copy_(p.grad)
    fg.exchange(None, mode, average=False)
    if mode == "allreduce":
        assert fg2._big == [2] and len(fg2._rest_spans) == 2          # f_rest goes by rows; [xyz | f_dc] and [opacity | scaling | rotation] are the two spans left
        for g0, g1 in ((0, 100), (100, 256), (256, 257)):
            fg2.exchange_rows(g0, g1)
        fg2.exchange_finish()
        assert torch.equal(fg2.flat, fg.flat)
    parallel.exchange_gradients(params, None, mode, average=False)
    flat = parallel.flatten_grads(params)
    assert flat.shape == (N, 59)
    for v_, p in zip(fg.views, params):
        assert torch.equal(v_, p.grad)
    out[rank] = (flat.clone(), mine)
    dist.destroy_process_group()


@pytest.mark.parametrize("mode", ["allgather", "allreduce"])
def test_gradient_exchange_world2_gloo(mode):
    """N>1 path on CPU: two gloo ranks shard 8 views 4/4, exchange once, and end with identical gradients equal to the
    single-process sum over all views."""
    world, port = 2, _free_port()
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(world, port, mode, out), nprocs=world, join=True)
    (f0, v0), (f1, v1) = out[0], out[1]
    assert sorted(v0 + v1) == list(range(8)) and len(v0) == len(v1) == 4
    assert torch.equal(f0, f1)                                 # replicas bit-identical
    # reference: rank-ordered sum of the two shards
    import sys
    from c3d_hip import parallel
    shapes = ((3,), (1, 3), (15, 3), (1,), (3,), (4,))
    tot = []
    for s in shapes:
        per_rank = []
        for mine in (v0, v1):
            per_rank.append(sum(torch.randn((257,) + s, generator=torch.Generator().manual_seed(1000 * v + i)) for i, v in enumerate(mine)) * 1.0)
        tot.append((per_rank[0] + per_rank[1]).reshape(257, -1))
    ref = torch.cat(tot, dim=1)
    assert torch.allclose(f0, ref, atol=1e-6)
    if mode == "allgather":
        assert torch.equal(f0, ref)                            # fixed rank-order summation


def test_gs_ply_wire_format_roundtrip(tmp_path):
    """3DGS PLY schema: property order/names, float32, channel-major SH; file round trip; model round trip."""
    from c3d_hip.ply import PlyData
    from mesh_processer.mesh_utils import calculate_max_sh_degree_from_gs_ply, read_gs_ply, switch_ply_axis_and_scale
    from MVs_Algorithms.GaussianSplatting.main_3DGS_renderer import GaussianModel
    torch.manual_seed(0)
    N, deg = 37, 2
    K = (deg + 1) ** 2
    g = GaussianModel(deg, device="cpu")
    g.create_from_tensors(torch.randn(N, 3), torch.randn(N, K, 3), torch.randn(N, 3), torch.randn(N, 4), torch.randn(N, 1))
    ply = g.to_ply()
    names = [p.name for p in ply.elements[0].properties]
    assert names[:9] == ['x', 'y', 'z', 'nx', 'ny', 'nz', 'f_dc_0', 'f_dc_1', 'f_dc_2']
    assert names[9:9 + 3 * (K - 1)] == ['f_rest_%d' % i for i in range(3 * (K - 1))]
    assert names[-8:] == ['opacity', 'scale_0', 'scale_1', 'scale_2', 'rot_0', 'rot_1', 'rot_2', 'rot_3']
    assert all(p.dtype == 'f4' for p in ply.elements[0].properties)
    assert calculate_max_sh_degree_from_gs_ply(ply)[0] == deg
    # channel-major: f_rest_j for j < K-1 is channel 0
    np.testing.assert_allclose(np.asarray(ply.elements[0]['f_rest_1']), g._features_rest[:, 1, 0].detach().numpy())
    np.testing.assert_allclose(np.asarray(ply.elements[0]['f_rest_%d' % (K - 1)]), g._features_rest[:, 0, 1].detach().numpy())
    path = tmp_path / "cloud.ply"
    ply.write(str(path))
    head = open(path, "rb").read(200)
    assert head.startswith(b"ply\nformat binary_little_endian 1.0\nelement vertex 37\nproperty float x\n")
    back = PlyData.read(str(path))
    g2 = GaussianModel(deg, device="cpu")
    g2.create_from_ply(back)
    for a, b in ((g._xyz, g2._xyz), (g._features_dc, g2._features_dc), (g._features_rest, g2._features_rest), (g._opacity, g2._opacity),
                 (g._scaling, g2._scaling), (g._rotation, g2._rotation)):
        assert torch.equal(a.detach(), b.detach())
    # ascii variant reads back too
    ply.text = True
    ply.write(str(tmp_path / "a.ply"))
    xyz2 = read_gs_ply(PlyData.read(str(tmp_path / "a.ply")))[0]
    np.testing.assert_allclose(xyz2, g._xyz.detach().numpy(), rtol=1e-6)
    # axis switch: identity is a no-op, a cyclic permutation applied three times returns to the start
    same = switch_ply_axis_and_scale(ply, [0, 1, 2], [1, 1, 1], 0)
    np.testing.assert_allclose(read_gs_ply(same)[0], read_gs_ply(ply)[0], atol=1e-6)
    q0 = read_gs_ply(ply)[5]; q0 = q0 / np.linalg.norm(q0, axis=1, keepdims=True)
    q1 = read_gs_ply(same)[5]
    np.testing.assert_allclose(np.abs((q0 * q1).sum(1)), 1.0, atol=1e-5)        # same rotation up to sign
    p3 = ply
    for _ in range(3):
        p3 = switch_ply_axis_and_scale(p3, [1, 2, 0], [1, 1, 1], 0)
    np.testing.assert_allclose(read_gs_ply(p3)[0], read_gs_ply(ply)[0], atol=1e-5)
    np.testing.assert_allclose(read_gs_ply(p3)[4], read_gs_ply(ply)[4], atol=1e-5)


def test_node_layer_contract_and_pose_stacking():
    """hot-path node classes keep the reference's names, input keys and the example workflow's 45-pose stack"""
    import nodes as N
    assert set(N.NODE_CLASS_MAPPINGS) >= {"[Comfy3D] Gaussian Splatting Orbit Renderer", "[Comfy3D] Gaussian Splatting 3D",
                                           "[Comfy3D] Mesh Orbit Renderer", "[Comfy3D] Fitting Mesh With Multiview Images",
                                           "[Comfy3D] Load 3DGS", "[Comfy3D] Save 3DGS", "[Comfy3D] Switch 3DGS Axis", "[Comfy3D] Stack Orbit Camera Poses"}
    req = N.Gaussian_Splatting_3D.INPUT_TYPES()["required"]
    assert list(req)[:6] == ["reference_images", "reference_masks", "reference_orbit_camera_poses", "reference_orbit_camera_fovy", "training_iterations", "batch_size"]
    assert req["training_iterations"][1]["default"] == 30000 and req["gaussian_sh_degree"][1]["default"] == 3 and len(req) == 28
    assert N.Gaussian_Splatting_Orbit_Renderer.RETURN_NAMES == ("rendered_gs_images", "rendered_gs_masks", "rendered_gs_depths")
    assert N.Mesh_Orbit_Renderer.FUNCTION == "render_mesh" and N.Fitting_Mesh_With_Multiview_Images.INPUT_TYPES()["required"]["batch_size"][1]["default"] == 3
    # example workflow (Render_Mesh_and_3DGS_Example.json): radius 1.75, elevation -45..45 step 45, azimuth 0 -> -0.01 step 24 (wraps): 3 x 15
    poses, r, e, a, cx, cy, cz = N.Stack_Orbit_Camera_Poses().get_camposes(1.75, 1.75, 0.1, -45.0, 45.0, 45.0, 0.0, -0.01, 24.0, 0, 0, 0.1, 0, 0, 0.1, 0, 0, 0.1)
    assert len(poses) == 45 and sorted(set(e)) == [-45.0, 0.0, 45.0] and len(set(round(x, 6) for x in a)) == 15
    assert all(len(p) == 6 and p[0] == 1.75 for p in poses) and min(a) >= -180.0 and max(a) <= 180.0
    az = [p[2] for p in poses if p[1] == -45.0]
    assert az[:8] == [0.0, 24.0, 48.0, 72.0, 96.0, 120.0, 144.0, 168.0] and abs(az[8] - (-168.0)) < 1e-6 and abs(az[-1] - (-24.0)) < 1e-6
    # simple linear case
    poses2, *_ = N.Stack_Orbit_Camera_Poses().get_camposes(1.0, 2.0, 0.5, 0, 0, 0, 0, 90, 45, 0, 0, 0.1, 0, 0, 0.1, 0, 0, 0.1)
    assert [p[2] for p in poses2[:3]] == [0, 45, 90] and [p[0] for p in poses2[::3]] == [1.0, 1.5, 2.0] and len(poses2) == 9   # radius slowest


def _three_round_reference(P, mom, stats, thr, percent_dense, extent, min_opacity, noise):
    """Literal restatement of the reference's densify_and_prune (main_3DGS_renderer.py:641-690, :748-781) on plain tensors: clone round
    (append copies), split round (append 2 children per selected point, drop the parents), prune round.  P / mom: dicts of per-point
    arrays (mom = (exp_avg, exp_avg_sq) per name); noise [2*nS,3] = the unit normal draws of the split round."""
    accum, denom = stats
    g = accum / denom
    g[g.isnan()] = 0.0
    n0 = P["xyz"].shape[0]
    smax = torch.exp(P["scaling"]).max(dim=1).values
    sel = (g.norm(dim=-1) >= thr) & (smax <= percent_dense * extent)
    cat = lambda d, add: {k: torch.cat((d[k], add[k]), dim=0) for k in d}
    P = cat(P, {k: v[sel] for k, v in P.items()})
    mom = {k: tuple(torch.cat((m, torch.zeros_like(m[sel])), dim=0) for m in mom[k]) for k in mom}
    padded = torch.zeros(P["xyz"].shape[0])
    padded[:n0] = g.squeeze()
    smax = torch.exp(P["scaling"]).max(dim=1).values
    sel = (padded >= thr) & (smax > percent_dense * extent)
    stds = torch.exp(P["scaling"][sel]).repeat(2, 1)
    q = P["rotation"][sel] / P["rotation"][sel].norm(dim=1, keepdim=True)
    w, x, y, z = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
    R = torch.stack((1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y), 2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x),
                     2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)), dim=1).view(-1, 3, 3).repeat(2, 1, 1)
    child = {k: v[sel].repeat(2, *([1] * (v.ndim - 1))) for k, v in P.items()}
    child["xyz"] = torch.bmm(R, (noise * stds).unsqueeze(-1)).squeeze(-1) + P["xyz"][sel].repeat(2, 1)
    child["scaling"] = torch.log(stds / (0.8 * 2))
    P = cat(P, child)
    mom = {k: tuple(torch.cat((m, torch.zeros_like(child[k])), dim=0) for m in mom[k]) for k in mom}
    drop = torch.cat((sel, torch.zeros(2 * int(sel.sum()), dtype=torch.bool)))
    P = {k: v[~drop] for k, v in P.items()}
    mom = {k: tuple(m[~drop] for m in mom[k]) for k in mom}
    # prune round: the statistics were reset by the two rounds above, so only opacity and world size can fire
    dead = (torch.sigmoid(P["opacity"]).squeeze(-1) < min_opacity) | (torch.exp(P["scaling"]).max(dim=1).values > 0.1 * extent)
    return {k: v[~dead] for k, v in P.items()}, {k: tuple(m[~dead] for m in mom[k]) for k in mom}


def test_densify_and_prune_one_pass_equals_three_rounds():
    """GaussianModel.densify_and_prune (one source-index list, one gather per array) against the clone -> split -> prune rounds of the
    reference; also reset_opacity and the optimizer-state surgery (moments follow their points, new points start at zero)."""
    from MVs_Algorithms.GaussianSplatting.main_3DGS_renderer import GaussianModel
    from MVs_Algorithms.GaussianSplatting.main_3DGS import GSParams
    torch.manual_seed(5)
    n = 400
    gm = GaussianModel(3, device="cpu")
    feats = torch.randn(n, 16, 3) * 0.2
    scal = torch.log((torch.rand(n, 1) * 0.09 + 0.005) * (0.8 + 0.2 * torch.rand(n, 3)))   # around the percent_dense * extent = 0.04 split/clone boundary
    scal[:10] = torch.log(torch.tensor(0.5))                        # world-size prune (> 0.1 * extent)
    opac = torch.randn(n, 1) * 2.0
    opac[10:30] = -7.0                                              # opacity prune
    gm.create_from_tensors(torch.randn(n, 3) * 0.3, feats, scal, torch.randn(n, 4), opac)
    gm.training_setup(GSParams())
    names = GaussianModel._NAMES
    for _ in range(3):                                              # populate the Adam moments
        for t in gm._param_dict().values():
            t.grad = torch.randn_like(t)
        gm.optimizer.step()
    gm.xyz_gradient_accum = torch.rand(n, 1) * 0.001
    gm.denom = torch.randint(0, 3, (n, 1)).float()                  # zeros -> NaN -> 0 as in the reference
    P0 = {k: v.detach().clone() for k, v in gm._param_dict().items()}
    M0 = {grp["name"]: (gm.optimizer.state[grp["params"][0]]["exp_avg"].clone(), gm.optimizer.state[grp["params"][0]]["exp_avg_sq"].clone())
          for grp in gm.optimizer.param_groups}
    stats0 = (gm.xyz_gradient_accum.clone(), gm.denom.clone())
    gen = torch.Generator().manual_seed(11)
    info = gm.densify_and_prune(0.0002, min_opacity=0.005, extent=4, max_screen_size=1, generator=gen)
    assert info["cloned"] > 20 and info["split"] > 20 and info["pruned"] >= 30
    noise = torch.randn((2 * info["split"], 3), generator=torch.Generator().manual_seed(11))
    P1, M1 = _three_round_reference(P0, M0, stats0, 0.0002, 0.01, 4, 0.005, noise)
    got = gm._param_dict()
    assert got["xyz"].shape[0] == P1["xyz"].shape[0] == info["points"]
    for k in names:
        assert torch.allclose(got[k].detach(), P1[k], rtol=1e-6, atol=1e-7), k
        assert got[k].requires_grad and got[k].is_leaf
    for grp in gm.optimizer.param_groups:
        st = gm.optimizer.state[grp["params"][0]]
        assert grp["params"][0] is got[grp["name"]]
        assert torch.equal(st["exp_avg"], M1[grp["name"]][0]) and torch.equal(st["exp_avg_sq"], M1[grp["name"]][1])
        assert int(st["step"]) == 3
    assert gm.init_xyz.shape == got["xyz"].shape
    assert gm.xyz_gradient_accum.shape == (info["points"], 1) and float(gm.xyz_gradient_accum.abs().sum()) == 0.0
    assert gm.max_radii2D.shape == (info["points"],) and gm.denom.shape == (info["points"], 1)
    # the optimizer keeps working on the new tensors
    for t in gm._param_dict().values():
        t.grad = torch.ones_like(t)
    gm.optimizer.step()
    # reset_opacity: opacity <= 0.01 afterwards, only that group loses its moments
    before = gm.optimizer.state[gm._xyz]["exp_avg"].clone()
    gm.reset_opacity()
    assert float(gm.get_opacity.detach().max()) <= 0.01 + 1e-6
    assert float(gm.optimizer.state[gm._opacity]["exp_avg"].abs().sum()) == 0.0
    assert torch.equal(gm.optimizer.state[gm._xyz]["exp_avg"], before)
    # statistics accumulate like the reference's add_densification_stats + max_radii2D update
    m = gm._xyz.shape[0]
    vg, radii = torch.randn(m, 3), torch.randint(0, 5, (m,))
    gm.add_densification_stats(vg, radii > 0, radii)
    vis = radii > 0
    assert torch.allclose(gm.xyz_gradient_accum[vis], vg[vis, :2].norm(dim=-1, keepdim=True)) and float(gm.xyz_gradient_accum[~vis].abs().sum()) == 0
    assert torch.equal(gm.denom.squeeze(-1), vis.float()) and torch.equal(gm.max_radii2D, torch.where(vis, radii, torch.zeros_like(radii)).float())


def test_lazy_results_and_flat_grads_layout():
    """LazyResults (DiffRastRenderer's result dict): deferred entries are computed once, on first access, and behave like ordinary items;
    FlatGrads: parameter-shaped views alias one 16-byte-aligned buffer in order."""
    from MVs_Algorithms.DiffRastMesh.diff_mesh_renderer import LazyResults
    from c3d_hip.parallel import FlatGrads
    calls = []
    r = LazyResults()
    r["image"] = 1
    r.defer("depth", lambda: calls.append("d") or 42)
    r.defer("normal", lambda: calls.append("n") or 7)
    assert set(r.keys()) == {"image", "depth", "normal"} and "depth" in r and len(r) == 3 and calls == []
    assert r["depth"] == 42 and r["depth"] == 42 and calls == ["d"]
    assert r.get("normal") == 7 and r.get("missing", 5) == 5
    assert dict(r.items()) == {"image": 1, "depth": 42, "normal": 7} and calls == ["d", "n"]
    ps = [torch.zeros(7, *s) for s in ((3,), (1, 3), (15, 3), (1,), (3,), (4,))]
    fg = FlatGrads(ps)
    assert fg.flat.numel() % 4 == 0 and fg.flat.numel() >= sum(p.numel() for p in ps)
    off = 0
    for p, v in zip(ps, fg.views):
        assert v.shape == p.shape and v.data_ptr() == fg.flat.data_ptr() + 4 * off and (4 * off) % 16 == 0
        off += (p.numel() + 3) // 4 * 4
    fg.views[2].fill_(2.0)
    assert float(fg.flat.sum()) == 2.0 * ps[2].numel()
    fg.exchange(None, "allreduce")            # no process group: a no-op


def test_package_init_registers_its_own_nodes_when_comfyui_nodes_is_loaded():
    """inside ComfyUI `nodes` is already in sys.modules -- it is the module that loads custom nodes -- so the package must
    import ITS nodes.py relatively, and must not put itself in front of the host's modules on sys.path.  Load the package the way ComfyUI
    does (spec_from_file_location on the directory's __init__.py) with a dummy `nodes` pre-seeded."""
    import importlib.util
    import types
    dummy = types.ModuleType("nodes")
    dummy.NODE_CLASS_MAPPINGS = {"KSampler": object}
    dummy.NODE_DISPLAY_NAME_MAPPINGS = {"KSampler": "KSampler"}
    saved, saved_path = sys.modules.get("nodes"), list(sys.path)
    sys.modules["nodes"] = dummy
    try:
        sys.path[:] = [p for p in sys.path if os.path.abspath(p) != PKG]        # ComfyUI does not have the pack on its path beforehand
        sys.path.insert(0, "/tmp/comfyui_stand_in")                              # whatever the host has in front stays in front
        spec = importlib.util.spec_from_file_location("comfyui_3d_pack_amd_under_test", os.path.join(PKG, "__init__.py"), submodule_search_locations=[PKG])
        mod = importlib.util.module_from_spec(spec)
        sys.modules[spec.name] = mod
        spec.loader.exec_module(mod)
        assert "KSampler" not in mod.NODE_CLASS_MAPPINGS
        assert "[Comfy3D] Gaussian Splatting 3D" in mod.NODE_CLASS_MAPPINGS and len(mod.NODE_CLASS_MAPPINGS) == 8
        assert set(mod.NODE_DISPLAY_NAME_MAPPINGS) == set(mod.NODE_CLASS_MAPPINGS)
        assert sys.modules["nodes"] is dummy                                      # the host's module is untouched
        assert sys.path[0] == "/tmp/comfyui_stand_in" and os.path.abspath(sys.path[-1]) == PKG      # appended, as the reference's __init__.py:12
    finally:
        sys.path[:] = saved_path
        sys.modules.pop("comfyui_3d_pack_amd_under_test", None)
        sys.modules.pop("comfyui_3d_pack_amd_under_test.nodes", None)
        if saved is None:
            sys.modules.pop("nodes", None)
        else:
            sys.modules["nodes"] = saved


def test_diffmesh_validates_the_remesh_interval_up_front():
    """with the node defaults (1024 iterations, remesh every 512) and geometry training the periodic pymeshlab remesh would be due
    mid-run.  It is not part of this implementation: the constructor says so before any work is spent, and the run goes through."""
    import warnings
    src = open(os.path.join(PKG, "MVs_Algorithms", "DiffRastMesh", "diff_mesh.py")).read()
    assert "raise NotImplementedError" not in src                                 # nothing aborts 512 steps into a run any more
    ctor = src[src.index("class DiffMesh:"):src.index("def prepare_training")]
    assert "warnings.warn(" in ctor and "remesh_after_n_iteration < training_iterations" in ctor
    # the skip is also surfaced in the node's output (a ComfyUI "ui" text entry), not only as a Python warning
    node = open(os.path.join(PKG, "nodes.py")).read()
    body = node[node.index("def fitting_mesh"):node.index("NODE_CLASS_MAPPINGS")]
    assert "fitter.remesh_skipped" in body and '"ui"' in body and '"result": out' in body


# ---------------------------------------------------------------- ZeRO-1 and wider worlds
_SHAPES = ((3,), (1, 3), (15, 3), (1,), (3,), (4,))
_LRS = (1.6e-4, 2.5e-3, 1.25e-4, 0.05, 5e-3, 1e-3)


def _rank_grads(N, views, step):
    """stand-in for 'the gradient of this rank's views at this step': a function of the view ids only, so any sharding sums to the same total"""
    out = []
    for i, s in enumerate(_SHAPES):
        t = torch.zeros((N,) + s)
        for v in views:
            t += torch.randn((N,) + s, generator=torch.Generator().manual_seed(100000 * step + 1000 * v + i))
        out.append(t)
    return out


def _zero1_worker(rank, world, port, N, steps, n_views, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path[:0] = [root, os.path.join(root, "comfyui-3d-pack_amd")]
    from c3d_hip import parallel
    torch.manual_seed(0)
    params = [torch.nn.Parameter(torch.randn(N, *s)) for s in _SHAPES]
    parallel.broadcast_parameters(params, src=0)
    opt = torch.optim.Adam([{"params": [p], "lr": lr, "name": str(i)} for i, (p, lr) in enumerate(zip(params, _LRS))], lr=0.0, eps=1e-15)
    z = parallel.ZeroOneAdam(opt, params, None, average=True)
    assert z.chunk % 4 == 0 and z.chunk * world == z.total and all(g.data_ptr() % 16 == 0 for g in z.grads)
    mid = None
    for step in range(steps):
        mine = parallel.shard_views(list(range(n_views)), rank, world)      # n_views < world: some ranks have no views this step
        for dst, g in zip(z.grads, _rank_grads(N, mine, step)):
            dst.copy_(g)
        opt.param_groups[0]["lr"] = _LRS[0] * (0.9 ** step)                   # the LR schedule moves a group's lr between steps
        z.step()
        if step == steps // 2:                                                # densification / checkpoint boundary: whole moments out, a fresh object adopts them
            z.unshard()
            mid = [opt.state[p]["exp_avg"].clone() for p in params]
            z = parallel.ZeroOneAdam(opt, params, None, average=True)
            assert all(p not in opt.state for p in params) and z.t == step + 1
    z.unshard()
    out[rank] = ([p.data.clone() for p in params], [opt.state[p]["exp_avg"].clone() for p in params], [opt.state[p]["exp_avg_sq"].clone() for p in params], mid)
    dist.destroy_process_group()


@pytest.mark.parametrize("world,n_views", [(2, 8), (4, 8), (4, 3)])
def test_zero1_equals_replicated_adam_bit_for_bit(world, n_views):
    """ZeRO-1 (all-to-all reduce-scatter in rank order -> Adam on the owned slice -> all-gather of the parameters) over gloo, worlds 2 and 4, also with
    ranks that hold no view: parameters AND moments equal, bit for bit, those of the replicated step -- every rank summing all ranks' gradients in rank
    order (exchange mode 'allgather') and running Adam on every element -- including across an unshard / re-adopt boundary and a changing lr."""
    N, steps, port = 203, 5, _free_port()          # 59 * 203 elements: neither tensors nor slices end on chunk boundaries
    out = mp.Manager().dict()
    mp.spawn(_zero1_worker, args=(world, port, N, steps, n_views, out), nprocs=world, join=True)
    from c3d_hip import parallel
    torch.manual_seed(0)
    params = [torch.nn.Parameter(torch.randn(N, *s)) for s in _SHAPES]
    twin = [torch.nn.Parameter(p.detach().clone()) for p in params]          # the same run through torch.optim.Adam itself (its exp_avg update is a lerp_: equal to rounding)
    opt = torch.optim.Adam([{"params": [p], "lr": lr} for p, lr in zip(twin, _LRS)], lr=0.0, eps=1e-15)
    ms, vs = [torch.zeros_like(p) for p in params], [torch.zeros_like(p) for p in params]
    for step in range(steps):
        per_rank = [_rank_grads(N, list(range(n_views))[r::world], step) for r in range(world)]
        for i, (p, q) in enumerate(zip(params, twin)):
            tot = per_rank[0][i].clone()
            for r in range(1, world):
                tot += per_rank[r][i]
            g = tot * (1.0 / world)
            q.grad = g.clone()
            lr = _LRS[i] * (0.9 ** step) if i == 0 else _LRS[i]
            parallel.adam_reference_(p.data, g, ms[i], vs[i], lr, 0.9, 0.999, 1e-15, step + 1)     # the replicated step: every element, one process
        opt.param_groups[0]["lr"] = _LRS[0] * (0.9 ** step)
        opt.step()
    for p, q in zip(params, twin):
        assert torch.allclose(p.data, q.data, rtol=1e-5, atol=1e-7)                                # adam_reference_ IS torch.optim.Adam's rule
    for r in range(world):
        ps, m_r, v_r, mid = out[r]
        for a, b in zip(ps, params):
            assert torch.equal(a, b.data), "rank %d parameters" % r
        for a, b in zip(m_r, ms):
            assert torch.equal(a, b), "rank %d exp_avg" % r
        for a, b in zip(v_r, vs):
            assert torch.equal(a, b), "rank %d exp_avg_sq" % r
        assert mid is not None


def _world8_worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path[:0] = [root, os.path.join(root, "comfyui-3d-pack_amd")]
    from c3d_hip import parallel
    N = 1030
    params = [torch.nn.Parameter(torch.zeros(N, *s)) for s in _SHAPES]
    res = {}
    for n_views in (64, 5):                       # 8 per rank; then 5 views on 8 ranks: ranks 5-7 exchange zeros, and still issue the same sequence of ranges
        mine = parallel.shard_views(list(range(n_views)), rank, world)
        grads = _rank_grads(N, mine, 0)
        fa, fb, fc = parallel.FlatGrads(params), parallel.FlatGrads(params), parallel.FlatGrads(params)
        for f in (fa, fb, fc):
            for v_, g in zip(f.views, grads):
                v_.copy_(g)
        fa.exchange(None, "allgather")
        fb.exchange(None, "allreduce")
        for g0, g1 in ((0, 256), (256, 512), (512, 768), (768, N)):
            fc.exchange_rows(g0, g1)
        fc.exchange_finish()
        # a library all-reduce picks its own summation order per message, so beyond two ranks the chunked form equals the one-message form only to
        # rounding; what matters -- and is asserted by the parent -- is that each form leaves the SAME bits on every rank
        assert torch.allclose(fb.flat, fc.flat, rtol=1e-5, atol=1e-5) and torch.allclose(fa.flat, fb.flat, rtol=1e-5, atol=1e-5)
        res[n_views] = (fa.flat.clone(), fb.flat.clone(), mine, fc.flat.clone())
    out[rank] = res
    dist.destroy_process_group()


def test_exchange_world8_gloo_including_ranks_without_views():
    """The shape of the driver's 8-GPU run on CPU: eight gloo ranks, 8 views each (config 4) and then 5 views in total (three ranks have none).  All three
    exchange forms -- all-gather + rank-ordered sum, all-reduce, all-reduce by Gaussian ranges -- leave identical bits on every rank (replicas cannot drift);
    only the all-gather form has a summation order the caller controls: it equals the rank-ordered sum computed in one process bit for bit, the two
    all-reduce forms equal it (and each other) to rounding."""
    world, port = 8, _free_port()
    out = mp.Manager().dict()
    mp.spawn(_world8_worker, args=(world, port, out), nprocs=world, join=True)
    for n_views in (64, 5):
        views = [out[r][n_views][2] for r in range(world)]
        assert sorted(sum(views, [])) == list(range(n_views))
        for r in range(1, world):
            assert torch.equal(out[r][n_views][0], out[0][n_views][0]) and torch.equal(out[r][n_views][1], out[0][n_views][1]) and torch.equal(out[r][n_views][3], out[0][n_views][3])
        per_rank = [_rank_grads(1030, views[r], 0) for r in range(world)]
        tot = [per_rank[0][i].clone() for i in range(6)]
        for r in range(1, world):
            for i in range(6):
                tot[i] += per_rank[r][i]
        from c3d_hip import parallel
        ref = parallel.FlatGrads([torch.zeros_like(t) for t in tot])
        for v_, t in zip(ref.views, tot):
            v_.copy_(t)
        assert torch.equal(out[0][n_views][0], ref.flat)


def test_kiui_mesh_regularisers_against_literal_restatements():
    """laplacian_smooth_loss / normal_consistency (MVs_Algorithms/DiffRastMesh/diff_mesh.py; reference call site diff_mesh.py:126-127) against the literal
    form of the published kiui.mesh_utils source in float64 -- the dense uniform Laplacian matrix, the edge-to-face table filled edge by edge with its
    zero-initialised missing sides -- on an OPEN mesh (a grid patch: boundary edges, valences 2..6) and a closed one; values and gradients."""
    from MVs_Algorithms.DiffRastMesh.diff_mesh import edge_to_face_mapping, laplacian_smooth_loss, normal_consistency
    from c3d_hip import synthetic as S
    g = torch.Generator().manual_seed(3)
    n = 7
    yy, xx = torch.meshgrid(torch.arange(n), torch.arange(n), indexing="ij")
    v_open = torch.stack([xx.flatten().double(), yy.flatten().double(), 0.3 * torch.randn(n * n, generator=g).double()], 1)
    idx = lambda r, c: r * n + c
    f_open = torch.tensor([t for r in range(n - 1) for c in range(n - 1) for t in ([idx(r, c), idx(r + 1, c), idx(r, c + 1)], [idx(r, c + 1), idx(r + 1, c), idx(r + 1, c + 1)])], dtype=torch.int32)
    vs, fs, _, _ = S.make_uv_sphere(6, 9)
    for v0, f in ((v_open, f_open), (torch.tensor(vs).double() + 0.02 * torch.randn(vs.shape, generator=g).double(), torch.tensor(fs))):
        V, fl = v0.shape[0], f.long()
        # literal laplacian_uniform: L = D - A over the unique edge set
        A = torch.zeros((V, V), dtype=torch.float64)
        for t in fl.tolist():
            for a, b in ((t[0], t[1]), (t[1], t[2]), (t[2], t[0])):
                A[a, b] = A[b, a] = 1.0
        L = torch.diag(A.sum(1)) - A
        # literal compute_edge_to_face_mapping: zeros, column by walking direction
        edges = {}
        for ti, t in enumerate(fl.tolist()):
            for a, b in ((t[0], t[1]), (t[1], t[2]), (t[2], t[0])):
                edges.setdefault((min(a, b), max(a, b)), [0, 0])[1 if a > b else 0] = ti
        keys = sorted(edges)
        tpe_ref = torch.tensor([edges[k] for k in keys])
        assert torch.equal(edge_to_face_mapping(f), tpe_ref)
        has_boundary = any(sum(1 for t in fl.tolist() if (k[0] in t and k[1] in t)) == 1 for k in keys)
        v = v0.clone().requires_grad_(True)
        w = v0.clone().requires_grad_(True)
        l1 = laplacian_smooth_loss(v, f)
        l1_ref = (L @ w).norm(dim=1).mean()
        fn = torch.cross(w[fl[:, 1]] - w[fl[:, 0]], w[fl[:, 2]] - w[fl[:, 0]], dim=-1)
        fn = fn / torch.sqrt(torch.clamp((fn * fn).sum(-1, keepdim=True), min=1e-20))
        l2 = normal_consistency(v, f)
        l2_ref = (1.0 - torch.clamp((fn[tpe_ref[:, 0]] * fn[tpe_ref[:, 1]]).sum(-1), -1.0, 1.0)).abs().mean()
        assert abs(float(l1) - float(l1_ref)) <= 1e-12 * max(1.0, abs(float(l1_ref))) and abs(float(l2) - float(l2_ref)) <= 1e-12
        (l1 + l2).backward(); (l1_ref + l2_ref).backward()
        assert torch.allclose(v.grad, w.grad, rtol=1e-10, atol=1e-12)
        assert float(l1) > 0 and float(l2) > 0
        if f is f_open:
            assert has_boundary


def test_drop_in_rasterizer_capacity_bookkeeping():
    """host side of the sync-free drop-in forward (diff_gaussian_rasterization/__init__.py): what is learnt per (device, H, W) shape, how a point count not seen yet is
    estimated, the launch hint and the buffer capacity of a call, and who counts as differentiated (grad MODE, not only requires_grad -- ADVICE r5)"""
    import diff_gaussian_rasterization as dgr
    key = (0, 1080, 1920)
    saved, dgr._learnt = dgr._learnt, {}
    try:
        assert dgr._capacity_for(key, 1000000) is None                       # nothing learnt: the synchronous path
        dgr._learn(key, 1000000, 4000000)
        dgr._learn(key, 1000000, 3900000)                                      # the largest count stays
        assert dgr._learnt[key][1000000] == 4000000 and dgr.last_num_rendered == 3900000
        first, cap = dgr._capacity_for(key, 1000000)
        assert first == int(4000000 * dgr._HEADROOM) + dgr._SLACK and cap == dgr._ROOM * first
        # a model that densified (+10 %) or was pruned (-30 %): the estimate follows the point count; beyond a factor of two it is another model -> learn again
        assert dgr._estimate(key, 1100000) == 4400000 and dgr._estimate(key, 700000) == 2800000
        assert dgr._estimate(key, 2000001) is None and dgr._estimate(key, 499999) is None and dgr._capacity_for(key, 10000) is None
        # no view holds more pairs than N x tiles: a small scene gets launches and buffers of that bound (hint 0)
        dgr._learn((0, 64, 64), 100, 900)
        assert dgr._capacity_for((0, 64, 64), 100) == (0, 100 * 16)
        # at most _MODELS point counts per shape, least recently used first out
        for n in range(20):
            dgr._learn(key, 3000000 + n, 5)
        assert len(dgr._learnt[key]) == dgr._MODELS and 1000000 not in dgr._learnt[key]
        was = dgr.sync_free(False)
        try:
            assert dgr._capacity_for(key, 3000001) is None
            assert dgr.sync_free("unverified") is False and dgr.sync_free(True) == "unverified" and dgr.sync_free(False) == "verified"
        finally:
            dgr.sync_free(was)
        p = torch.nn.Parameter(torch.zeros(3))
        assert dgr._differentiated(torch.zeros(3), None, p)
        with torch.no_grad():
            assert not dgr._differentiated(p)
        assert not dgr._differentiated(torch.zeros(3), None)
    finally:
        dgr._learnt = saved
