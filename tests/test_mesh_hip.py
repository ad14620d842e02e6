"""Parity tests (-m gpu) of the HIP mesh ops behind `nvdiffrast.torch`, through the C-ABI, against oracle/mesh_oracle.c.
Integers (triangle ids) exact up to a vanishing number of depth near-ties; images L1 <= 1e-4; gradients <= 1e-3 relative."""
import ctypes as C
import os

import numpy as np
import pytest
import torch

from c3d_hip import synthetic as S
from oracle import mesh_oracle as M
from helpers import assert_grad_close, rel_err

pytestmark = pytest.mark.gpu
IMG_L1 = 1e-4
GRAD_REL = 1e-3


@pytest.fixture(scope="module", autouse=True)
def _need_gpu():
    if not torch.cuda.is_available():
        pytest.fail("-m gpu tests need a HIP device (no CPU fallback exists)")
    M.build()
    import c3d_hip
    c3d_hip.lib()


def T(a, dtype=torch.float32, grad=False):
    return torch.tensor(np.asarray(a), dtype=dtype, device="cuda", requires_grad=grad)


def _scene(H, W, n_lat, n_lon, el=-20.0, az=35.0, rad=2.0):
    v, f, vt, vn = S.make_uv_sphere(n_lat, n_lon, radius=0.7, displacement=0.15)
    pos, vcam, pose = S.mesh_clip_positions(v, el, az, rad, W, H)
    return pos, f, vt, vn


SCENES = [(48, 64, 10, 16), (130, 97, 24, 40), (256, 256, 64, 128)]


@pytest.mark.parametrize("sc", SCENES)
def test_rasterize_matches_oracle(sc):
    import nvdiffrast.torch as dr
    H, W, nl, no = sc
    pos, f, vt, vn = _scene(H, W, nl, no)
    ctx = dr.RasterizeCudaContext()
    rast, db = dr.rasterize(ctx, T(pos), T(f, torch.int32), (H, W))
    orast, odb = M.rasterize(pos, f, (H, W))
    r = rast.cpu().numpy()
    same = r[..., 3] == orast[..., 3]
    assert (~same).sum() <= max(2, int(2e-5 * H * W)), (~same).sum()
    # u,v come out of differences of products of clip coordinates: float32 cancellation on ~1-pixel triangles (FMA contraction
    # on the GPU, none in the oracle) -> compare the mean tightly and the max loosely
    assert np.abs(r[same][:, :3] - orast[same][:, :3]).mean() <= 1e-5
    assert np.abs(r[same][:, :3] - orast[same][:, :3]).max() <= 1e-3
    assert np.abs(db.cpu().numpy()[same] - odb[same]).max() <= 1e-3 * max(1.0, np.abs(odb).max())
    # determinism (atomicMin on a packed word is order independent)
    rast2, _ = dr.rasterize(ctx, T(pos), T(f, torch.int32), (H, W))
    assert torch.equal(rast, rast2)


def test_rasterize_large_triangles_and_fill_rule():
    """two triangles covering the screen: the workgroup-per-triangle path; every pixel exactly once, also on the diagonal"""
    import nvdiffrast.torch as dr
    pos = np.array([[[-1, -1, 0, 1], [1, -1, 0, 1], [1, 1, 0, 1], [-1, 1, 0, 1]]], np.float32)
    tri = np.array([[0, 1, 2], [0, 2, 3]], np.int32)
    ctx = dr.RasterizeCudaContext()
    for (H, W) in [(64, 64), (100, 37), (1080, 1920)]:
        rast, db = dr.rasterize(ctx, T(pos), T(tri, torch.int32), (H, W))
        orast, odb = M.rasterize(pos, tri, (H, W))
        assert (rast[..., 3] > 0).all()
        assert np.array_equal(rast.cpu().numpy()[..., 3], orast[..., 3])
        assert np.abs(rast.cpu().numpy() - orast).max() < 1e-5
    # batch of 2 with different positions, a back-facing and a degenerate triangle, a vertex behind the camera
    pos2 = np.stack([pos[0], pos[0] * np.array([0.5, 0.5, 1, 1], np.float32)])
    tri2 = np.array([[0, 2, 1], [0, 0, 3], [0, 2, 3]], np.int32)
    rast, _ = dr.rasterize(ctx, T(pos2), T(tri2, torch.int32), (32, 32))
    orast, _ = M.rasterize(pos2, tri2, (32, 32))
    assert np.array_equal(rast.cpu().numpy()[..., 3], orast[..., 3])
    posn = pos.copy(); posn[0, 2, 3] = -1.0
    rast, _ = dr.rasterize(ctx, T(posn), T(tri, torch.int32), (16, 16))
    assert (rast[..., 3] == 0).all()
    # empty mesh -> background
    rast, _ = dr.rasterize(ctx, T(pos), torch.zeros((0, 3), dtype=torch.int32, device="cuda"), (8, 8))
    assert (rast == 0).all()


@pytest.mark.parametrize("sc", SCENES[:2])
def test_full_pipeline_forward_and_gradients(sc):
    """rasterize -> interpolate(uv, diff all) -> texture -> antialias, the op sequence of DiffRastRenderer.render (:97-138)"""
    import nvdiffrast.torch as dr
    H, W, nl, no = sc
    pos, f, vt, vn = _scene(H, W, nl, no)
    rng = np.random.default_rng(1)
    tex = rng.normal(size=(1, 32, 32, 3)).astype(np.float32)
    ctx = dr.RasterizeCudaContext()
    tpos, ttri, tvt, ttex = T(pos, grad=True), T(f, torch.int32), T(vt[None], grad=True), T(tex, grad=True)
    rast, db = dr.rasterize(ctx, tpos, ttri, (H, W))
    texc, texd = dr.interpolate(tvt, rast, ttri, rast_db=db, diff_attrs='all')
    col = dr.texture(ttex, texc, uv_da=texd, filter_mode='linear')
    aa = dr.antialias(col, rast, tpos, ttri)
    alpha = dr.antialias(torch.clamp(rast[..., -1:], 0, 1).contiguous(), rast, tpos, ttri)
    # oracle (float32 forward for images)
    orast, odb = M.rasterize(pos, f, (H, W))
    otexc, otexd = M.interpolate(vt[None], orast, f, odb, "all")
    ocol = M.texture(tex, otexc)
    oaa = M.antialias(ocol, orast, pos, f)
    oalpha = M.antialias(np.clip(orast[..., 3:], 0, 1), orast, pos, f)
    assert (rast.detach().cpu().numpy()[..., 3] != orast[..., 3]).sum() <= 2
    assert np.abs(texc.detach().cpu().numpy() - otexc).mean() <= IMG_L1
    assert np.abs(texd.detach().cpu().numpy() - otexd).mean() <= IMG_L1
    assert np.abs(col.detach().cpu().numpy() - ocol).mean() <= IMG_L1
    assert np.abs(aa.detach().cpu().numpy() - oaa).mean() <= IMG_L1
    assert np.abs(alpha.detach().cpu().numpy() - oalpha).mean() <= IMG_L1
    # gradients: float64 oracle chain rule, same upstream gradients
    gA = rng.normal(size=oaa.shape).astype(np.float32); gB = rng.normal(size=oalpha.shape).astype(np.float32)
    ((aa * T(gA)).sum() + (alpha * T(gB)).sum()).backward()
    d = np.float64
    r64, db64 = M.rasterize(pos, f, (H, W), dtype=d)
    texc64, _ = M.interpolate(vt[None], r64, f, db64, "all", dtype=d)
    col64 = M.texture(tex, texc64, dtype=d)
    dcol, dpos_aa = M.antialias_bwd(col64, r64, pos, f, gA, dtype=d)
    _, dpos_al = M.antialias_bwd(np.clip(r64[..., 3:], 0, 1), r64, pos, f, gB, dtype=d)
    dtex, duv = M.texture_bwd(tex, texc64, dcol, dtype=d)
    dvt, drast = M.interpolate_bwd(vt[None], r64, f, duv, dtype=d)
    dpos_r = M.rasterize_bwd(pos, f, r64, drast, dtype=d)
    assert rel_err(ttex.grad.cpu().numpy(), dtex) <= GRAD_REL
    assert rel_err(tvt.grad.cpu().numpy(), dvt) <= GRAD_REL
    assert rel_err(tpos.grad.cpu().numpy(), dpos_aa + dpos_al + dpos_r) <= 2 * GRAD_REL


def test_baseline_config5_499k_triangles_1024px_vs_float64_oracle():
    """BASELINE config 5 at FULL size: one view of the 499,000-triangle displaced sphere, 1024 x 1024, 1024^2 x 3 texture,
    rasterize -> interpolate(uv, diff all) -> texture(linear) -> antialias(colour) + antialias(alpha), forward against the float32 oracle and
    backward against the float64 oracle's chain rule.  ~2 pixels per triangle: the regime the one-lane-per-triangle rasterizer is built for."""
    import nvdiffrast.torch as dr
    H = W = 1024
    v, f, vt, vn = S.make_uv_sphere(500, 500, radius=0.7, displacement=0.05)
    pos, vcam, pose = S.mesh_clip_positions(v, -20.0, 45.0, 2.0, W, H)
    rng = np.random.default_rng(1)
    # Texture: band-limited (a few sinusoids per channel), not config 5's white noise.  d(colour)/d(uv) of a bilinear fetch is discontinuous
    # across texel boundaries, and with ~1 texel per pixel some hundred of the 560k covered pixels have their float32 and float64 uv in
    # different texels: on white noise such a pixel's uv gradient is simply a different number (measured: 10 % max-norm on dL/dvt), which says
    # nothing about the kernels.  Images and dL/dtex are insensitive to this (bilinear weights are continuous); a smooth texture makes the
    # uv gradient comparable as well.
    ty, tx = np.meshgrid(np.arange(1024) / 1024.0, np.arange(1024) / 1024.0, indexing="ij")
    tex = np.stack([np.sin(2 * np.pi * (3 * tx + 2 * ty)) + 0.5 * np.cos(2 * np.pi * 5 * ty), np.cos(2 * np.pi * (2 * tx - 4 * ty)),
                    np.sin(2 * np.pi * 6 * tx) * np.cos(2 * np.pi * 3 * ty)], -1)[None].astype(np.float32)
    ctx = dr.RasterizeCudaContext()
    tpos, ttri, tvt, ttex = T(pos, grad=True), T(f, torch.int32), T(vt[None], grad=True), T(tex, grad=True)
    rast, db = dr.rasterize(ctx, tpos, ttri, (H, W))
    texc, texd = dr.interpolate(tvt, rast, ttri, rast_db=db, diff_attrs='all')
    col = dr.texture(ttex, texc, uv_da=texd, filter_mode='linear')
    aa = dr.antialias(col, rast, tpos, ttri)
    alpha = dr.antialias(torch.clamp(rast[..., -1:], 0, 1).contiguous(), rast, tpos, ttri)
    orast, odb = M.rasterize(pos, f, (H, W))
    r = rast.detach().cpu().numpy()
    same = r[..., 3] == orast[..., 3]
    n_diff = int((~same).sum())
    print("[mesh config5] covered %.3f of the pixels, %d triangle ids differ" % (float((orast[..., 3] > 0).mean()), n_diff))
    assert n_diff <= max(2, int(2e-5 * H * W)), n_diff            # depth near-ties between neighbouring ~1-pixel triangles
    assert np.abs(r[same][:, :3] - orast[same][:, :3]).mean() <= 1e-5
    # downstream images against the oracle fed with the oracle's own rast (ids that differ move single pixels, covered by the L1 bounds)
    otexc, otexd = M.interpolate(vt[None], orast, f, odb, "all")
    ocol = M.texture(tex, otexc)
    oaa = M.antialias(ocol, orast, pos, f)
    oalpha = M.antialias(np.clip(orast[..., 3:], 0, 1), orast, pos, f)
    assert np.abs(texc.detach().cpu().numpy() - otexc).mean() <= IMG_L1
    assert np.abs(col.detach().cpu().numpy() - ocol).mean() <= IMG_L1
    assert np.abs(aa.detach().cpu().numpy() - oaa).mean() <= IMG_L1
    assert np.abs(alpha.detach().cpu().numpy() - oalpha).mean() <= IMG_L1
    gA = rng.normal(size=oaa.shape).astype(np.float32); gB = rng.normal(size=oalpha.shape).astype(np.float32)
    ((aa * T(gA)).sum() + (alpha * T(gB)).sum()).backward()
    d = np.float64
    r64, db64 = M.rasterize(pos, f, (H, W), dtype=d)
    texc64, _ = M.interpolate(vt[None], r64, f, db64, "all", dtype=d)
    col64 = M.texture(tex, texc64, dtype=d)
    dcol, dpos_aa = M.antialias_bwd(col64, r64, pos, f, gA, dtype=d)
    _, dpos_al = M.antialias_bwd(np.clip(r64[..., 3:], 0, 1), r64, pos, f, gB, dtype=d)
    dtex, duv = M.texture_bwd(tex, texc64, dcol, dtype=d)
    dvt, drast = M.interpolate_bwd(vt[None], r64, f, duv, dtype=d)
    dpos_r = M.rasterize_bwd(pos, f, r64, drast, dtype=d)
    assert rel_err(ttex.grad.cpu().numpy(), dtex) <= GRAD_REL
    # A pixel whose triangle id differs (depth near-tie; n_diff above) hands its whole gradient to other vertices, by construction: the vertices of
    # both candidate triangles of such pixels and of their 8 neighbours (antialias pairs) are left out of the per-vertex comparisons.
    keep = np.ones(v.shape[0], bool)
    ys, xs = np.nonzero(~same[0])
    for y, x in zip(ys, xs):
        for dy in (-1, 0, 1):
            for dx in (-1, 0, 1):
                yy, xx = min(max(y + dy, 0), H - 1), min(max(x + dx, 0), W - 1)
                for ids in (r[0, yy, xx, 3], orast[0, yy, xx, 3]):
                    if ids > 0:
                        keep[f[int(ids) - 1]] = False
    print("[mesh config5] %d vertices excluded around %d differing pixels" % (int((~keep).sum()), n_diff))
    gvt, gpos, rpos = tvt.grad.cpu().numpy()[0], tpos.grad.cpu().numpy()[0], (dpos_aa + dpos_al + dpos_r)[0]
    assert_grad_close(ttex.grad.cpu().numpy(), dtex, "config5 dL/dtex", rel_l2=2e-3, max_frac=5e-3, hard=1e9)
    assert_grad_close(gvt[keep], dvt[0][keep], "config5 dL/dvt", rel_l2=2e-3, max_frac=5e-3, hard=1e9)
    assert_grad_close(gpos[keep], rpos[keep], "config5 dL/dpos", rel_l2=5e-3, max_frac=2e-2, hard=1e9)
    # max-norm at 3e-3 / 4e-3 instead of 1e-3 / 2e-3 (the small scenes above hold those): a pixel whose float32 uv sits in the neighbouring texel of
    # its float64 uv still changes that pixel's uv gradient by (texture curvature x one texel), measured 1.5e-3 of the largest entry here
    assert rel_err(gvt[keep], dvt[0][keep]) <= 3 * GRAD_REL
    assert rel_err(gpos[keep], rpos[keep]) <= 4 * GRAD_REL


def test_texture_modes_and_batches():
    import nvdiffrast.torch as dr
    rng = np.random.default_rng(3)
    tex = rng.normal(size=(2, 17, 23, 4)).astype(np.float32)
    uv = rng.uniform(-1.5, 2.5, size=(2, 31, 29, 2)).astype(np.float32)
    for fm in ("linear", "nearest"):
        for bm in ("wrap", "clamp"):
            out = dr.texture(T(tex), T(uv), filter_mode=fm, boundary_mode=bm)
            ref = M.texture(tex, uv, fm, bm)
            assert np.abs(out.cpu().numpy() - ref).max() <= 2e-5, (fm, bm)
    out = dr.texture(T(tex[:1]), T(uv), filter_mode="linear")            # texture batch 1 broadcast over uv batch 2
    assert np.abs(out.cpu().numpy() - M.texture(tex[:1], uv)).max() <= 2e-5
    tt, tu = T(tex, grad=True), T(uv, grad=True)
    g = rng.normal(size=(2, 31, 29, 4)).astype(np.float32)
    (dr.texture(tt, tu, filter_mode="linear") * T(g)).sum().backward()
    dtex, duv = M.texture_bwd(tex, uv, g, dtype=np.float64)
    assert rel_err(tt.grad.cpu().numpy(), dtex) <= GRAD_REL and rel_err(tu.grad.cpu().numpy(), duv) <= GRAD_REL
    with pytest.raises(ValueError):
        dr.texture(T(tex), T(uv), filter_mode="linear-mipmap-linear")       # mip-mapped modes need uv_da or mip_level_bias
    # 'zero': composed in the shim from a zero-padded texture fetched in 'clamp' mode (the composition itself is checked on the CPU)
    tz, uz = T(tex, grad=True), T(uv, grad=True)
    out = dr.texture(tz, uz, filter_mode="linear", boundary_mode="zero")
    assert np.abs(out.detach().cpu().numpy() - M.texture(tex, uv, "linear", "zero")).max() <= 5e-5
    (out * T(g)).sum().backward()
    dtex, duv = M.texture_bwd(tex, uv, g, "linear", "zero", dtype=np.float64)
    assert rel_err(tz.grad.cpu().numpy(), dtex) <= GRAD_REL and rel_err(uz.grad.cpu().numpy(), duv) <= GRAD_REL
    with pytest.raises(NotImplementedError):
        dr.texture(T(tex), T(uv), boundary_mode="cube")


def test_interpolate_variants():
    import nvdiffrast.torch as dr
    H, W = 64, 80
    pos, f, vt, vn = _scene(H, W, 12, 20)
    ctx = dr.RasterizeCudaContext()
    rast, db = dr.rasterize(ctx, T(pos), T(f, torch.int32), (H, W))
    orast = rast.cpu().numpy(); odb = db.cpu().numpy()
    # [V,A] attributes without batch dim, no differentials (depth / normal calls of the reference :110,131)
    out, da = dr.interpolate(T(vn), rast, T(f, torch.int32))
    assert da.numel() == 0 and out.shape == (1, H, W, 3)
    assert np.abs(out.cpu().numpy() - M.interpolate(vn, orast, f)[0]).max() <= 1e-5
    # subset of differentiated attributes
    out, da = dr.interpolate(T(vn[None]), rast, T(f, torch.int32), rast_db=db, diff_attrs=[2, 0])
    oo, oda = M.interpolate(vn[None], orast, f, odb, [2, 0])
    assert da.shape == (1, H, W, 4) and np.abs(da.cpu().numpy() - oda).max() <= 1e-4


def test_edge_cases_and_errors():
    import nvdiffrast.torch as dr
    ctx = dr.RasterizeCudaContext()
    pos = torch.zeros((1, 3, 4))
    with pytest.raises(RuntimeError, match="HIP device"):
        dr.rasterize(ctx, pos, torch.zeros((1, 3), dtype=torch.int32), (8, 8))
    with pytest.raises(ValueError):      # range mode takes one shared vertex buffer [V,4]
        dr.rasterize(ctx, pos.cuda(), torch.zeros((1, 3), dtype=torch.int32).cuda(), (8, 8), ranges=torch.zeros((1, 2), dtype=torch.int32))
    assert isinstance(dr.RasterizeGLContext(), dr.RasterizeCudaContext)
    # antialias is the identity when there is no triangle-id discontinuity (full-screen single triangle)
    p = T(np.array([[[-1, -1, 0, 1], [3, -1, 0, 1], [-1, 3, 0, 1]]], np.float32))
    t = T(np.array([[0, 1, 2]]), torch.int32)
    rast, _ = dr.rasterize(ctx, p, t, (16, 16))
    col = torch.rand((1, 16, 16, 3), device="cuda")
    assert torch.equal(dr.antialias(col, rast, p, t), col)
    with torch.inference_mode():
        r, _ = dr.rasterize(ctx, p, t, (8, 8))
    assert (r[..., 3] == 1).all()


def _torch_mesh(n_lat=48, n_lon=96, tex=64, device="cuda"):
    from mesh_processer.mesh import Mesh
    v, f, vt, vn = S.make_uv_sphere(n_lat, n_lon, radius=0.7, displacement=0.1)
    m = Mesh(v=T(v), f=T(f, torch.int32), vt=T(vt), ft=T(f, torch.int32), device=device)
    m.auto_normal()
    yy, xx = np.meshgrid(np.arange(tex), np.arange(tex), indexing="ij")
    chk = (((yy // 8) + (xx // 8)) % 2).astype(np.float32)
    m.albedo = T(np.stack([0.2 + 0.6 * chk, 0.8 - 0.6 * chk, np.full_like(chk, 0.5)], -1))
    return m


def test_diffrast_renderer_mirror_and_trainer():
    """DiffRastRenderer.render via the camera controller against the oracle op chain, then a few fitting steps."""
    from MVs_Algorithms.DiffRastMesh.diff_mesh import DiffMesh, DiffMeshCameraController
    from MVs_Algorithms.DiffRastMesh.diff_mesh_renderer import DiffRastRenderer
    from shared_utils.camera_utils import OrbitCamera, orbit_camera
    H = W = 192
    mesh = _torch_mesh()
    r = DiffRastRenderer(mesh, force_cuda_rast=True).cuda()
    ctl = DiffMeshCameraController(r, W, H, 49.1, static_bg=[1.0, 1.0, 1.0], device="cuda")
    poses = [[2.0, -20.0, az, 0.0, 0.0, 0.0] for az in (0.0, 90.0, 180.0, -90.0)]
    with torch.no_grad():
        imgs, masks, extra = ctl.render_all_pose(poses)
    assert imgs.shape == (4, H, W, 3) and masks.shape == (4, H, W, 1)
    assert set(extra) >= {"image", "alpha", "depth", "normal", "viewcos"}
    # oracle chain for view 1
    cam = OrbitCamera(W, H, fovy=49.1)
    pose = orbit_camera(-20.0, 90.0, 2.0)
    v = mesh.v.cpu().numpy(); f = mesh.f.cpu().numpy(); vt = mesh.vt.cpu().numpy()
    vh = np.concatenate([v, np.ones((v.shape[0], 1), np.float32)], 1)
    v_cam = (vh @ np.linalg.inv(pose).T.astype(np.float32)).astype(np.float32)
    v_clip = (v_cam @ cam.perspective.T).astype(np.float32)[None]
    orast, odb = M.rasterize(v_clip, f, (H, W))
    oalpha = np.clip(M.antialias(np.clip(orast[..., 3:], 0, 1), orast, v_clip, f), 0, 1)[0]
    otexc, _ = M.interpolate(vt[None], orast, f, odb, "all")
    raw = r.raw_albedo.detach().cpu().numpy()[None]
    oalb = 1 / (1 + np.exp(-M.texture(raw, otexc)))
    oalb = M.antialias(oalb, orast, v_clip, f)[0]
    oimg = np.clip(oalpha * oalb + (1 - oalpha) * 1.0, 0, 1)
    assert np.abs(imgs[1].cpu().numpy() - oimg).mean() <= IMG_L1
    assert np.abs(masks[1].cpu().numpy() - oalpha).mean() <= IMG_L1
    odepth, _ = M.interpolate(-v_cam[None][..., 2:3], orast, f)
    assert np.abs(extra["depth"][1].cpu().numpy() - odepth[0]).mean() <= IMG_L1
    # fitting: start from a grey texture (and slightly wrong geometry) and recover the checker renders
    ref_images = [imgs[i].cpu() for i in range(4)]
    ref_masks = [masks[i, ..., 0].cpu() for i in range(4)]
    m2 = _torch_mesh()
    m2.albedo = None                                                # set_new_albedo resizes an existing texture (as the reference's does): drop it first
    m2.set_new_albedo(64, 64)
    with torch.no_grad():
        m2.v *= 1.03
    tr = DiffMesh(m2, training_iterations=40, batch_size=2, texture_learning_rate=0.05, train_mesh_geometry=True, geometry_learning_rate=2e-4,
                  ms_ssim_loss_weight=0.2, remesh_after_n_iteration=10 ** 9, invert_bg_prob=1.0, force_cuda_rasterize=True)
    tr.cam_controller = None
    tr.prepare_training(ref_images, ref_masks, poses, 49.1)
    tr.cam_controller.static_bg = torch.ones(3, device="cuda")
    losses = [tr.training_step(s, [s % 4, (s + 2) % 4]).item() for s in range(40)]
    assert np.isfinite(losses).all() and np.mean(losses[-5:]) < 0.7 * np.mean(losses[:5]), losses
    assert torch.isfinite(tr.renderer.v_offsets).all() and tr.renderer.v_offsets.abs().max().item() > 0


@pytest.mark.parametrize("lam,geo,lanes", [(0.0, True, 1), (0.2, True, 3), (0.5, False, 4)])
def test_diffmesh_fused_step_equals_per_view_autograd_step(lam, geo, lanes):
    """DiffMesh.training_step as ONE library call (c3d_mesh_train_views: render, image loss incl. MS-SSIM, backward of every view on view lanes,
    gradients summed over the views) against the per-view autograd path of the same trainer: same loss, same gradients (the texture gradient sums
    with float atomics on both sides: equal to rounding), same per-view background draws, same parameters after the Adam step."""
    from MVs_Algorithms.DiffRastMesh.diff_mesh import DiffMesh, DiffMeshCameraController
    from MVs_Algorithms.DiffRastMesh.diff_mesh_renderer import DiffRastRenderer
    H = W = 192
    mesh = _torch_mesh()
    r0 = DiffRastRenderer(mesh, force_cuda_rast=True).cuda()
    ctl = DiffMeshCameraController(r0, W, H, 49.1, static_bg=[1.0, 1.0, 1.0], device="cuda")
    poses = [[2.0, -20.0, az, 0.0, 0.0, 0.0] for az in (0.0, 90.0, 180.0, -90.0, 45.0)]
    with torch.no_grad():
        imgs, masks, _ = ctl.render_all_pose(poses)
    rng = np.random.default_rng(5)
    ref_images = [(imgs[i].cpu() * 0.8 + 0.1 * torch.tensor(rng.uniform(size=(H, W, 3)).astype(np.float32))) for i in range(5)]
    ref_masks = [torch.tensor((rng.uniform(size=(H, W)) * 0.5 + 0.5).astype(np.float32)) * masks[i, ..., 0].cpu() for i in range(5)]     # soft masks
    out = []
    for fused in (False, True):
        m2 = _torch_mesh()
        m2.albedo = None
        m2.set_new_albedo(64, 64)
        tr = DiffMesh(m2, training_iterations=3, batch_size=5, texture_learning_rate=0.05, train_mesh_geometry=geo, geometry_learning_rate=2e-4,
                      ms_ssim_loss_weight=lam, remesh_after_n_iteration=10 ** 9, invert_bg_prob=0.5, force_cuda_rasterize=True)
        tr.use_fused_step, tr.view_lanes = fused, lanes
        tr.prepare_training(ref_images, ref_masks, poses, 49.1)
        assert tr._can_fuse() == fused
        with torch.no_grad():
            tr.renderer.raw_albedo.add_(torch.tensor(np.random.default_rng(1).normal(size=tuple(tr.renderer.raw_albedo.shape)).astype(np.float32), device="cuda"))
        np.random.seed(11)                                           # the background of every view: one np.random draw per view, in view order
        grads = {}
        hooks = [q.register_hook(lambda g, k=k: grads.__setitem__(k, g.clone())) for k, q in (("albedo", tr.renderer.raw_albedo), ("offsets", tr.renderer.v_offsets))] if not fused else []
        losses = []
        for s_ in range(2):
            if fused:                                                  # the kernels write .grad directly: read it before the optimizer clears it
                opt_step = tr.optimizer.step
                def spy(*a_, _o=opt_step, **k_):
                    grads["albedo"] = tr.renderer.raw_albedo.grad.clone()
                    if geo:
                        grads["offsets"] = tr.renderer.v_offsets.grad.clone()
                    return _o(*a_, **k_)
                tr.optimizer.step = spy
            losses.append(tr.training_step(s_, [0, 1, 2, 3, 4]).item())
            if s_ == 0:
                first = {k: v.clone() for k, v in grads.items()}
        for h in hooks:
            h.remove()
        out.append((losses, first, tr.renderer.raw_albedo.detach().clone(), tr.renderer.v_offsets.detach().clone()))
    (l0, g0, a0, o0), (l1, g1, a1, o1) = out
    print("[mesh step] lam %.1f geo %s lanes %d: losses %s vs %s" % (lam, geo, lanes, l0, l1))
    assert np.allclose(l0, l1, rtol=2e-4, atol=1e-6)
    assert rel_err(g1["albedo"].cpu().numpy(), g0["albedo"].cpu().numpy()) <= 1e-3
    if geo:
        assert rel_err(g1["offsets"].cpu().numpy(), g0["offsets"].cpu().numpy()) <= 2e-3
        assert float(o0.abs().max()) > 0 and torch.allclose(o0, o1, atol=3e-4)      # Adam turns tiny gradient differences near zero into +-lr steps (lr 2e-4)
    assert torch.allclose(a0, a1, atol=0.11) and float((a0 - a1).abs().mean()) <= 2e-3


def _small_mesh_step_case():
    """three views (two backgrounds, masks of all kinds) of the test mesh with a 64^2 texture and vertex offsets, 192^2"""
    from shared_utils.camera_utils import OrbitCamera, orbit_camera
    import nvdiffrast.torch as dr
    H = W = 192
    mesh = _torch_mesh()
    mesh.albedo = None
    mesh.set_new_albedo(64, 64)
    g = torch.Generator(device="cpu").manual_seed(11)
    raw = (torch.randn((64, 64, 3), generator=g) * 0.5).cuda()
    off = (torch.randn(tuple(mesh.v.shape), generator=g) * 1e-3).cuda()
    cam = OrbitCamera(W, H, fovy=49.1)
    proj = cam.perspective.astype(np.float32)
    views = [((proj @ np.linalg.inv(orbit_camera(-20.0, az, 2.0).astype(np.float32)).astype(np.float32)).astype(np.float32), bg) for az, bg in ((0.0, (1, 1, 1)), (100.0, (0, 0, 0)), (-130.0, (1, 1, 1)))]
    rng = np.random.default_rng(3)
    targets = [T(rng.uniform(size=(3, H, W)).astype(np.float32)) for _ in views]
    masks = [T(rng.uniform(0.3, 1.0, size=(1, H, W)).astype(np.float32)), None, T(np.ones((1, H, W), np.float32))]
    f, ft, vt = mesh.f.to(torch.int32).contiguous(), mesh.ft.to(torch.int32).contiguous(), mesh.vt.float().contiguous()
    return H, W, mesh, raw, off, views, targets, masks, f, ft, vt, dr.RasterizeCudaContext()


def _run_small_mesh_step(lanes, out=None):
    """one fused step of the case above -> (loss, d_raw_albedo, d_v_offsets) as numpy; also the entry point of the child process of the test below"""
    from c3d_hip.mesh_step import FusedMeshStep
    H, W, mesh, raw, off, views, targets, masks, f, ft, vt, glctx = _small_mesh_step_case()
    st = FusedMeshStep("cuda", lanes=lanes)
    d_ra, d_vo = torch.empty_like(raw), torch.empty_like(off)
    loss = st.run(views, mesh.v, off, f, vt, ft, raw, glctx, targets, masks, d_ra, d_vo, H, W, w_mse=0.7, w_ssim=0.3, scale=1 / 3)
    r = dict(loss=loss.detach().cpu().numpy().copy(), d_ra=d_ra.cpu().numpy(), d_vo=d_vo.cpu().numpy())
    if out:
        np.savez(out, **r)
    return r


def test_fused_mesh_step_is_bit_reproducible():
    """The fused view has no float atomics (the antialias blends and their gradients are gathers over a pixel's four pairs, texel gradients are added as
    64-bit integers): loss, texture gradient and vertex gradient have the SAME BITS on every run (and for every value of the `lanes` argument, which the
    round-4 library keeps for the ABI but no longer uses: the views of a step go through every stage together)."""
    runs = [_run_small_mesh_step(l) for l in (3, 3, 1, 2)]
    assert float(np.abs(runs[0]["d_ra"]).max()) > 0 and float(np.abs(runs[0]["d_vo"]).max()) > 0
    for r in runs[1:]:
        for k in ("loss", "d_ra", "d_vo"):
            assert np.array_equal(runs[0][k], r[k]), k


def test_fused_mesh_step_is_bit_reproducible_at_config5_size():
    """The same at BASELINE config 5's size (499 k triangles, 1024^2 texture, 1024 x 1024, eight views, MSE + MS-SSIM): two runs on four lanes and one on two lanes
    give the same bits for the loss, the 3 M texel gradients and the 250 k vertex gradients -- with ~4 M integer atomics per view landing in one pair of planes."""
    from c3d_hip.mesh_step import FusedMeshStep
    from shared_utils.camera_utils import OrbitCamera, orbit_camera
    import nvdiffrast.torch as dr
    H = W = 1024
    v, f, vt, _ = S.make_uv_sphere(500, 500, radius=0.7, displacement=0.05)
    g = torch.Generator(device="cpu").manual_seed(7)
    raw = torch.randn((1024, 1024, 3), generator=g).cuda()
    tv, tf, tvt = T(v), T(f, torch.int32), T(vt)
    off = (torch.randn(tuple(tv.shape), generator=g) * 1e-4).cuda()
    cam = OrbitCamera(W, H, fovy=49.1)
    proj = cam.perspective.astype(np.float32)
    views = [((proj @ np.linalg.inv(orbit_camera(e, az, 2.0).astype(np.float32)).astype(np.float32)).astype(np.float32), (1.0, 1.0, 1.0)) for e in (-20.0, 20.0) for az in (0.0, 90.0, 180.0, 270.0)]
    targets = [torch.rand((3, H, W), generator=g).cuda() for _ in views]
    glctx = dr.RasterizeCudaContext()
    outs = []
    for lanes in (4, 4, 2):
        st = FusedMeshStep("cuda", lanes=lanes)
        d_ra, d_vo = torch.empty_like(raw), torch.empty_like(off)
        loss = st.run(views, tv, off, tf, tvt, tf, raw, glctx, targets, None, d_ra, d_vo, H, W, w_mse=0.7, w_ssim=0.3, scale=1.0 / len(views)).clone()
        outs.append((loss, d_ra.clone(), d_vo.clone()))
    assert float(outs[0][1].abs().max()) > 0 and float(outs[0][2].abs().max()) > 0 and bool(torch.isfinite(outs[0][0]).all())
    for o in outs[1:]:
        assert torch.equal(o[0], outs[0][0]) and torch.equal(o[1], outs[0][1]) and torch.equal(o[2], outs[0][2])


def test_fused_mesh_step_accumulate_lanes_and_argument_checks():
    """c3d_mesh_train_views through FusedMeshStep: accumulate=True adds to the gradient buffers, the result does not depend on the number of view
    lanes, a rank without views leaves zero gradients, and bad arguments are refused with a message (NULL pointers, the MS-SSIM term on images that
    are too small for its five scales)."""
    from c3d_hip.mesh_step import FusedMeshStep
    from c3d_hip.mesh_sigs import MeshStepLoss, MeshView
    import nvdiffrast.torch as dr
    H, W, mesh, raw, off, views, targets, masks, f, ft, vt, glctx = _small_mesh_step_case()
    res = {}
    for lanes in (1, 3):
        st = FusedMeshStep("cuda", lanes=lanes)
        d_ra, d_vo = torch.empty_like(raw), torch.empty_like(off)
        loss = st.run(views, mesh.v, off, f, vt, ft, raw, glctx, targets, masks, d_ra, d_vo, H, W, w_mse=0.7, w_ssim=0.3, scale=1 / 3).clone()
        res[lanes] = (loss, d_ra.clone(), d_vo.clone())
        loss2 = st.run(views, mesh.v, off, f, vt, ft, raw, glctx, targets, masks, d_ra, d_vo, H, W, w_mse=0.7, w_ssim=0.3, scale=1 / 3, accumulate=True).clone()
        assert torch.allclose(loss2, loss, rtol=1e-5)
        assert rel_err(d_ra.cpu().numpy(), 2 * res[lanes][1].cpu().numpy()) <= 1e-5 and rel_err(d_vo.cpu().numpy(), 2 * res[lanes][2].cpu().numpy()) <= 1e-5
        # no views on this rank: zero gradients, zero loss
        z = st.run([], mesh.v, off, f, vt, ft, raw, glctx, [], None, d_ra, d_vo, H, W)
        assert float(z) == 0.0 and float(d_ra.abs().max()) == 0.0 and float(d_vo.abs().max()) == 0.0
    assert torch.allclose(res[1][0], res[3][0], rtol=1e-5) and float(res[1][1].abs().max()) > 0 and float(res[1][2].abs().max()) > 0
    assert rel_err(res[3][1].cpu().numpy(), res[1][1].cpu().numpy()) <= 1e-5 and rel_err(res[3][2].cpu().numpy(), res[1][2].cpu().numpy()) <= 1e-5
    # geometry not trained: no vertex gradient is produced, the texture gradient is the same
    st = FusedMeshStep("cuda", lanes=2)
    d_ra = torch.empty_like(raw)
    st.run(views, mesh.v, off, f, vt, ft, raw, glctx, targets, masks, d_ra, None, H, W, w_mse=0.7, w_ssim=0.3, scale=1 / 3)
    assert rel_err(d_ra.cpu().numpy(), res[1][1].cpu().numpy()) <= 1e-5
    # more views than one group holds (16): the groups run one after the other on the same state; equals two accumulated calls of 16 + 3 views
    many = [views[i % 3] for i in range(19)]
    tg19, mk19 = [targets[i % 3] for i in range(19)], [masks[i % 3] for i in range(19)]
    d_a, d_b, v_a, v_b = torch.empty_like(raw), torch.empty_like(raw), torch.empty_like(off), torch.empty_like(off)
    la = st.run(many, mesh.v, off, f, vt, ft, raw, glctx, tg19, mk19, d_a, v_a, H, W, w_mse=0.7, w_ssim=0.3, scale=1 / 19).clone()
    lb = st.run(many[:16], mesh.v, off, f, vt, ft, raw, glctx, tg19[:16], mk19[:16], d_b, v_b, H, W, w_mse=0.7, w_ssim=0.3, scale=1 / 19).clone()
    lb = lb + st.run(many[16:], mesh.v, off, f, vt, ft, raw, glctx, tg19[16:], mk19[16:], d_b, v_b, H, W, w_mse=0.7, w_ssim=0.3, scale=1 / 19, accumulate=True)
    assert torch.allclose(la, lb, rtol=1e-5) and rel_err(d_a.cpu().numpy(), d_b.cpu().numpy()) <= 1e-5 and rel_err(v_a.cpu().numpy(), v_b.cpu().numpy()) <= 1e-5
    assert abs(float(la) * 19 / 3 - float(res[1][0]) * (6 * 3 + 1) / 3) <= 0.2 * float(res[1][0]) * 19 / 3      # 6 x the three views + the first again: the scale of the loss is right
    # argument checks
    import c3d_hip
    lib = c3d_hip.lib()
    one = (MeshView * 1)(MeshView(int(mesh.v.shape[0]), int(f.shape[0]), int(vt.shape[0]), 128, 128, 64, 64, (C.c_float * 16)(*[float(x) for x in views[0][0].reshape(-1)]), (C.c_float * 3)(1, 1, 1)))
    assert lib.c3d_mesh_train_views(one, 1, *([None] * 8), None, None, None, None, None, None, 0, 1, None, None) != 0 and b"NULL" in lib.c3d_last_error()
    tg = (C.c_void_p * 1)(targets[0].data_ptr())
    ws = torch.empty((lib.c3d_mesh_step_workspace_bytes(one[0].V, one[0].T, 128, 128, 64, 64, 1, 1),), dtype=torch.uint8, device="cuda")
    bad = MeshStepLoss(1.0, 0.5, 1.0)
    p = c3d_hip.ptr
    rc = lib.c3d_mesh_train_views(one, 1, p(mesh.v), p(off), p(f), p(vt), p(ft), p(raw), p(dr._topology(f)), p(glctx.vertex_topology(f, one[0].V)), tg, None, C.byref(bad), p(d_ra),
                                  None, None, 0, 1, p(ws), None)
    assert rc != 0 and b"160" in lib.c3d_last_error()


def test_config1_example_workflow_through_the_nodes(tmp_path):
    """BASELINE config 1 (Render_Mesh_and_3DGS_Example): Load 3DGS -> GS Orbit Renderer and Mesh Orbit Renderer, 256x256, the
    MVDream(4) orbit, driven through the node classes; checked against the oracles."""
    import nodes as N
    from oracle import gs_oracle as O
    from mesh_processer.mesh_utils import construct_list_of_gs_attributes, write_gs_ply
    sc = S.make_ball_cloud(N=10000, seed=0)
    raw_scale = np.log(sc["scales"]); raw_op = np.log(sc["opacities"] / (1 - sc["opacities"]))
    f_dc = sc["shs"][:, :1].transpose(0, 2, 1).reshape(10000, -1); f_rest = sc["shs"][:, 1:].transpose(0, 2, 1).reshape(10000, -1)
    ply = write_gs_ply(sc["means3D"], np.zeros_like(sc["means3D"]), f_dc, f_rest, raw_op, raw_scale, sc["rotations"],
                       construct_list_of_gs_attributes(np.zeros((1, 1, 3)), np.zeros((1, 15, 3)), raw_scale, sc["rotations"]))
    path = str(tmp_path / "ball.ply")
    assert N.Save_3DGS().save_gs(ply, path)[0] == path
    (gs_ply,) = N.Load_3DGS().load_gs(path)
    poses = [[1.75, 0.0, az, 0.0, 0.0, 0.0] for az in (0.0, 90.0, 180.0, -90.0)]
    with torch.inference_mode():
        imgs, masks, depths = N.Gaussian_Splatting_Orbit_Renderer().render_gs(gs_ply, 256, 256, poses, 49.1, 1.0, 1.0, 1.0)
    assert imgs.shape == (4, 256, 256, 3) and masks.shape == (4, 256, 256) and depths.shape == (4, 256, 256, 3)
    for i, az in enumerate((0.0, 90.0, 180.0, -90.0)):
        st = S.camera_settings(256, 256, 49.1, 0.0, az, 1.75)
        oc, orad, od, oa, _ = O.forward(sc["means3D"], sc["opacities"], st, shs=sc["shs"], scales=sc["scales"], rotations=sc["rotations"], nthreads=8)
        assert np.abs(imgs[i].cpu().numpy() - np.clip(oc, 0, 1).transpose(1, 2, 0)).mean() <= IMG_L1
        assert np.abs(masks[i].cpu().numpy() - oa[0]).mean() <= IMG_L1
    # mesh half
    mesh = _torch_mesh(24, 48, tex=32)
    with torch.inference_mode():
        mi, mm, md, mn, mv = N.Mesh_Orbit_Renderer().render_mesh(mesh, 256, 256, poses, 49.1, 0.0, 0.0, 0.0, True, render_depth=True, render_normal=True)
    assert mi.shape == (4, 256, 256, 3) and mm.shape == (4, 256, 256) and md.shape == (4, 256, 256, 3) and mn.shape == (4, 256, 256, 3)
    assert 0.05 < mm.mean().item() < 0.9 and torch.isfinite(mi).all() and (mi[mm == 0] == 0).all()


@pytest.mark.parametrize("init", ["pointcloud", "mesh", "ply_wins_over_mesh"])
def test_gaussian_splatting_3d_node_accepts_every_initialiser(init):
    """[Comfy3D] Gaussian Splatting 3D with a point cloud / mesh / ply initialiser (reference nodes.py:1294-1301: point cloud, else ply, else
    mesh): a short training run through the node -- default loss (MS-SSIM 0.2, random backgrounds), i.e. the fused forward / backward halves."""
    import nodes as N
    from MVs_Algorithms.GaussianSplatting.main_3DGS_renderer import PointCloud, GaussianSplattingRenderer
    H = W = 176
    poses = [[1.75, 0.0, az, 0.0, 0.0, 0.0] for az in (0.0, 90.0, 180.0, -90.0)]
    rng = np.random.default_rng(2)
    yy, xx = np.meshgrid(np.linspace(-1, 1, H), np.linspace(-1, 1, W), indexing="ij")
    disk = ((xx ** 2 + yy ** 2) < 0.4).astype(np.float32)
    imgs = torch.tensor(np.stack([np.stack([disk * 0.8, disk * 0.3, disk * 0.5], -1)] * 4).astype(np.float32))
    masks = torch.tensor(np.stack([disk] * 4))
    kw = {}
    n_expect = None
    if init == "pointcloud":
        pts = rng.normal(size=(3000, 3)) * 0.25
        kw["points_cloud_to_initialize_gaussian"] = PointCloud(points=pts, colors=rng.uniform(size=(3000, 3)), normals=np.zeros((3000, 3)))
        n_expect = 3000
    elif init == "mesh":
        kw["mesh_to_initialize_gaussian"] = _torch_mesh(12, 24, tex=16)
    else:
        src = GaussianSplattingRenderer(sh_degree=3, device="cuda")
        src.initialize(None, num_pts=1234)
        kw["ply_to_initialize_gaussian"] = src.gaussians.to_ply()
        kw["mesh_to_initialize_gaussian"] = _torch_mesh(12, 24, tex=16)
        n_expect = 1234
    np.random.seed(0); torch.manual_seed(0)
    (ply,) = N.Gaussian_Splatting_3D().run_gs(imgs, masks, poses, 49.1, 6, 2, 0.2, 3, 0.0, 0.0, 0.5, 0.0025, 0.05, 0.005, 0.001, 0.00016, 0.0000016, 0.01, 30000,
                                              2000, 3, 0.01, 10 ** 6, 10 ** 6, 100, 3000, 0.0002, 3, **kw)
    n = ply.elements[0].count if hasattr(ply, "elements") else len(ply["vertex"])
    assert n > 0 and (n_expect is None or n == n_expect)
    (gs_ply,) = (ply,)
    with torch.inference_mode():
        out, m, d = N.Gaussian_Splatting_Orbit_Renderer().render_gs(gs_ply, W, H, poses[:1], 49.1, 1.0, 1.0, 1.0)
    assert torch.isfinite(out).all() and out.shape == (1, H, W, 3)


def test_renderer_initialises_from_a_uv_grid_triple():
    """GaussianSplattingRenderer.initialize with the UV-grid triple (reference main_3DGS_renderer.py:526-539, :827-828: the `else` branch of initialize): one Gaussian
    per texel position, random near-black colours, the k-nn scale, learning-rate scale 10 -- the same model create_from_pcd builds from those points"""
    from MVs_Algorithms.GaussianSplatting.main_3DGS_renderer import GaussianSplattingRenderer, PointCloud, SH2RGB
    rng = np.random.default_rng(0)
    pts = rng.uniform(-0.5, 0.5, size=(700, 3))
    uv = (rng.uniform(size=(700, 2)), [p for p in pts], rng.normal(size=(700, 3)))
    np.random.seed(3)
    a = GaussianSplattingRenderer(sh_degree=3, device="cuda")
    a.initialize(uv)
    np.random.seed(3)
    b = GaussianSplattingRenderer(sh_degree=3, device="cuda")
    b.gaussians.create_from_pcd(PointCloud(points=pts, colors=SH2RGB(np.random.random((700, 3)) / 255.0), normals=np.zeros((700, 3))), 10)
    ga, gb = a.gaussians, b.gaussians
    assert ga._xyz.shape == (700, 3) and ga.spatial_lr_scale == 10
    for x, y in ((ga._xyz, gb._xyz), (ga._features_dc, gb._features_dc), (ga._scaling, gb._scaling), (ga._opacity, gb._opacity), (ga._rotation, gb._rotation)):
        assert torch.equal(x, y)
    with pytest.raises(TypeError):
        a.initialize((1, 2))


@pytest.mark.parametrize("sc", [(130, 97, 24, 40), (256, 256, 64, 128), (96, 96, 2, 3)])
def test_rasterize_backward_gather_equals_scatter(sc):
    """The atomic-free rasterize backward (per-triangle corner records + fixed-order per-vertex sum; large triangles by a workgroup) against
    the scatter-add formulation and the oracle; two runs give identical bits."""
    import nvdiffrast.torch as dr
    H, W, nl, no = sc
    pos, f, vt, vn = _scene(H, W, nl, no)          # the last scene has a handful of huge triangles: the workgroup-per-triangle path
    ctx = dr.RasterizeCudaContext()
    rng = np.random.default_rng(5)
    gy = rng.normal(size=(1, H, W, 4)).astype(np.float32)
    grads = {}
    for mode in (True, False, True):
        dr.ATOMIC_FREE_BACKWARD = mode
        try:
            p = T(pos, grad=True)
            rast, _ = dr.rasterize(ctx, p, T(f, torch.int32), (H, W))
            (rast * T(gy)).sum().backward()
        finally:
            dr.ATOMIC_FREE_BACKWARD = True
        grads.setdefault(mode, []).append(p.grad.clone())
    a, a2 = grads[True]
    assert torch.equal(a, a2)                                            # bit-reproducible
    assert rel_err(a.cpu().numpy(), grads[False][0].cpu().numpy()) <= 1e-5   # same sums, different order
    orast, _ = M.rasterize(pos, f, (H, W))
    od = M.rasterize_bwd(pos, f, rast.detach().cpu().numpy(), gy)      # the oracle on the SAME winners (depth near-ties may differ)
    assert rel_err(a.cpu().numpy(), np.asarray(od).reshape(a.shape)) <= GRAD_REL


def test_renderer_fused_glue_equals_torch_chain():
    """DiffRastRenderer with the fused vertex transform / shade kernels (c3d_mesh_transform_*, c3d_mesh_shade_*) against the reference's torch
    op chain around the same nvdiffrast calls: images, alpha, lazily produced depth / normal, and the gradients of texture and geometry."""
    from MVs_Algorithms.DiffRastMesh.diff_mesh_renderer import DiffRastRenderer
    from shared_utils.camera_utils import OrbitCamera, orbit_camera
    H = W = 160
    cam = OrbitCamera(W, H, fovy=49.1)
    pose = orbit_camera(-15.0, 40.0, 2.0)
    bg = torch.tensor([0.2, 0.6, 0.9], device="cuda")
    rng = np.random.default_rng(9)
    gi = torch.tensor(rng.normal(size=(H, W, 3)).astype(np.float32), device="cuda")
    ga = torch.tensor(rng.normal(size=(H, W, 1)).astype(np.float32), device="cuda")
    outs = []
    # three ways: the whole view as one library call each way (c3d_mesh_view_*, the default), op by op with the fused glue kernels, the torch chain
    for fused_view, fused in ((True, True), (False, True), (False, False)):
        r = DiffRastRenderer(_torch_mesh(), force_cuda_rast=True).cuda()
        r.train_geo, r.fused_glue, r.fused_view = True, fused, fused_view
        with torch.no_grad():
            r.raw_albedo.mul_(3.0)                     # push part of the composite outside [0,1]: the clamps must agree too
            r.v_offsets.add_(torch.tensor(rng.normal(size=tuple(r.v_offsets.shape)).astype(np.float32), device="cuda") * 0.004 if fused_view else 0.0)
            if not fused_view:
                r.v_offsets.copy_(outs[0][6])          # the same offsets in all three
        out = r.render(pose, cam.perspective, H, W, bg_color=bg)
        ((out["image"] * gi).sum() + (out["alpha"] * ga).sum()).backward()
        outs.append((out["image"].detach(), out["alpha"].detach(), out["depth"].detach(), out["normal"].detach(), r.raw_albedo.grad.clone(), r.v_offsets.grad.clone(),
                     r.v_offsets.detach().clone()))
    b = outs[2]
    for a in outs[:2]:
        assert a[0].shape == (H, W, 3) and a[1].shape == (H, W, 1)
        # one combined 4x4 instead of two successive ones: clip coordinates differ in the last bits, the silhouette blend amplifies that by 1/pixel
        for x, y, name in zip(a[:4], b[:4], ("image", "alpha", "depth", "normal")):
            assert (x - y).abs().mean().item() <= 1e-5 and (x - y).abs().max().item() <= 2e-3, name
        assert rel_err(a[4].cpu().numpy(), b[4].cpu().numpy()) <= GRAD_REL
        assert rel_err(a[5].cpu().numpy(), b[5].cpu().numpy()) <= 5e-3
    # texture-only training (no geometry gradient, no vertex topology) and a scalar background through the fused view
    r = DiffRastRenderer(_torch_mesh(), force_cuda_rast=True).cuda()
    r2 = DiffRastRenderer(_torch_mesh(), force_cuda_rast=True).cuda()
    r2.fused_view = False
    o1, o2 = r.render(pose, cam.perspective, H, W, bg_color=1), r2.render(pose, cam.perspective, H, W, bg_color=1)
    (o1["image"] * gi).sum().backward(); (o2["image"] * gi).sum().backward()
    assert (o1["image"] - o2["image"]).abs().max().item() <= 2e-3 and rel_err(r.raw_albedo.grad.cpu().numpy(), r2.raw_albedo.grad.cpu().numpy()) <= GRAD_REL
    assert r.v_offsets.grad is None or r.v_offsets.grad.abs().max().item() == 0
    # super-sampling (reference diff_mesh_renderer.py:14-36,142-149): the fused view at the super-sampled size + the reference's bilinear down-scaling,
    # against the torch chain; ssaa = 1.5 makes the super-sampled size (rounded up to a multiple of 8) a non-integer multiple of the output size
    h0, w0 = 72, 96
    cam2 = OrbitCamera(w0, h0, fovy=49.1)
    g2 = torch.tensor(rng.normal(size=(h0, w0, 3)).astype(np.float32), device="cuda")
    for ssaa in (2, 1.5):
        res = []
        for fv in (True, False):
            r = DiffRastRenderer(_torch_mesh(), force_cuda_rast=True).cuda()
            r.train_geo, r.fused_view, r.fused_glue = True, fv, fv
            o = r.render(pose, cam2.perspective, h0, w0, ssaa=ssaa, bg_color=bg)
            ((o["image"] * g2).sum() + o["alpha"].sum()).backward()
            res.append((o["image"].detach(), o["alpha"].detach(), o["depth"].detach(), o["normal"].detach(), r.raw_albedo.grad.clone(), r.v_offsets.grad.clone()))
        a, b = res
        assert a[0].shape == (h0, w0, 3) and a[1].shape == (h0, w0, 1) and a[2].shape == b[2].shape and a[3].shape == b[3].shape
        for x, y, name in zip(a[:4], b[:4], ("image", "alpha", "depth", "normal")):
            assert (x - y).abs().mean().item() <= 1e-5 and (x - y).abs().max().item() <= 2e-3, (ssaa, name)
        assert rel_err(a[4].cpu().numpy(), b[4].cpu().numpy()) <= GRAD_REL and rel_err(a[5].cpu().numpy(), b[5].cpu().numpy()) <= 5e-3, ssaa


def test_other_in_tree_consumers_op_sequences():
    """The two other in-tree users of the same ops, restated call for call against the oracle:
    FlexiCubes' render_mesh (flexicubes_renderer.py:41-74: mask antialias, per-face normals through an (i,i,i) index buffer, vertex normals +
    antialias, two views in one batch) and the UV bake of color_func_to_albedo (mesh_utils.py:533-541: rasterize in UV space with `ft`,
    interpolate positions with `f`)."""
    import nvdiffrast.torch as dr
    H, W = 120, 136
    v, f, vt, vn = S.make_uv_sphere(20, 32, radius=0.7, displacement=0.1)
    pos = np.concatenate([S.mesh_clip_positions(v, el, az, 2.0, W, H)[0] for el, az in ((-10.0, 20.0), (25.0, 200.0))], 0)     # [2,V,4]
    fi = f.astype(np.int32)
    fnrm = np.cross(v[fi[:, 1]] - v[fi[:, 0]], v[fi[:, 2]] - v[fi[:, 0]]); fnrm /= np.linalg.norm(fnrm, axis=1, keepdims=True) + 1e-20
    nidx = np.repeat(np.arange(fi.shape[0], dtype=np.int32)[:, None], 3, 1)
    ctx = dr.RasterizeCudaContext()
    tp, tf = T(pos), T(fi, torch.int32)
    rast, db = dr.rasterize(ctx, tp, tf.int(), [H, W])
    alpha = (rast[..., -1:] > 0).float()
    mask = dr.antialias(alpha, rast, tp, tf)
    fn_img, _ = dr.interpolate(T(fnrm.astype(np.float32)).unsqueeze(0).contiguous(), rast, T(nidx, torch.int32))
    vn_img, _ = dr.interpolate(T(vn).unsqueeze(0).contiguous(), rast, tf)
    vn_aa = dr.antialias((vn_img + 1) * 0.5, rast, tp, tf)
    orast, _ = M.rasterize(pos, fi, (H, W))
    r = rast.cpu().numpy()
    same = r[..., 3] == orast[..., 3]
    assert (~same).mean() <= 2e-4
    use = rast.cpu().numpy()                                   # feed the oracle the same winners
    oalpha = (use[..., 3:] > 0).astype(np.float32)
    assert np.abs(mask.cpu().numpy() - M.antialias(oalpha, use, pos, fi)).mean() <= IMG_L1
    ofn, _ = M.interpolate(fnrm.astype(np.float32)[None], use, nidx)
    assert np.abs(fn_img.cpu().numpy() - ofn).max() <= 1e-5      # all three corners carry the same value: exactly the face normal
    covered = use[..., 3] > 0
    assert np.allclose(np.linalg.norm(fn_img.cpu().numpy()[covered], axis=-1), 1.0, atol=1e-5)
    ovn, _ = M.interpolate(vn[None], use, fi)
    assert np.abs(vn_aa.cpu().numpy() - M.antialias((ovn + 1) * 0.5, use, pos, fi)).mean() <= IMG_L1
    # UV bake: positions as a texture
    res = 96
    uv = vt * 2.0 - 1.0
    uv4 = np.concatenate([uv, np.zeros_like(uv[:, :1]), np.ones_like(uv[:, :1])], -1).astype(np.float32)[None]
    rast_uv, _ = dr.rasterize(ctx, T(uv4), tf, (res, res))
    xyzs, _ = dr.interpolate(T(v).unsqueeze(0), rast_uv, tf)
    ones, _ = dr.interpolate(torch.ones_like(T(v)[:, :1]).unsqueeze(0), rast_uv, tf)
    use_uv = rast_uv.cpu().numpy()
    orast_uv, _ = M.rasterize(uv4, fi, (res, res))
    assert (use_uv[..., 3] != orast_uv[..., 3]).mean() <= 5e-4
    oxyz, _ = M.interpolate(v[None], use_uv, fi)
    assert np.abs(xyzs.cpu().numpy() - oxyz).max() <= 1e-5
    m = ones.cpu().numpy()[..., 0] > 0
    assert m.mean() > 0.8 and np.array_equal(m, use_uv[..., 3] > 0)      # the lat-long chart fills most of the texture
    assert np.abs(np.linalg.norm(xyzs.cpu().numpy()[0][m[0]], axis=-1) - 0.7).max() <= 0.1 + 1e-3     # baked positions lie on the displaced sphere


def test_antialias_leaves_pairs_of_a_near_plane_clipped_triangle_alone():
    """VERDICT r4, next-round 4 ii: what the HIP antialias does with a pixel pair whose nearer triangle is near-plane clipped, pinned: the pair is skipped (colour and
    gradient pass through), as the oracle does (tests/test_mesh_oracle.py holds the rule and its control case); the stand-alone op and the fused view agree."""
    import nvdiffrast.torch as dr
    from test_mesh_oracle import _clipped_edge_scene
    ctx = dr.RasterizeCudaContext()
    for clipped in (False, True):
        pos64, tri, (H, W) = _clipped_edge_scene(clipped, W=32, H=24, xe_px=17.25)
        tp = T(pos64.astype(np.float32), grad=True)
        tt = T(tri, torch.int32)
        rast, _ = dr.rasterize(ctx, tp, tt, (H, W))
        orast, _ = M.rasterize(pos64.astype(np.float32), tri, (H, W))
        assert (rast.detach().cpu().numpy()[..., 3] == orast[..., 3]).all()
        col = (rast[..., 3:] > 0).float().detach().requires_grad_(True)
        out = dr.antialias(col, rast.detach(), tp, tt)
        oout = M.antialias(col.detach().cpu().numpy(), orast, pos64.astype(np.float32), tri)
        assert np.abs(out.detach().cpu().numpy() - oout).max() <= 1e-6
        out.sum().backward()
        if clipped:
            assert torch.equal(out.detach(), col.detach())
            assert bool((col.grad == 1).all()) and bool((tp.grad == 0).all())
        else:
            rows = slice(10, 14)
            assert np.allclose(out.detach().cpu().numpy()[0, rows, 17, 0], 0.25, atol=1e-5) and float(tp.grad.abs().max()) > 0


def test_near_plane_clipping_matches_oracle_forward_and_backward():
    """Triangles with vertices at / behind the camera plane (w <= 0) are clipped against the near plane, not dropped:
    HIP against the oracle (which tests/test_mesh_oracle.py holds to hand-clipped geometry) -- ids, barycentrics, depth; the gradient of the
    atomic-free rasterize backward against the scatter formulation and the oracle; the whole pipeline stays finite."""
    import nvdiffrast.torch as dr
    from test_mesh_oracle import _near_plane_scene
    pos64, tri, (H, W) = _near_plane_scene()
    pos = pos64.astype(np.float32)
    ctx = dr.RasterizeCudaContext()
    tp = T(pos, grad=True)
    rast, db = dr.rasterize(ctx, tp, T(tri, torch.int32), (H, W))
    orast, odb = M.rasterize(pos, tri, (H, W))
    r = rast.detach().cpu().numpy()
    assert (orast[..., 3] > 0).mean() > 0.3
    differ = r[..., 3] != orast[..., 3]
    assert differ.mean() <= 2e-3, differ.mean()                       # float coverage test on the clipped triangles: edge pixels may fall either way
    m = ~differ
    assert np.abs(r[m][:, :3] - orast[m][:, :3]).max() <= 2e-4
    rng = np.random.default_rng(3)
    gy = rng.normal(size=(1, H, W, 4)).astype(np.float32)
    (rast * T(gy)).sum().backward()
    g_gather = tp.grad.clone()
    dr.ATOMIC_FREE_BACKWARD = False
    try:
        tp2 = T(pos, grad=True)
        rast2, _ = dr.rasterize(ctx, tp2, T(tri, torch.int32), (H, W))
        (rast2 * T(gy)).sum().backward()
    finally:
        dr.ATOMIC_FREE_BACKWARD = True
    assert torch.equal(rast, rast2)
    assert rel_err(g_gather.cpu().numpy(), tp2.grad.cpu().numpy()) <= 1e-4
    r64, _ = M.rasterize(pos, tri, (H, W), dtype=np.float64)
    dpos = M.rasterize_bwd(pos, tri, r64, gy, dtype=np.float64)
    gm = gy.copy(); gm[differ] = 0                                    # pixels owned differently carry different gradients by construction
    assert rel_err(g_gather.cpu().numpy(), dpos) <= 5e-2              # loose: a handful of differing edge pixels on two huge triangles
    # downstream ops on a clipped rast
    attr = T(rng.normal(size=(1, pos.shape[1], 3)).astype(np.float32))
    out, _ = dr.interpolate(attr, rast.detach(), T(tri, torch.int32))
    oout, _ = M.interpolate(attr.cpu().numpy(), orast, tri)
    assert np.abs(out.cpu().numpy()[m] - oout[m]).max() <= 2e-3 and torch.isfinite(out).all()
    aa = dr.antialias(out, rast.detach(), tp.detach(), T(tri, torch.int32))
    assert torch.isfinite(aa).all()
