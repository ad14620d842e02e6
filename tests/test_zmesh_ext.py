"""The wider `nvdiffrast.torch` surface other consumers in the reference use: mip-mapped texture modes
('linear-mipmap-linear', 'linear-mipmap-nearest', texture_construct_mip), DepthPeeler and range mode -- HIP through the C-ABI against
oracle/mesh_oracle.c (-m gpu), plus the host-side checks that run without a device.
Reference call sites that reach these modes via filter_mode='auto' + uv_da: Gen_3D_Modules/LGM/nerf_marching_cubes_converter.py:232,
Gen_3D_Modules/TRELLIS/trellis/utils/postprocessing_utils.py:384, Gen_3D_Modules/Stable3DGen/trellis/utils/_rasterization.py:88."""
import numpy as np
import pytest
import torch

from c3d_hip import synthetic as S
from oracle import mesh_oracle as M
from helpers import rel_err

IMG_L1 = 1e-4
GRAD_REL = 1e-3


def T(a, dtype=torch.float32, grad=False):
    return torch.tensor(np.asarray(a), dtype=dtype, device="cuda", requires_grad=grad)


def test_level_table_matches_the_oracle_and_cpu_tensors_are_refused():
    import nvdiffrast.torch as dr
    M.build()
    for (h, w, ml) in [(8, 4, None), (64, 64, None), (64, 64, 3), (6, 6, 1), (1, 1, None), (2, 32, None), (1024, 2048, None)]:
        assert dr._mip_levels(h, w, ml) == M.mip_info(h, w, ml)
    with pytest.raises(ValueError):
        dr._mip_levels(6, 6, None)
    with pytest.raises(ValueError):
        dr.texture(torch.zeros(1, 4, 4, 3), torch.zeros(1, 2, 2, 2), filter_mode="linear-mipmap-linear")     # no uv_da / bias
    with pytest.raises(RuntimeError, match="HIP device"):
        dr.texture(torch.zeros(1, 4, 4, 3), torch.zeros(1, 2, 2, 2), uv_da=torch.zeros(1, 2, 2, 4))          # 'auto' -> mip mode; no CPU path
    with pytest.raises(RuntimeError, match="HIP device"):
        dr.texture_construct_mip(torch.zeros(1, 4, 4, 3))
    ctx = dr.RasterizeCudaContext(device="cuda:0")
    tri = torch.zeros((2, 3), dtype=torch.int32)
    with pytest.raises(ValueError):
        dr.rasterize(ctx, torch.zeros(1, 4, 4), tri, (8, 8), ranges=torch.tensor([[0, 1]], dtype=torch.int32))     # range mode: pos is [V,4]
    with pytest.raises(ValueError):
        dr.rasterize(ctx, torch.zeros(4, 4), tri, (8, 8), ranges=torch.tensor([[1, 2]], dtype=torch.int32))        # range past the end
    with pytest.raises(ValueError):
        dr.rasterize(ctx, torch.zeros(4, 4), tri, (8, 8), ranges=torch.tensor([0, 1], dtype=torch.int32))          # not [B,2]
    with pytest.raises(RuntimeError, match="HIP device"):
        dr.rasterize(ctx, torch.zeros(4, 4), tri, (8, 8), ranges=torch.tensor([[0, 2]], dtype=torch.int32))


@pytest.fixture()
def gpu():
    if not torch.cuda.is_available():
        pytest.fail("-m gpu tests need a HIP device (no CPU fallback exists)")
    M.build()


def _inputs(rng, Bt, B, Ht, Wt, C, H, W, spread=0.08):
    tex = rng.normal(size=(Bt, Ht, Wt, C)).astype(np.float32)
    uv = rng.uniform(-0.6, 1.6, size=(B, H, W, 2)).astype(np.float32)
    da = (rng.normal(size=(B, H, W, 4)) * spread).astype(np.float32)
    bias = rng.uniform(-0.5, 0.5, size=(B, H, W)).astype(np.float32)
    return tex, uv, da, bias


@pytest.mark.gpu
def test_pyramid_matches_oracle_and_its_backward_is_the_transpose(gpu):
    import nvdiffrast.torch as dr
    rng = np.random.default_rng(0)
    for shape, ml in [((2, 64, 32, 3), None), ((1, 16, 16, 4), 2), ((1, 2, 64, 1), None), ((1, 1, 1, 3), None), ((1, 12, 20, 2), 2)]:
        tex = rng.normal(size=shape).astype(np.float32)
        tt = T(tex, grad=True)
        mip = dr.texture_construct_mip(tt, max_mip_level=ml)
        ref = M.mip_build(tex, ml)
        assert tuple(mip.stack.shape) == ref.shape
        if ref.size == 0:
            continue
        assert np.abs(mip.stack.detach().cpu().numpy() - ref).max() <= 1e-6
        g = rng.normal(size=ref.shape).astype(np.float32)
        (mip.stack * T(g)).sum().backward()
        assert rel_err(tt.grad.cpu().numpy(), M.mip_build_bwd(g, shape, ml, dtype=np.float64)) <= 1e-5


@pytest.mark.gpu
@pytest.mark.parametrize("C,Bt", [(3, 1), (4, 2), (2, 2), (1, 1)])
@pytest.mark.parametrize("boundary", ["wrap", "clamp", "zero"])
def test_trilinear_forward_and_gradients(gpu, C, Bt, boundary):
    """C = 1, 3, 4 take the LDS-combined backward, C = 2 the direct-atomics one."""
    import nvdiffrast.torch as dr
    rng = np.random.default_rng(10 + C)
    tex, uv, da, bias = _inputs(rng, Bt, 2, 32, 64, C, 37, 45)
    tt, tu = T(tex, grad=True), T(uv, grad=True)
    out = dr.texture(tt, tu, uv_da=T(da), mip_level_bias=T(bias), boundary_mode=boundary)           # 'auto' -> linear-mipmap-linear
    ref = M.texture_mip(tex, uv, da, bias, boundary_mode=boundary)
    assert np.abs(out.detach().cpu().numpy() - ref).mean() <= 1e-6
    assert np.abs(out.detach().cpu().numpy() - ref).max() <= 1e-4
    g = rng.normal(size=ref.shape).astype(np.float32)
    (out * T(g)).sum().backward()
    dtex, dstack, duv = M.texture_mip_bwd(tex, uv, g, da, bias, boundary_mode=boundary, dtype=np.float64)
    total = dtex + M.mip_build_bwd(dstack, tex.shape, dtype=np.float64)
    assert rel_err(tt.grad.cpu().numpy(), total) <= GRAD_REL
    assert rel_err(tu.grad.cpu().numpy(), duv) <= GRAD_REL
    # uv_da alone, bias alone
    o2 = dr.texture(T(tex), T(uv), uv_da=T(da), boundary_mode=boundary)
    assert np.abs(o2.cpu().numpy() - M.texture_mip(tex, uv, da, None, boundary_mode=boundary)).max() <= 1e-4
    o3 = dr.texture(T(tex), T(uv), mip_level_bias=T(bias + 1.5), boundary_mode=boundary)
    assert np.abs(o3.cpu().numpy() - M.texture_mip(tex, uv, None, bias + 1.5, boundary_mode=boundary)).max() <= 1e-4


@pytest.mark.gpu
def test_gradients_wrt_uv_da_mip_level_bias_and_through_out_da(gpu):
    """'linear-mipmap-linear' propagates to uv_da and mip_level_bias (d out / d level = sample(level + 1) - sample(level) where the level is
    not clamped), and interpolate() passes what arrives at its pixel differentials on to the attributes and to rast_db -- the chain the reference's
    texture bakers build (Gen_3D_Modules/Hunyuan3D_V2/hy3dgen/texgen/differentiable_renderer/mesh_render.py:363: texture(tex, uv, uv_da, 'linear-mipmap-linear')
    behind interpolate(uv, rast, tri, rast_db, diff_attrs='all')).  HIP through nvdiffrast.torch against the float64 oracle."""
    import nvdiffrast.torch as dr
    rng = np.random.default_rng(31)
    tex, uv, da, bias = _inputs(rng, 1, 2, 32, 64, 3, 37, 45)
    tt, tu, tda, tb = T(tex, grad=True), T(uv, grad=True), T(da, grad=True), T(bias, grad=True)
    out = dr.texture(tt, tu, uv_da=tda, mip_level_bias=tb)
    g = rng.normal(size=out.shape).astype(np.float32)
    (out * T(g)).sum().backward()
    _, _, duv, dda, dbias = M.texture_mip_bwd(tex, uv, g, da, bias, dtype=np.float64, level_grads=True)
    assert np.abs(dda).sum() > 0 and np.abs(dbias).sum() > 0
    # a pixel whose float32 level sits within rounding of an integer takes the other pair of levels: allow a handful of such pixels
    for got, ref in ((tda.grad.cpu().numpy(), dda), (tb.grad.cpu().numpy(), dbias)):
        err = np.abs(got - ref)
        tol = GRAD_REL * np.abs(ref).max()
        assert (err > tol).sum() <= 4, int((err > tol).sum())
    assert rel_err(tu.grad.cpu().numpy(), duv) <= GRAD_REL
    # nearest-level mode: no level gradient
    tda2, tb2 = T(da, grad=True), T(bias, grad=True)
    (dr.texture(T(tex), T(uv), uv_da=tda2, mip_level_bias=tb2, filter_mode="linear-mipmap-nearest") * T(g)).sum().backward()
    assert float(tda2.grad.abs().max()) == 0.0 and float(tb2.grad.abs().max()) == 0.0
    # interpolate: out_da -> attributes and rast_db, on a rasterized mesh
    v, f, vt, _ = S.make_uv_sphere(12, 18)
    pos, _, _ = S.mesh_clip_positions(v, 20.0, 40.0, 2.2, 96, 80)
    ctx = dr.RasterizeCudaContext(device="cuda")
    rast, rast_db = dr.rasterize(ctx, T(pos), T(f, dtype=torch.int32), (80, 96))
    attr = np.concatenate([vt, rng.normal(size=(vt.shape[0], 1)).astype(np.float32)], 1)            # 3 attributes, differentials of the first two
    ta = T(attr, grad=True)
    db = rast_db.detach().clone().requires_grad_(True)
    o, o_da = dr.interpolate(ta, rast, T(f, dtype=torch.int32), rast_db=db, diff_attrs=[0, 1])
    g1, g2 = rng.normal(size=tuple(o.shape)).astype(np.float32), rng.normal(size=tuple(o_da.shape)).astype(np.float32)
    ((o * T(g1)).sum() + (o_da * T(g2)).sum()).backward()
    r_np, db_np = rast.detach().cpu().numpy(), rast_db.detach().cpu().numpy()
    d1, _ = M.interpolate_bwd(attr, r_np, f, g1, dtype=np.float64)
    d2, ddb = M.interpolate_da_bwd(attr, r_np, f, db_np, [0, 1], g2, dtype=np.float64)
    assert np.abs(d2).sum() > 0 and np.abs(ddb).sum() > 0
    assert rel_err(ta.grad.cpu().numpy(), d1 + d2) <= GRAD_REL
    assert rel_err(db.grad.cpu().numpy(), ddb) <= GRAD_REL


@pytest.mark.gpu
def test_nearest_level_mode_prebuilt_and_custom_stacks(gpu):
    import nvdiffrast.torch as dr
    rng = np.random.default_rng(20)
    tex, uv, da, _ = _inputs(rng, 1, 1, 16, 16, 3, 33, 31)
    lv = rng.integers(0, 5, size=(1, 33, 31)).astype(np.float32) + rng.uniform(-0.3, 0.3, size=(1, 33, 31)).astype(np.float32)   # never near x.5
    out = dr.texture(T(tex), T(uv), mip_level_bias=T(lv), filter_mode="linear-mipmap-nearest")
    assert np.abs(out.cpu().numpy() - M.texture_mip(tex, uv, None, lv, filter_mode="linear-mipmap-nearest")).max() <= 2e-5
    # max_mip_level limits the pyramid; a pre-built stack gives the same result as the internal one and carries the gradient to tex
    tt = T(tex, grad=True)
    mip = dr.texture_construct_mip(tt, max_mip_level=2)
    a = dr.texture(tt, T(uv), uv_da=T(da * 4), mip=mip, max_mip_level=2)
    b = dr.texture(T(tex), T(uv), uv_da=T(da * 4), max_mip_level=2)
    assert torch.equal(a.detach(), b)
    assert np.abs(b.cpu().numpy() - M.texture_mip(tex, uv, da * 4, None, max_mip_level=2)).max() <= 1e-4
    g = rng.normal(size=(1, 33, 31, 3)).astype(np.float32)
    (a * T(g)).sum().backward()
    dtex, dstack, _ = M.texture_mip_bwd(tex, uv, g, da * 4, None, max_mip_level=2, dtype=np.float64)
    assert rel_err(tt.grad.cpu().numpy(), dtex + M.mip_build_bwd(dstack, tex.shape, 2, dtype=np.float64)) <= GRAD_REL
    with pytest.raises(ValueError):
        dr.texture(T(np.zeros((1, 8, 8, 3))), T(uv), uv_da=T(da), mip=mip)                            # stack of another texture
    # custom stack: a list of level tensors that receive their own gradients; tex only gets its level-0 taps
    levels, _ = M.mip_info(16, 16, 2)
    custom = [rng.normal(size=(1, h, w, 3)).astype(np.float32) for (h, w) in levels[1:]]
    tc = [T(c, grad=True) for c in custom]
    tt2 = T(tex, grad=True)
    o = dr.texture(tt2, T(uv), uv_da=T(da * 4), mip=tc)
    stack = np.concatenate([c.reshape(1, -1, 3) for c in custom], axis=1)
    assert np.abs(o.detach().cpu().numpy() - M.texture_mip(tex, uv, da * 4, None, stack=stack, max_mip_level=2)).max() <= 1e-4
    (o * T(g)).sum().backward()
    dtex, dstack, _ = M.texture_mip_bwd(tex, uv, g, da * 4, None, stack=stack, max_mip_level=2, dtype=np.float64)
    assert rel_err(tt2.grad.cpu().numpy(), dtex) <= GRAD_REL
    assert rel_err(tc[0].grad.cpu().numpy().reshape(-1), dstack[:, :64].reshape(-1)) <= GRAD_REL
    assert rel_err(tc[1].grad.cpu().numpy().reshape(-1), dstack[:, 64:].reshape(-1)) <= GRAD_REL
    with pytest.raises(ValueError):
        dr.texture(T(np.zeros((1, 6, 6, 3))), T(uv), uv_da=T(da))                                     # 3x3 cannot be halved
    assert dr.texture(T(np.ones((1, 6, 6, 3))), T(uv), uv_da=T(da), max_mip_level=1).shape == (1, 33, 31, 3)


@pytest.mark.gpu
def test_textured_mesh_with_auto_mipmapping(gpu):
    """rasterize -> interpolate(uv, diff 'all') -> texture(tex, uv, uv_da): the sequence of LGM's mesh refiner
    (nerf_marching_cubes_converter.py:215-232) and TRELLIS's texture baker (postprocessing_utils.py:376-384)."""
    import nvdiffrast.torch as dr
    H, W = 96, 128
    v, f, vt, vn = S.make_uv_sphere(24, 40, radius=0.7, displacement=0.15)
    pos, _, _ = S.mesh_clip_positions(v, -20.0, 35.0, 2.0, W, H)
    rng = np.random.default_rng(2)
    tex = rng.normal(size=(1, 256, 256, 3)).astype(np.float32)                                          # minified: several texels per pixel
    ctx = dr.RasterizeCudaContext()
    ttex, tvt, ttri, tpos = T(tex, grad=True), T(vt[None], grad=True), T(f, torch.int32), T(pos, grad=True)
    rast, db = dr.rasterize(ctx, tpos, ttri, (H, W))
    texc, texd = dr.interpolate(tvt, rast, ttri, rast_db=db, diff_attrs='all')
    col = dr.texture(ttex, texc, texd)                                                                   # uv_da positional, filter 'auto'
    orast, odb = M.rasterize(pos, f, (H, W))
    otexc, otexd = M.interpolate(vt[None], orast, f, odb, "all")
    ocol = M.texture_mip(tex, otexc, otexd)
    assert (rast.detach().cpu().numpy()[..., 3] != orast[..., 3]).sum() <= 2
    assert np.abs(col.detach().cpu().numpy() - ocol).mean() <= IMG_L1
    lin = M.texture(tex, otexc)
    assert np.abs(ocol - lin).mean() > 0.05                                                              # the pyramid is really in play
    g = rng.normal(size=ocol.shape).astype(np.float32)
    (col * T(g)).sum().backward()
    d = np.float64
    r64, db64 = M.rasterize(pos, f, (H, W), dtype=d)
    texc64, texd64 = M.interpolate(vt[None], r64, f, db64, "all", dtype=d)
    dtex, dstack, duv, dda, _ = M.texture_mip_bwd(tex, texc64, g, texd64, None, dtype=d, level_grads=True)
    dvt, drast = M.interpolate_bwd(vt[None], r64, f, duv, dtype=d)
    # the texture coordinates also steer the mip LEVEL through their pixel differentials: texture() hands d/d(uv_da) back, interpolate() passes it on to vt
    dvt_da, ddb = M.interpolate_da_bwd(vt[None], r64, f, db64, "all", dda, dtype=d)
    assert np.abs(dvt_da).max() > 1e-3 * np.abs(dvt).max()
    assert rel_err(ttex.grad.cpu().numpy(), dtex + M.mip_build_bwd(dstack, tex.shape, dtype=d)) <= GRAD_REL
    assert rel_err(tvt.grad.cpu().numpy(), dvt + dvt_da) <= 3 * GRAD_REL      # a white-noise texture: d out / d level is as large as the signal, float32 levels differ in the last bits
    # ... and rasterize(grad_db=True) passes what arrives at rast_db on to the positions (round 3), next to the (u, v) path
    dpos_uv = M.rasterize_bwd(pos, f, r64, drast, dtype=d)
    dpos = M.rasterize_bwd(pos, f, r64, drast, ddb=ddb, dtype=d)
    assert np.abs(dpos - dpos_uv).max() > 1e-3 * np.abs(dpos_uv).max()
    assert rel_err(tpos.grad.cpu().numpy(), dpos) <= 4 * GRAD_REL
    tpos2 = T(pos, grad=True)          # grad_db=False: only the (u, v) path
    r2, db2 = dr.rasterize(ctx, tpos2, ttri, (H, W), grad_db=False)
    t2, d2 = dr.interpolate(T(vt[None]), r2, ttri, rast_db=db2, diff_attrs='all')
    (dr.texture(T(tex), t2, d2) * T(g)).sum().backward()
    assert rel_err(tpos2.grad.cpu().numpy(), dpos_uv) <= 4 * GRAD_REL


@pytest.mark.gpu
def test_depth_peeler_layers_match_oracle_and_are_differentiable(gpu):
    """InstantMesh's use (models/geometry/render/neural_render.py:103-106) plus what it does not exercise: layers > 0."""
    import nvdiffrast.torch as dr
    H, W = 96, 80
    v, f, vt, vn = S.make_uv_sphere(16, 24, radius=0.7, displacement=0.25)
    pos, _, _ = S.mesh_clip_positions(v, -20.0, 35.0, 2.0, W, H)
    ctx = dr.RasterizeCudaContext()
    tpos, ttri = T(pos, grad=True), T(f, torch.int32)
    plain, plain_db = dr.rasterize(ctx, tpos, ttri, (H, W))
    layers, oprev = [], None
    with dr.DepthPeeler(ctx, tpos, ttri, (H, W)) as peeler:
        with pytest.raises(RuntimeError):
            dr.rasterize(ctx, tpos, ttri, (H, W))                    # the context is busy peeling
        for k in range(4):
            rast, db = peeler.rasterize_next_layer()
            orast, odb = M.rasterize_next_layer(pos, f, (H, W), oprev)
            r = rast.detach().cpu().numpy()
            same = r[..., 3] == orast[..., 3]
            assert (~same).sum() <= 4, (k, (~same).sum())            # depth near-ties between float32 pipelines
            assert np.abs(r[same][:, :3] - orast[same][:, :3]).max() <= 1e-3
            assert np.abs(db.detach().cpu().numpy()[same] - odb[same]).max() <= 1e-3 * max(1.0, np.abs(odb).max())
            layers.append(rast)
            oprev = orast                                            # each pipeline peels against its own previous layer
    assert torch.equal(layers[0], plain)                             # layer 0 is rasterize()
    dr.rasterize(ctx, tpos, ttri, (H, W))                            # and the context is free again
    cov = [(l[..., 3] > 0) for l in layers]
    assert cov[1].sum() > 0.5 * cov[0].sum()                         # the back of the sphere
    for k in range(1, 4):
        assert (cov[k] & ~cov[k - 1]).sum() == 0                     # nothing appears where the previous layer was empty
        both = cov[k]
        assert (layers[k][..., 2][both] > layers[k - 1][..., 2][both]).all()     # strictly behind
    with pytest.raises(RuntimeError):
        peeler.rasterize_next_layer()                                # outside the with block
    # gradients flow through a peeled layer exactly as through rasterize(): d(sum g*(u,v)) / dpos against the float64 oracle
    g = np.random.default_rng(4).normal(size=(1, H, W, 4)).astype(np.float32); g[..., 2:] = 0
    r64_0, _ = M.rasterize_next_layer(pos, f, (H, W), None, dtype=np.float64)
    r64_1, _ = M.rasterize_next_layer(pos, f, (H, W), r64_0, dtype=np.float64)
    ok = r64_1[..., 3] == layers[1].detach().cpu().numpy()[..., 3]
    g64 = g.astype(np.float64) * ok[..., None]
    dpos = M.rasterize_bwd(pos, f, r64_1, g64, dtype=np.float64)
    (layers[1] * T(g64.astype(np.float32))).sum().backward()
    assert rel_err(tpos.grad.cpu().numpy(), dpos) <= 2 * GRAD_REL


@pytest.mark.gpu
def test_depth_peeler_in_range_mode(gpu):
    """DepthPeeler(glctx, pos [V,4], tri, res, ranges): every minibatch item peels its own triangle range of one shared vertex buffer; triangle ids index
    the full index buffer.  Each item and layer against the oracle peeling that sub-mesh alone; the vertex gradient is the sum over the items."""
    import nvdiffrast.torch as dr
    H, W = 80, 72
    v, f, vt, vn = S.make_uv_sphere(14, 20, radius=0.7, displacement=0.25)
    pos3, _, _ = S.mesh_clip_positions(v, -20.0, 35.0, 2.0, W, H)
    pos = pos3[0]
    Tn = f.shape[0]
    ranges = np.array([[0, Tn], [Tn // 3, Tn - Tn // 3], [7, 0], [0, Tn // 2]], np.int32)
    ctx = dr.RasterizeCudaContext()
    tpos, ttri = T(pos, grad=True), T(f, torch.int32)
    with pytest.raises(ValueError):
        dr.DepthPeeler(ctx, tpos[None], ttri, (H, W), ranges=torch.tensor(ranges))          # range mode wants pos [V,4]
    with pytest.raises(ValueError):
        dr.DepthPeeler(ctx, tpos, ttri, (H, W), ranges=torch.tensor([[0, Tn + 1]]))
    oprev = [None] * len(ranges)
    layers = []
    with dr.DepthPeeler(ctx, tpos, ttri, (H, W), ranges=torch.tensor(ranges)) as peeler:
        for k in range(3):
            rast, db = peeler.rasterize_next_layer()
            assert rast.shape == (len(ranges), H, W, 4) and db.shape == rast.shape
            r = rast.detach().cpu().numpy()
            for b, (start, count) in enumerate(ranges.tolist()):
                if count == 0:
                    assert (r[b] == 0).all()
                    continue
                orast, odb = M.rasterize_next_layer(pos[None], f[start:start + count], (H, W), oprev[b])
                oprev[b] = orast
                oid = np.where(orast[0, ..., 3] > 0, orast[0, ..., 3] + start, 0)                # ids index the full `tri`
                same = r[b, ..., 3] == oid
                assert (~same).sum() <= 4, (k, b, (~same).sum())
                assert np.abs(r[b][same][:, :3] - orast[0][same][:, :3]).max() <= 1e-3
            layers.append(rast)
    plain, _ = dr.rasterize(ctx, tpos, ttri, (H, W), ranges=torch.tensor(ranges))
    assert torch.equal(layers[0], plain)                              # layer 0 is rasterize(..., ranges)
    with dr.DepthPeeler(ctx, tpos.detach()[None], ttri, (H, W)) as whole:
        whole.rasterize_next_layer()
        second, _ = whole.rasterize_next_layer()
    assert torch.equal(layers[1][0].detach(), second[0])              # the item that covers everything peels like the plain peeler
    # gradient of a peeled layer w.r.t. the shared vertex buffer: sum over the items (float64 oracle, where both pipelines drew the same triangle)
    g = np.random.default_rng(8).normal(size=(len(ranges), H, W, 4)).astype(np.float32); g[..., 2:] = 0
    d = np.float64
    dpos = np.zeros((1,) + pos.shape, d)
    r1 = layers[1].detach().cpu().numpy()
    gm = np.zeros_like(g)
    for b, (start, count) in enumerate(ranges.tolist()):
        if count == 0:
            continue
        fs = f[start:start + count]
        o0, _ = M.rasterize_next_layer(pos[None], fs, (H, W), None, dtype=d)
        o1, _ = M.rasterize_next_layer(pos[None], fs, (H, W), o0, dtype=d)
        ok = np.where(o1[0, ..., 3] > 0, o1[0, ..., 3] + start, 0) == r1[b, ..., 3]
        gm[b] = g[b] * ok[..., None]
        dpos += M.rasterize_bwd(pos[None], fs, o1, gm[b:b + 1].astype(d), dtype=d)
    (layers[1] * T(gm)).sum().backward()
    assert rel_err(tpos.grad.cpu().numpy(), dpos[0]) <= 2 * GRAD_REL


@pytest.mark.gpu
def test_range_mode_rasterize_interpolate_antialias(gpu):
    """rasterize(pos [V,4], tri, res, ranges) -> interpolate(attr [V,A]) -> antialias(pos [V,4]): forward against the oracle's range
    mode, gradients (summed over the minibatch by autograd) against the float64 oracle chain."""
    import nvdiffrast.torch as dr
    H, W = 64, 72
    v, f, vt, vn = S.make_uv_sphere(12, 20, radius=0.7, displacement=0.2)
    pos3, _, _ = S.mesh_clip_positions(v, -20.0, 35.0, 2.0, W, H)
    pos = pos3[0]
    Tn = f.shape[0]
    ranges = np.array([[0, Tn // 3], [Tn // 3, Tn - Tn // 3], [0, Tn], [5, 0], [Tn // 4, Tn // 2]], np.int32)
    rng = np.random.default_rng(6)
    attr = rng.normal(size=(pos.shape[0], 3)).astype(np.float32)
    ctx = dr.RasterizeCudaContext()
    tpos, tattr, ttri = T(pos, grad=True), T(attr, grad=True), T(f, torch.int32)
    rast, db = dr.rasterize(ctx, tpos, ttri, (H, W), ranges=torch.tensor(ranges))
    col, _ = dr.interpolate(tattr, rast, ttri)
    aa = dr.antialias(col, rast, tpos, ttri)
    orast, odb = M.rasterize_ranges(pos, f, (H, W), ranges)
    r = rast.detach().cpu().numpy()
    assert r.shape == orast.shape == (5, H, W, 4)
    same = r[..., 3] == orast[..., 3]
    assert (~same).sum() <= 6 and (r[3] == 0).all()
    assert np.abs(r[same][:, :3] - orast[same][:, :3]).max() <= 1e-3
    full, _ = dr.rasterize(ctx, tpos.detach()[None], ttri, (H, W))
    assert torch.equal(rast[2].detach(), full[0])                      # the range covering everything is the plain call
    ocol, _ = M.interpolate(attr, orast, f)
    oaa = np.concatenate([M.antialias(ocol[b:b + 1], orast[b:b + 1], pos[None], f) for b in range(5)], 0)
    assert np.abs(col.detach().cpu().numpy() - ocol).mean() <= IMG_L1 and np.abs(aa.detach().cpu().numpy() - oaa).mean() <= IMG_L1
    g = rng.normal(size=oaa.shape).astype(np.float32)
    d = np.float64
    r64, _ = M.rasterize_ranges(pos, f, (H, W), ranges, dtype=d)
    g64 = g.astype(d) * (r64[..., 3:] == r[..., 3:])                   # compare where both pipelines drew the same triangle
    (aa * T(g64.astype(np.float32))).sum().backward()
    col64, _ = M.interpolate(attr, r64, f, dtype=d)
    dpos = np.zeros((1,) + pos.shape, d); dattr = np.zeros(attr.shape, d)
    for b in range(5):
        dcol, dp = M.antialias_bwd(col64[b:b + 1], r64[b:b + 1], pos[None], f, g64[b:b + 1], dtype=d)
        da, drast = M.interpolate_bwd(attr, r64[b:b + 1], f, dcol, dtype=d)
        dpos += dp + M.rasterize_bwd(pos[None], f, r64[b:b + 1], drast, dtype=d)
        dattr += da
    assert rel_err(tattr.grad.cpu().numpy(), dattr) <= GRAD_REL
    assert rel_err(tpos.grad.cpu().numpy(), dpos[0]) <= 2 * GRAD_REL
