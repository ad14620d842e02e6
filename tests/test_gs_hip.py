"""Parity tests proper (-m gpu): the HIP path, called through the C-ABI, against the CPU oracle on the same seeded
inputs.  Tolerances are the north star's: image L1 <= 1e-4, gradients <= 1e-3 relative, integers exact.  "Relative" for gradients is
held three ways against the FLOAT64 oracle (helpers.assert_grad_close): per-tensor relative L2, the element-wise bound
|got - ref| <= 1e-3 |ref| + 1e-6 max|ref|, and the max-norm of round 1."""
import ctypes as C
import os

import numpy as np
import pytest
import torch

from c3d_hip import synthetic as S
from oracle import gs_oracle as O
from helpers import GS_KEYS, assert_grad_close, hip_forward, hip_settings, oracle_forward, rel_err

pytestmark = pytest.mark.gpu
IMG_L1 = 1e-4
GRAD_REL = 1e-3
# element-wise gradient bound (helpers.assert_grad_close): share of elements allowed outside 1e-3 |ref| + 1e-6 max|ref| and the hard cap in
# units of that tolerance.  The float32 ORACLE itself (same arithmetic on the CPU) leaves 0 .. 2e-4 of the elements outside at worst 1.1 x
# on these scenes (measured in the build container); the kernels add v_exp_f32 / v_rcp_f32 (1 ulp) on top.
SMALL_FRAC, SMALL_HARD = 2e-3, 20.0
LARGE_FRAC, LARGE_HARD = 1e-2, 200.0


@pytest.fixture(scope="module", autouse=True)
def _need_gpu(oracle_built):
    if not torch.cuda.is_available():
        pytest.fail("-m gpu tests need a HIP device (no CPU fallback exists)")
    import c3d_hip
    c3d_hip.lib()


def _dev(a, dtype):
    return torch.tensor(a, dtype=dtype, device="cuda")


# ---------------------------------------------------------------- binning primitives
@pytest.mark.parametrize("n", [1, 63, 64, 2048, 2049, 5000, 1 << 20, (1 << 22) + 12345])
@pytest.mark.parametrize("excl", [0, 1])
def test_scan(n, excl):
    import c3d_hip as h
    rng = np.random.default_rng(n)
    a = rng.integers(0, 9, size=n, dtype=np.uint32)
    x = torch.tensor(a.astype(np.int64), device="cuda").to(torch.int32)   # same bits
    y = torch.empty_like(x)
    h.check(h.lib().c3d_test_scan_u32(h.ptr(x), h.ptr(y), n, excl, h.stream()), "scan")
    ref = np.cumsum(a.astype(np.uint64))
    if excl:
        ref = ref - a
    got = y.cpu().numpy().view(np.uint32)
    assert (got == (ref & 0xFFFFFFFF).astype(np.uint32)).all()


@pytest.mark.parametrize("n", [1, 63, 1024, 1025, 65543, 1 << 20, 4_200_000 + 777, 9_000_001])
def test_record_base_scan_one_wave_tiles(n):
    """csrc/scan_wave.h -- the record-base scan that rides in the recording forward compositing launch -- on its own: one-wave tiles of 1024 elements, two-level direct
    hand-over (groups of 64 tiles; beyond 64 groups, i.e. above 4.19 M elements, a tile sums the group totals in more than one step of 64)"""
    import c3d_hip as h
    rng = np.random.default_rng(n)
    a = rng.integers(0, 7, size=n).astype(np.uint32)
    a[rng.random(n) < 0.3] = 0
    rect = rng.integers(0, 1 << 30, size=(n, 2)).astype(np.uint32)
    d_in, d_rect = _dev(a.view(np.int32), torch.int32), _dev(rect.view(np.int32), torch.int32)
    out = torch.full((n,), -1, dtype=torch.int32, device="cuda")
    einfo = torch.full((n, 4), -1, dtype=torch.int32, device="cuda")
    h.check(h.lib().c3d_test_scan_wave(h.ptr(d_in), h.ptr(d_rect), h.ptr(out), h.ptr(einfo), n, h.stream()), "c3d_test_scan_wave")
    want = np.concatenate([[0], np.cumsum(a.astype(np.int64))[:-1]]).astype(np.uint32)
    got = out.cpu().numpy().view(np.uint32)
    assert (got == want).all()
    e = einfo.cpu().numpy().view(np.uint32)
    nz = a != 0
    assert (e[nz, 0] == 0).all() and (e[nz, 1] == rect[nz, 0]).all() and (e[nz, 2] == rect[nz, 1]).all() and (e[nz, 3] == want[nz]).all()
    assert (e[~nz] == 0xFFFFFFFF).all()                       # elements without tiles get no record


@pytest.mark.parametrize("n,bits", [(1, 32), (100, 32), (4096, 32), (4097, 8), (100000, 13), (1 << 20, 32), (3000001, 15), (9000001, 16)])
def test_sort_pairs_is_stable_and_correct(n, bits):
    """(the last case is more than two rounds of tiles for the chip: the pass instance whose workgroups stay and draw tile after tile, with one view)"""
    import c3d_hip as h
    rng = np.random.default_rng(n + bits)
    hi = (1 << bits) - 1
    k = rng.integers(0, min(hi, 1 << 31), size=n, dtype=np.uint32) & np.uint32(hi)
    if n > 10:
        k[rng.integers(0, n, size=n // 3)] = k[0]          # many ties -> exercises stability
    v = np.arange(n, dtype=np.uint32)
    kt = torch.tensor(k.astype(np.int64), device="cuda").to(torch.int32)
    vt = torch.tensor(v.astype(np.int64), device="cuda").to(torch.int32)
    h.check(h.lib().c3d_test_sort_pairs_u32(h.ptr(kt), h.ptr(vt), n, bits, h.stream()), "sort")
    order = np.argsort(k, kind="stable")
    assert (kt.cpu().numpy().view(np.uint32) == k[order]).all()
    assert (vt.cpu().numpy().view(np.uint32) == v[order]).all()


# ---------------------------------------------------------------- forward parity
@pytest.mark.parametrize("n,bits", [(1, 32), (63, 32), (64, 8), (65, 32), (1023, 13), (1024, 32), (1025, 32), (10000, 32), (16383, 32), (16384, 32), (16384, 5), (16385, 32), (40000, 32)])
def test_sort_with_index_values_small_and_large(n, bits):
    """the form of the sort the depth ordering and the topology builders use (values = element indices): up to 16384 keys ONE workgroup per view does every pass through LDS
    (k_sort_small, round 6), above that the histogram + onesweep launches -- both stable, both against numpy's stable argsort; duplicate-heavy keys included"""
    import c3d_hip as h
    rng = np.random.default_rng(n * 31 + bits)
    hi = (1 << bits) - 1 if bits < 32 else 0xFFFFFFFF
    for kind in ("uniform", "few"):
        a = (rng.integers(0, hi + 1, size=n, dtype=np.uint64) if kind == "uniform" else rng.integers(0, 7, size=n, dtype=np.uint64) * (hi // 7 if hi >= 7 else 1)).astype(np.uint32)
        k = torch.tensor(a.astype(np.int64), device="cuda").to(torch.int32)
        v = torch.full((n,), -1, dtype=torch.int32, device="cuda")
        h.check(h.lib().c3d_test_sort_iota_u32(h.ptr(k), h.ptr(v), n, bits, h.stream()), "sort")
        order = np.argsort(a, kind="stable")
        assert (k.cpu().numpy().view(np.uint32) == a[order]).all(), (n, bits, kind)
        assert (v.cpu().numpy() == order).all(), (n, bits, kind)


CASES = [
    dict(N=48, W=48, H=32, el=-20, az=30, rad=2.0, seed=7, deg=3),
    dict(N=300, W=100, H=70, el=35, az=200, rad=1.6, seed=3, deg=2),     # W,H not multiples of 16
    dict(N=500, W=64, H=64, el=0, az=0, rad=1.2, seed=11, deg=0),
    dict(N=2000, W=160, H=96, el=-60, az=-135, rad=2.5, seed=5, deg=1, scale=0.03),
]


def _scene(c):
    sc = S.make_small_scene(N=c["N"], seed=c["seed"], scale=c.get("scale", 0.08))
    st = S.camera_settings(c["W"], c["H"], 49.1, c["el"], c["az"], c["rad"], bg=(0.3, 0.7, 0.1), sh_degree=c["deg"])
    return sc, st


@pytest.mark.parametrize("c", CASES)
def test_forward_matches_oracle(c):
    sc, st = _scene(c)
    color, radii, depth, alpha, *_ = hip_forward(sc, st)
    oc, orad, od, oa, ostate = oracle_forward(sc, st, dtype=np.float32)
    assert (radii.cpu().numpy() == orad).all()
    assert np.abs(color.cpu().numpy() - oc).mean() <= IMG_L1
    assert np.abs(alpha.cpu().numpy() - oa).mean() <= IMG_L1
    assert np.abs(depth.cpu().numpy() - od).mean() <= IMG_L1
    assert np.abs(color.cpu().numpy() - oc).max() <= 5e-3   # no isolated wrong pixels either


def test_internal_state_matches_oracle():
    """Projected records match the oracle; per-tile lists are the oracle's lists minus provably non-contributing pairs."""
    import c3d_hip as h
    c = CASES[1]
    sc, st = _scene(c)
    N, H, W = c["N"], c["H"], c["W"]
    lib = h.lib()
    keep = []
    import diff_gaussian_rasterization as dgr
    rs = hip_settings(st, "cuda")
    stc = dgr._settings_struct(rs, keep)
    t = {k: _dev(sc[k], torch.float32) for k in ("means3D", "opacities", "shs", "scales", "rotations")}
    radii = torch.empty(N, dtype=torch.int32, device="cuda")
    geom = torch.empty(lib.c3d_gs_geom_bytes(N), dtype=torch.uint8, device="cuda")
    nr = C.c_int64(0)
    h.check(lib.c3d_gs_forward_project(C.byref(stc), N, 16, h.ptr(t["means3D"]), h.ptr(t["shs"]), None, h.ptr(t["opacities"]),
                                       h.ptr(t["scales"]), h.ptr(t["rotations"]), None, h.ptr(radii), h.ptr(geom), C.byref(nr), h.stream()), "project")
    D = nr.value
    binning = torch.empty(lib.c3d_gs_binning_bytes(D, H, W), dtype=torch.uint8, device="cuda")
    img = torch.empty(lib.c3d_gs_image_bytes(H, W), dtype=torch.uint8, device="cuda")
    out = [torch.empty(s, device="cuda") for s in ((3, H, W), (1, H, W), (1, H, W))]
    h.check(lib.c3d_gs_forward_render(C.byref(stc), N, 16, h.ptr(radii), h.ptr(geom), D, h.ptr(binning), h.ptr(img),
                                      h.ptr(out[0]), h.ptr(out[1]), h.ptr(out[2]), h.stream()), "render")
    tiles = ((W + 15) // 16) * ((H + 15) // 16)
    pl = torch.empty(max(D, 1), dtype=torch.int32, device="cuda")
    rg = torch.empty(tiles * 2, dtype=torch.int32, device="cuda")
    xy = torch.empty(N * 2, device="cuda"); dep = torch.empty(N, device="cuda"); co = torch.empty(N * 4, device="cuda")
    rgb = torch.empty(N * 3, device="cuda"); tt = torch.empty(N, dtype=torch.int32, device="cuda")
    h.check(lib.c3d_gs_debug_state(N, H, W, h.ptr(geom), D, h.ptr(binning), h.ptr(pl), h.ptr(rg), h.ptr(xy), h.ptr(dep), h.ptr(co),
                                   h.ptr(rgb), h.ptr(tt), h.stream()), "debug_state")
    oc, orad, od, oa, ostate = oracle_forward(sc, st, dtype=np.float32)
    g, b = ostate.geometry(), ostate.binning()
    # The HIP path emits a splat only to tiles where its alpha can reach 1/255 (an exact narrowing of the
    # dependency's bounding square).  So: every HIP tile list is a SUBSEQUENCE of the oracle's list (same
    # depth order), and every pair it drops contributes nothing to any pixel of that tile.
    assert D <= ostate.num_rendered
    assert (tt.cpu().numpy().view(np.uint32) <= g["tiles_touched"]).all()
    hpl = pl.cpu().numpy()[:D].view(np.uint32)
    hrg = rg.cpu().numpy().view(np.uint32).reshape(-1, 2)
    gx = (W + 15) // 16
    dropped = 0
    for t in range(tiles):
        ol = b["point_list"][b["ranges"][t, 0]:b["ranges"][t, 1]]
        hl = hpl[hrg[t, 0]:hrg[t, 1]]
        keep = np.isin(ol, hl)
        assert (ol[keep] == hl).all(), "tile %d: not an order-preserving subsequence" % t
        drop = ol[~keep]
        dropped += drop.size
        if drop.size:
            ys, xs = np.meshgrid(np.arange(16) + 16 * (t // gx), np.arange(16) + 16 * (t % gx), indexing="ij")
            dx = g["xy"][drop, 0][:, None, None].astype(np.float64) - xs[None]
            dy = g["xy"][drop, 1][:, None, None].astype(np.float64) - ys[None]
            cq = g["conic_opacity"][drop].astype(np.float64)
            power = -0.5 * (cq[:, 0, None, None] * dx * dx + cq[:, 2, None, None] * dy * dy) - cq[:, 1, None, None] * dx * dy
            alpha = cq[:, 3, None, None] * np.exp(np.minimum(power, 0))
            assert (alpha < 1.0 / 255.0).all() or (power[alpha >= 1 / 255.0] > 0).all(), "tile %d: dropped a contributing splat" % t
    assert dropped == ostate.num_rendered - D
    vis = orad > 0
    np.testing.assert_allclose(xy.cpu().numpy().reshape(N, 2)[vis], g["xy"][vis], atol=2e-3)
    np.testing.assert_allclose(dep.cpu().numpy()[vis], g["depths"][vis], rtol=1e-6)
    np.testing.assert_allclose(co.cpu().numpy().reshape(N, 4)[vis], g["conic_opacity"][vis], rtol=2e-4, atol=1e-6)
    np.testing.assert_allclose(rgb.cpu().numpy().reshape(N, 3)[vis], g["rgb"][vis], atol=2e-6)


def test_precomputed_colour_and_covariance_paths():
    sc, st = _scene(CASES[0])
    _, _, _, _, ostate = oracle_forward(sc, st, dtype=np.float64)
    q = sc["rotations"].astype(np.float64); s = sc["scales"].astype(np.float64)
    r, x, y, z = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
    Rm = np.stack([1 - 2 * (y * y + z * z), 2 * (x * y - r * z), 2 * (x * z + r * y), 2 * (x * y + r * z), 1 - 2 * (x * x + z * z),
                   2 * (y * z - r * x), 2 * (x * z - r * y), 2 * (y * z + r * x), 1 - 2 * (x * x + y * y)], -1).reshape(-1, 3, 3)
    Mm = Rm * s[:, None, :]
    Sg = Mm @ Mm.transpose(0, 2, 1)
    sc2 = {"means3D": sc["means3D"], "opacities": sc["opacities"], "colors_precomp": ostate.geometry()["rgb"].astype(np.float32),
           "cov3D_precomp": np.stack([Sg[:, 0, 0], Sg[:, 0, 1], Sg[:, 0, 2], Sg[:, 1, 1], Sg[:, 1, 2], Sg[:, 2, 2]], 1).astype(np.float32)}
    color, radii, depth, alpha, inp, m2d = hip_forward(sc2, st, requires_grad=True)
    oc, orad, od, oa, ost = oracle_forward(sc2, st, dtype=np.float64)
    assert np.abs(color.detach().cpu().numpy() - oc).mean() <= IMG_L1
    gC = np.random.default_rng(0).normal(size=oc.shape)
    color.backward(torch.tensor(gC, dtype=torch.float32, device="cuda"))
    og = O.backward(ost, gC)
    assert rel_err(inp["colors_precomp"].grad.cpu().numpy(), og["colors"]) <= GRAD_REL
    assert rel_err(inp["cov3D_precomp"].grad.cpu().numpy(), og["cov3D"]) <= GRAD_REL
    assert rel_err(inp["means3D"].grad.cpu().numpy(), og["means3D"]) <= GRAD_REL


def test_tile_grids_wider_than_255_take_the_unpacked_rect_path():
    """Up to 255 x 255 tiles (4080 px a side) the tile rect of a Gaussian travels in four bytes and the backward pass's per-Gaussian record info in eight (c3d_rect_pack,
    GsParams::rect4); a wider grid keeps the 8- / 16-byte forms.  A 4128 x 512 strip (258 x 32 tiles) through forward and backward against the float64 oracle, and the same
    scene at 4064 x 512 (254 tiles: packed) for the pair."""
    sc = S.make_small_scene(N=2500, seed=9, scale=0.06)
    for W in (4128, 4064):
        st = S.camera_settings(W, 512, 20.0, 5.0, 20.0, 2.0)
        color, radii, depth, alpha, inp, m2d = hip_forward(sc, st, requires_grad=True)
        oc, orad, od, oa, ost = oracle_forward(sc, st, dtype=np.float64)
        assert ost.num_rendered > 2000
        assert np.abs(color.detach().cpu().numpy() - oc).mean() <= IMG_L1 and (radii.cpu().numpy() == orad).all()
        assert np.abs(alpha.detach().cpu().numpy() - oa).mean() <= IMG_L1 and np.abs(depth.detach().cpu().numpy() - od).mean() <= IMG_L1
        gC = np.random.default_rng(1).normal(size=oc.shape).astype(np.float32)
        (color * _dev(gC, torch.float32)).sum().backward()
        og = O.backward(ost, gC)
        for k in ("means3D", "opacities", "shs", "scales", "rotations"):      # (an 8 : 1 strip at 20 degrees vertically is a 110 degree horizontal field of view)
            assert rel_err(inp[k].grad.cpu().numpy(), og[k]) <= 2 * GRAD_REL, (W, k)
        assert rel_err(m2d.grad.cpu().numpy(), og["means2D"]) <= 2 * GRAD_REL, W


def test_call_patterns_of_the_other_consumers_in_the_reference():
    """The rasterizer has three more callers in the reference tree besides the MVs path; each drives it a little differently.  Their call sequences,
    statement by statement, against the oracle (forward and, where the caller differentiates, backward):
      LGM            Gen_3D_Modules/LGM/core/gs.py:27-90            per (batch, view) loop, sh_degree = 0, colours given (colors_precomp), plain zeros for means2D
      TriplaneGaussian  .../TriplaneGaussian/models/renderer.py:205-275   means2D = zeros(requires_grad) + 0 with retain_grad() (a NON-leaf), float32 autocast around the call
      TRELLIS        .../TRELLIS/trellis/renderers/gaussian_render.py:55-135   covariance built in Python (cov3D_precomp, with a scaling modifier), colours from a Python SH evaluation"""
    from diff_gaussian_rasterization import GaussianRasterizationSettings, GaussianRasterizer
    sc = S.make_small_scene(N=300)
    st = S.camera_settings(160, 112, 49.1, 15.0, 40.0, 2.2, sh_degree=0)
    dev = "cuda"
    t = lambda a: torch.tensor(np.ascontiguousarray(a), dtype=torch.float32, device=dev)
    rgb = np.clip(sc["shs"][:, 0, :] * 0.28209479177387814 + 0.5, 0, None).astype(np.float32)
    # --- LGM: a [B, N, 14] tensor sliced per batch entry, two views per entry
    gauss = torch.cat([t(sc["means3D"]), t(sc["opacities"]), t(sc["scales"]), t(sc["rotations"]), t(rgb)], dim=1)[None].repeat(2, 1, 1)
    for b in range(2):
        means3D, opacity = gauss[b, :, 0:3].contiguous().float(), gauss[b, :, 3:4].contiguous().float()
        scales, rotations, rgbs = gauss[b, :, 4:7].contiguous().float(), gauss[b, :, 7:11].contiguous().float(), gauss[b, :, 11:].contiguous().float()
        for az in (40.0, 160.0):
            stv = S.camera_settings(160, 112, 49.1, 15.0, az, 2.2, sh_degree=0)
            rs = GaussianRasterizationSettings(image_height=112, image_width=160, tanfovx=stv["tanfovx"], tanfovy=stv["tanfovy"], bg=t(stv["bg"]), scale_modifier=1,
                                               viewmatrix=t(stv["viewmatrix"]).reshape(4, 4).float(), projmatrix=t(stv["projmatrix"]).reshape(4, 4).float(), sh_degree=0,
                                               campos=t(stv["campos"]).float(), prefiltered=False, debug=False)
            img, radii, depth, alpha = GaussianRasterizer(raster_settings=rs)(means3D=means3D, means2D=torch.zeros_like(means3D, dtype=torch.float32, device=dev), shs=None,
                                                                               colors_precomp=rgbs, opacities=opacity, scales=scales, rotations=rotations, cov3D_precomp=None)
            oc, orad, od, oa, _ = O.forward(sc["means3D"], sc["opacities"], stv, colors_precomp=rgb, scales=sc["scales"], rotations=sc["rotations"])
            assert np.abs(img.cpu().numpy() - oc).mean() <= IMG_L1 and (radii.cpu().numpy() == orad).all() and np.abs(alpha.cpu().numpy() - oa).mean() <= IMG_L1
    # --- TriplaneGaussian: non-leaf screen-space points that keep their gradient, autocast(float32) around the rasterizer
    st3 = S.camera_settings(160, 112, 49.1, 15.0, 40.0, 2.2, sh_degree=3)
    xyz = t(sc["means3D"]).requires_grad_(True)
    shs, op, scl, rot = (t(sc[k]).requires_grad_(True) for k in ("shs", "opacities", "scales", "rotations"))
    screenspace_points = torch.zeros_like(xyz, dtype=xyz.dtype, requires_grad=True, device=dev) + 0
    screenspace_points.retain_grad()
    rs = GaussianRasterizationSettings(image_height=112, image_width=160, tanfovx=st3["tanfovx"], tanfovy=st3["tanfovy"], bg=t(st3["bg"]), scale_modifier=1.0,
                                       viewmatrix=t(st3["viewmatrix"]).reshape(4, 4), projmatrix=t(st3["projmatrix"]).reshape(4, 4).float(), sh_degree=3, campos=t(st3["campos"]),
                                       prefiltered=False, debug=False)
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        with torch.autocast(device_type="cuda", dtype=torch.float32):
            img, radii, depth, alpha = GaussianRasterizer(raster_settings=rs)(means3D=xyz, means2D=screenspace_points, shs=shs, colors_precomp=None, opacities=op, scales=scl,
                                                                               rotations=rot, cov3D_precomp=None)
    gC = np.random.default_rng(3).normal(size=(3, 112, 160)).astype(np.float32)
    (img.permute(1, 2, 0) * t(gC).permute(1, 2, 0)).sum().backward()
    oc, orad, od, oa, ost = O.forward(sc["means3D"], sc["opacities"], st3, shs=sc["shs"], scales=sc["scales"], rotations=sc["rotations"], dtype=np.float64)
    og = O.backward(ost, gC)
    assert np.abs(img.detach().cpu().numpy() - oc).mean() <= IMG_L1
    assert screenspace_points.grad is not None and rel_err(screenspace_points.grad.cpu().numpy(), og["means2D"]) <= GRAD_REL
    for k, q in (("means3D", xyz), ("shs", shs), ("opacities", op), ("scales", scl), ("rotations", rot)):
        assert rel_err(q.grad.cpu().numpy(), og[k]) <= GRAD_REL, k
    # --- TRELLIS: covariance and colours computed in Python (scaling modifier 0.8 folded into the covariance), override colour path
    m = 0.8
    q, s_ = sc["rotations"].astype(np.float64), sc["scales"].astype(np.float64) * m
    r, x, y, z = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
    Rm = np.stack([1 - 2 * (y * y + z * z), 2 * (x * y - r * z), 2 * (x * z + r * y), 2 * (x * y + r * z), 1 - 2 * (x * x + z * z),
                   2 * (y * z - r * x), 2 * (x * z - r * y), 2 * (y * z + r * x), 1 - 2 * (x * x + y * y)], -1).reshape(-1, 3, 3)
    Mm = Rm * s_[:, None, :]
    Sg = Mm @ Mm.transpose(0, 2, 1)
    cov6 = np.stack([Sg[:, 0, 0], Sg[:, 0, 1], Sg[:, 0, 2], Sg[:, 1, 1], Sg[:, 1, 2], Sg[:, 2, 2]], 1).astype(np.float32)
    ocol = O.forward(sc["means3D"], sc["opacities"], st3, shs=sc["shs"], scales=sc["scales"], rotations=sc["rotations"])[4].geometry()["rgb"].astype(np.float32)   # = clamp_min(eval_sh + 0.5, 0)
    rs = GaussianRasterizationSettings(image_height=112, image_width=160, tanfovx=st3["tanfovx"], tanfovy=st3["tanfovy"], bg=t(st3["bg"]), scale_modifier=m,
                                       viewmatrix=t(st3["viewmatrix"]).reshape(4, 4), projmatrix=t(st3["projmatrix"]).reshape(4, 4), sh_degree=3, campos=t(st3["campos"]),
                                       prefiltered=False, debug=False)
    cov_t, col_t = t(cov6).requires_grad_(True), t(ocol).requires_grad_(True)
    img, radii, depth, alpha = GaussianRasterizer(raster_settings=rs)(means3D=t(sc["means3D"]), means2D=torch.zeros((300, 3), device=dev, requires_grad=True) + 0, shs=None,
                                                                       colors_precomp=col_t, opacities=t(sc["opacities"]), scales=None, rotations=None, cov3D_precomp=cov_t)
    st_m = dict(st3, scale_modifier=m)
    oc, orad, od, oa, ost = O.forward(sc["means3D"], sc["opacities"], st_m, colors_precomp=ocol, cov3D_precomp=cov6, dtype=np.float64)
    assert np.abs(img.detach().cpu().numpy() - oc).mean() <= IMG_L1 and (radii.cpu().numpy() == orad).all()
    (img * t(gC)).sum().backward()
    og = O.backward(ost, gC)
    assert rel_err(cov_t.grad.cpu().numpy(), og["cov3D"]) <= GRAD_REL and rel_err(col_t.grad.cpu().numpy(), og["colors"]) <= GRAD_REL


# ---------------------------------------------------------------- backward parity
@pytest.mark.parametrize("c", CASES)
def test_backward_matches_oracle(c):
    sc, st = _scene(c)
    H, W = c["H"], c["W"]
    rng = np.random.default_rng(5)
    gC, gD, gA = (rng.normal(size=s).astype(np.float32) for s in ((3, H, W), (1, H, W), (1, H, W)))
    color, radii, depth, alpha, inp, m2d = hip_forward(sc, st, requires_grad=True)
    loss = (color * _dev(gC, torch.float32)).sum() + (depth * _dev(gD, torch.float32)).sum() + (alpha * _dev(gA, torch.float32)).sum()
    loss.backward()
    _, _, _, _, ost = oracle_forward(sc, st, dtype=np.float64)
    og = O.backward(ost, gC, gD, gA)
    for k in ("means3D", "opacities", "shs", "scales", "rotations"):
        assert rel_err(inp[k].grad.cpu().numpy(), og[k]) <= GRAD_REL, k
        assert_grad_close(inp[k].grad.cpu().numpy(), og[k], "%s N=%d" % (k, c["N"]), max_frac=SMALL_FRAC, hard=SMALL_HARD)
    assert rel_err(m2d.grad.cpu().numpy(), og["means2D"]) <= GRAD_REL
    assert_grad_close(m2d.grad.cpu().numpy(), og["means2D"], "means2D N=%d" % c["N"], max_frac=SMALL_FRAC, hard=SMALL_HARD)
    if c["deg"] < 3:
        assert inp["shs"].grad[:, (c["deg"] + 1) ** 2:].abs().max().item() == 0


def test_golden_vectors():
    z = np.load(os.path.join(os.path.dirname(__file__), "golden", "gs_small.npz"))
    sc = {k: z[k] for k in ("means3D", "opacities", "shs", "scales", "rotations")}
    st = {k: (z["st_" + k].item() if z["st_" + k].ndim == 0 else z["st_" + k]) for k in
          ("image_height", "image_width", "tanfovx", "tanfovy", "bg", "scale_modifier", "viewmatrix", "projmatrix", "sh_degree", "campos")}
    color, radii, depth, alpha, inp, m2d = hip_forward(sc, st, requires_grad=True)
    assert (radii.cpu().numpy() == z["radii"]).all()
    assert np.abs(color.detach().cpu().numpy() - z["color"]).mean() <= IMG_L1
    assert np.abs(depth.detach().cpu().numpy() - z["depth"]).mean() <= IMG_L1
    assert np.abs(alpha.detach().cpu().numpy() - z["alpha"]).mean() <= IMG_L1
    dv = lambda k: torch.tensor(z[k], device="cuda")
    ((color * dv("gC")).sum() + (depth * dv("gD")).sum() + (alpha * dv("gA")).sum()).backward()
    for k in ("means3D", "opacities", "shs", "scales", "rotations"):
        assert rel_err(inp[k].grad.cpu().numpy(), z["grad_" + k]) <= GRAD_REL, k
    assert rel_err(m2d.grad.cpu().numpy(), z["grad_means2D"]) <= GRAD_REL


# ---------------------------------------------------------------- BASELINE configs at oracle-friendly sizes
def test_config1_ball_10k_256px_4views():
    """BASELINE config 1: 10k Gaussians, 256x256, MVDream(4) orbit, white bg."""
    sc = S.make_ball_cloud(N=10000, seed=0)
    for az in (0.0, 90.0, 180.0, -90.0):
        st = S.camera_settings(256, 256, 49.1, 0.0, az, 1.75)
        color, radii, depth, alpha, *_ = hip_forward(sc, st)
        oc, orad, od, oa, _ = oracle_forward(sc, st, nthreads=8)
        assert (radii.cpu().numpy() != orad).sum() <= 2
        assert np.abs(color.cpu().numpy() - oc).mean() <= IMG_L1
        assert np.abs(alpha.cpu().numpy() - oa).mean() <= IMG_L1


def test_medium_cloud_fwd_bwd():
    """the 1M-cloud generator at 100k / 640x360: long per-tile lists, early termination, many rounds."""
    sc = S.make_cloud(100000, seed=1234, log_scale_mean=np.log(0.01))
    st = S.camera_settings(640, 360, 49.1, 30.0, 45.0, 2.2)
    color, radii, depth, alpha, inp, m2d = hip_forward(sc, st, requires_grad=True)
    oc, orad, od, oa, ost = oracle_forward(sc, st, dtype=np.float64, nthreads=os.cpu_count() or 8)
    assert (radii.cpu().numpy() != orad).sum() <= 5
    assert np.abs(color.detach().cpu().numpy() - oc).mean() <= IMG_L1
    assert np.abs(alpha.detach().cpu().numpy() - oa).mean() <= IMG_L1
    rng = np.random.default_rng(1)
    gC = rng.normal(size=oc.shape).astype(np.float32)
    color.backward(_dev(gC, torch.float32))
    og = O.backward(ost, gC, nthreads=os.cpu_count() or 8)
    for k in ("means3D", "opacities", "shs", "scales", "rotations"):
        assert rel_err(inp[k].grad.cpu().numpy(), og[k]) <= GRAD_REL, k       # against the float64 oracle: no slack factor
        assert_grad_close(inp[k].grad.cpu().numpy(), og[k], "%s 100k" % k, max_frac=LARGE_FRAC, hard=LARGE_HARD)


def test_baseline_size_1M_1080p_forward_backward_vs_float64_oracle():
    """BASELINE configs 2 / 3 at FULL size: one view (elevation 30, azimuth 22.5) of the 1,000,000-Gaussian cloud (seed 1234, SH degree 3) at
    1920 x 1080, forward and backward, HIP against the float64 oracle.  The oracle takes seconds on the GPU box's
    host cores."""
    sc = S.make_cloud(1_000_000, seed=1234)
    st = S.camera_settings(1920, 1080, 49.1, 30.0, 22.5, 2.2, bg=(1, 1, 1))
    nt = os.cpu_count() or 8
    color, radii, depth, alpha, inp, m2d = hip_forward(sc, st, requires_grad=True)
    oc, orad, od, oa, ost = oracle_forward(sc, st, dtype=np.float64, nthreads=nt)
    assert (radii.cpu().numpy() != orad).sum() <= 20                 # ceil(3 sigma) on a float32 / float64 boundary
    assert np.abs(color.detach().cpu().numpy() - oc).mean() <= IMG_L1
    assert np.abs(alpha.detach().cpu().numpy() - oa).mean() <= IMG_L1
    assert np.abs(depth.detach().cpu().numpy() - od).mean() <= IMG_L1
    assert np.abs(color.detach().cpu().numpy() - oc).max() <= 2e-2     # no isolated wrong pixels (a flipped 1/255 or 1e-4 decision moves one pixel by < this)
    rng = np.random.default_rng(7)
    gC = rng.normal(size=oc.shape).astype(np.float32)
    gA = rng.normal(size=oa.shape).astype(np.float32)
    ((color * _dev(gC, torch.float32)).sum() + (alpha * _dev(gA, torch.float32)).sum()).backward()
    og = O.backward(ost, gC, None, gA, nthreads=nt)
    # Relative L2 <= 1e-3 per tensor and the element-wise bound on all but a small share of the entries.  The max-norm is held to 1e-2 here,
    # not 1e-3: with 4 M (tile, splat) pairs a handful of per-pixel decisions (alpha >= 1/255, T < 1e-4, ceil(3 sigma)) fall differently in
    # float32 and float64, and one flipped pixel moves one entry by a pixel's worth of gradient.  That is a property of float32, not of the
    # kernel: the float32 build of the ORACLE against its float64 build shows max-norm 1.4e-3 / relative L2 1.6e-4 already at 100k Gaussians
    # (measured in the build container), the kernel 1.7e-4 / 3.5e-5 on the same scene (test_medium_cloud_fwd_bwd).
    for k in ("means3D", "opacities", "shs", "scales", "rotations"):
        r = assert_grad_close(inp[k].grad.cpu().numpy(), og[k], "%s 1M/1080p" % k, max_frac=LARGE_FRAC, hard=2000.0)      # round 2: 1e9.  Worst measured 628 x; where the tail comes from: tests/test_zz_baseline_1m.py
        assert r["max_norm"] <= 1e-2, (k, r)
    r = assert_grad_close(m2d.grad.cpu().numpy(), og["means2D"], "means2D 1M/1080p", max_frac=LARGE_FRAC, hard=2000.0)
    assert r["max_norm"] <= 1e-2, r


@pytest.mark.parametrize("exact", [False, True])
def test_scaling_modifier_gradient_convention(exact):
    """scaling_modifier != 1 (a public argument of render(), main_3DGS_renderer.py:830): dL/dscale as the dependency's backward returns it
    (no modifier factor; the default) and the exact derivative, asked for PER CALL through the settings (C3D_GS_FLAG_EXACT_DSCALE,
    diff_gaussian_rasterization.with_exact_dscale) -- the library keeps no process-wide switch.  Kernel and oracle agree in both conventions."""
    import diff_gaussian_rasterization as dgr
    from diff_gaussian_rasterization import GaussianRasterizer
    c = CASES[3]
    sc = S.make_small_scene(N=c["N"], seed=c["seed"], scale=c.get("scale", 0.08))
    st = S.camera_settings(c["W"], c["H"], 49.1, c["el"], c["az"], c["rad"], bg=(0.3, 0.7, 0.1), sh_degree=c["deg"])
    st["scale_modifier"] = 0.7
    gC = np.random.default_rng(9).normal(size=(3, c["H"], c["W"])).astype(np.float32)

    def run(rs):
        inp = {k: torch.tensor(sc[k], dtype=torch.float32, device="cuda", requires_grad=True) for k in ("means3D", "opacities", "shs", "scales", "rotations")}
        m2d = torch.zeros_like(inp["means3D"], requires_grad=True)
        color, radii, depth, alpha = GaussianRasterizer(rs)(means3D=inp["means3D"], means2D=m2d, opacities=inp["opacities"], shs=inp["shs"], scales=inp["scales"],
                                                            rotations=inp["rotations"])
        color.backward(_dev(gC, torch.float32))
        return color, radii, inp

    rs = hip_settings(st, "cuda")
    color, radii, inp = run(dgr.with_exact_dscale(rs) if exact else rs)
    old_o = O.set_exact_dscale(exact)
    try:
        oc, orad, od, oa, ost = oracle_forward(sc, st, dtype=np.float64)
        og = O.backward(ost, gC)
    finally:
        O.set_exact_dscale(old_o)
    assert (radii.cpu().numpy() == orad).all() and np.abs(color.detach().cpu().numpy() - oc).mean() <= IMG_L1
    for k in ("means3D", "opacities", "shs", "scales", "rotations"):
        assert_grad_close(inp[k].grad.cpu().numpy(), og[k], "%s mod=0.7 exact=%s" % (k, exact), max_frac=SMALL_FRAC, hard=SMALL_HARD)
    if not exact:      # the two conventions differ by exactly the modifier, and asking for one call does not change the next
        _, _, inp2 = run(dgr.with_exact_dscale(rs))
        assert torch.allclose(inp2["scales"].grad, 0.7 * inp["scales"].grad, rtol=1e-6, atol=0)
        _, _, inp3 = run(rs)
        assert torch.equal(inp3["scales"].grad, inp["scales"].grad)


# ---------------------------------------------------------------- full size: properties (no oracle run)
def test_full_size_properties_1M_1080p():
    sc = S.make_cloud(1_000_000, seed=1234)
    st = S.camera_settings(1920, 1080, 49.1, 30.0, 22.5, 2.2, bg=(1, 1, 1))
    c1, r1, d1, a1, *_ = hip_forward(sc, st)
    c2, r2, d2, a2, *_ = hip_forward(sc, st)
    assert torch.equal(c1, c2) and torch.equal(r1, r2) and torch.equal(a1, a2)      # forward is deterministic
    assert a1.min().item() >= 0 and a1.max().item() <= 1 + 1e-5
    assert torch.isfinite(c1).all()
    # affine in bg: C(bg) = C(0) + (1 - alpha) * bg
    st0 = dict(st, bg=np.zeros(3, np.float32))
    c0, _, _, a0, *_ = hip_forward(sc, st0)
    assert (c1 - (c0 + (1 - a0))).abs().max().item() <= 2e-6
    # input order does not matter beyond float association (depth ties are broken by index)
    perm = np.random.default_rng(0).permutation(1_000_000)
    scp = {k: v[perm] for k, v in sc.items()}
    cp, rp, dp, ap, *_ = hip_forward(scp, st)
    assert (cp - c1).abs().mean().item() <= 1e-6
    assert torch.equal(rp.cpu(), r1.cpu()[torch.tensor(perm)])
    assert (r1 > 0).sum().item() > 500000


# ---------------------------------------------------------------- edge cases
def test_edge_cases():
    from diff_gaussian_rasterization import GaussianRasterizer
    st = S.camera_settings(40, 24, 49.1, 0, 0, 2.0, bg=(0.1, 0.2, 0.3))
    bgv = torch.tensor([0.1, 0.2, 0.3], device="cuda").reshape(3, 1, 1)
    # N = 0
    sc0 = {"means3D": np.zeros((0, 3), np.float32), "opacities": np.zeros((0, 1), np.float32), "shs": np.zeros((0, 16, 3), np.float32),
           "scales": np.zeros((0, 3), np.float32), "rotations": np.zeros((0, 4), np.float32)}
    color, radii, depth, alpha, *_ = hip_forward(sc0, st)
    assert torch.allclose(color, bgv.expand_as(color)) and alpha.abs().max().item() == 0 and radii.numel() == 0
    # all culled (behind the camera) -> background, zero gradients
    sc = S.make_small_scene(N=10)
    sc["means3D"] = sc["means3D"] + np.array([0, 0, 10], np.float32)
    color, radii, depth, alpha, inp, m2d = hip_forward(sc, st, requires_grad=True)
    assert (radii == 0).all() and torch.allclose(color, bgv.expand_as(color))
    color.sum().backward()
    assert all(inp[k].grad.abs().max().item() == 0 for k in ("means3D", "opacities", "shs", "scales", "rotations"))
    # one huge Gaussian covering every tile
    big = {"means3D": np.zeros((1, 3), np.float32), "opacities": np.full((1, 1), 0.9, np.float32), "shs": np.ones((1, 16, 3), np.float32),
           "scales": np.full((1, 3), 5.0, np.float32), "rotations": np.array([[1, 0, 0, 0]], np.float32)}
    color, radii, depth, alpha, *_ = hip_forward(big, st)
    oc, orad, od, oa, _ = oracle_forward(big, st)
    assert (radii.cpu().numpy() == orad).all() and np.abs(alpha.cpu().numpy() - oa).max() < 1e-5
    # argument rules keep the dependency's messages
    rast = GaussianRasterizer(hip_settings(st, "cuda"))
    t = {k: _dev(sc[k], torch.float32) for k in sc}
    with pytest.raises(Exception, match="excatly one of either SHs"):
        rast(means3D=t["means3D"], means2D=None, opacities=t["opacities"], scales=t["scales"], rotations=t["rotations"])
    with pytest.raises(Exception, match="scale/rotation pair or precomputed 3D covariance"):
        rast(means3D=t["means3D"], means2D=None, opacities=t["opacities"], shs=t["shs"])
    # markVisible
    vis = rast.markVisible(_dev(S.make_small_scene(N=100)["means3D"] * 3, torch.float32))
    ref = O.mark_visible(S.make_small_scene(N=100)["means3D"] * 3, st["viewmatrix"], st["projmatrix"])
    assert (vis.cpu().numpy() == ref).all()
    # works under inference_mode (the orbit-renderer nodes run that way: nodes.py worker thread)
    with torch.inference_mode():
        c, *_ = hip_forward(S.make_small_scene(N=20), st)
    assert torch.isfinite(c).all()
    # ... and inside the float32 autocast wrapper one external caller uses (TriplaneGaussian/models/renderer.py:261,295): same bits
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        with torch.autocast("cuda", dtype=torch.float32):
            c2, *_ = hip_forward(S.make_small_scene(N=20), st)
    assert torch.equal(c, c2)


def test_backward_is_bit_reproducible():
    """no atomics anywhere in the backward pass: two runs give identical bits"""
    sc = S.make_cloud(200000, seed=3, log_scale_mean=np.log(0.008))
    st = S.camera_settings(800, 450, 49.1, 10.0, 100.0, 2.2)
    gC = _dev(np.random.default_rng(2).normal(size=(3, 450, 800)).astype(np.float32), torch.float32)
    grads = []
    for _ in range(2):
        color, radii, depth, alpha, inp, m2d = hip_forward(sc, st, requires_grad=True)
        ((color * gC).sum() + alpha.sum() + depth.sum()).backward()
        grads.append({k: inp[k].grad.clone() for k in ("means3D", "opacities", "shs", "scales", "rotations")})
    for k in grads[0]:
        assert torch.equal(grads[0][k], grads[1][k]), k


def _rasterize_node(color):
    node, seen, stack = None, set(), [color.grad_fn]
    while stack and node is None:
        f_ = stack.pop()
        if f_ is None or id(f_) in seen:
            continue
        seen.add(id(f_))
        if "Rasterize" in type(f_).__name__:
            node = f_
        stack.extend(x[0] for x in f_.next_functions)
    return node


def test_recorded_pair_activity_loses_nothing():
    """The forward compositing pass records, per sorted (tile, splat) pair and 8x8 quadrant of the tile (four byte planes, csrc/gs_internal.h:
    gs_pair_activity), whether the quadrant blended the splat into at least one pixel; the backward pass walks exactly those (quadrant, splat) pairs.
    Overwriting the planes with 'every quadrant blended every pair' makes the backward pass walk a superset -- lanes that did not blend contribute
    exact zeros -- so the gradients must come out bit for bit the same; and the recording must really prune (most pairs of a dense scene are blended
    nowhere or in part of the tile only)."""
    import diff_gaussian_rasterization as dgr
    sc = S.make_cloud(200000, seed=3, log_scale_mean=np.log(0.008))
    W, H = 800, 450
    st = S.camera_settings(W, H, 49.1, 10.0, 100.0, 2.2)
    gC = _dev(np.random.default_rng(2).normal(size=(3, H, W)).astype(np.float32), torch.float32)
    was = dgr.sync_free(False)      # the wheel's synchronous path: buffers carved for the exact pair count, which the plane arithmetic below relies on
    try:
        color, radii, depth, alpha, inp, m2d = hip_forward(sc, st, requires_grad=True)
    finally:
        dgr.sync_free(was)
    D = int(dgr.last_num_rendered)
    node = _rasterize_node(color)
    binning, img = node.saved_tensors[-2], node.saved_tensors[-1]
    # binning buffer (gs_carve_binning): tkey[0] | tkey[1] | tval[0] | tval[1] | ranges ...; the four planes live in the key buffer the sort's last pass did NOT read
    # (that pass writes no keys: the per-tile ranges are all that is left of them) -- 800 x 450 is 50 x 29 tiles, two passes, result index 0
    d_al = (4 * D + 255) // 256 * 256
    res = 0
    stride = d_al // 4
    dead = binning.data[(1 - res) * d_al:(2 - res) * d_al]      # .data: the in-place overwrite below must not trip autograd's version check of the saved buffer
    planes = [dead[w * stride:w * stride + D] for w in range(4)]
    loss = (color * gC).sum() + alpha.sum() + depth.sum()
    names = ("means3D", "opacities", "shs", "scales", "rotations")
    g1 = torch.autograd.grad(loss, [inp[k] for k in names] + [m2d], retain_graph=True)
    before = [pl.clone() for pl in planes]
    dead.fill_(0x01)
    g2 = torch.autograd.grad(loss, [inp[k] for k in names] + [m2d], retain_graph=True)
    for a, b, k in zip(g1, g2, names + ("means2D",)):
        assert torch.equal(a, b), k
    # what the recording pruned, over the list positions a quadrant is read at: below the deepest position one of its pixels blended (n_contrib)
    gx, gy = (W + 15) // 16, (H + 15) // 16
    ncon = img.data[(4 * W * H + 255) // 256 * 256:][:4 * W * H].view(torch.int32).reshape(H, W)
    pad = torch.zeros((gy * 16, gx * 16), dtype=torch.int32, device="cuda")
    pad[:H, :W] = ncon
    upto = pad.reshape(gy, 2, 8, gx, 2, 8).permute(0, 3, 1, 4, 2, 5).reshape(gy * gx, 4, 64).max(dim=2).values      # [tile, quadrant]
    ranges = binning.data[4 * d_al:4 * d_al + 8 * gx * gy].view(torch.int32).reshape(gx * gy, 2)
    starts, ends = (~ranges[:, 0]).long(), ranges[:, 1].long()
    nonempty = ends > 0      # {0, 0} in the stored words = no pair of that tile
    counts = torch.where(nonempty, ends - starts, torch.zeros_like(ends))
    assert int(counts.sum()) == D
    tile_of = torch.repeat_interleave(torch.arange(gx * gy, device="cuda"), counts)      # the sorted list is tile after tile
    assert bool((torch.arange(D, device="cuda") >= starts[tile_of]).all()) and bool((torch.arange(D, device="cuda") < ends[tile_of]).all())
    posn = torch.arange(D, device="cuda") - (~ranges[tile_of, 0]).long()      # the stored words are {~start, end} (gs_tile_range, csrc/gs_internal.h)
    valid = torch.stack([posn < upto[tile_of, w].long() for w in range(4)])
    bits = torch.stack([(before[w] & 1).bool() for w in range(4)]) & valid
    reached = valid.any(0)
    frac_any = float(bits.any(0)[reached].float().mean())
    quads = float(bits.sum()) / max(int(bits.any(0).sum()), 1)
    print("[activity] %d pairs; of the %d list positions a tile reached %.0f %% blended somewhere, %.2f quadrants each" % (D, int(reached.sum()), 100 * frac_any, quads))
    assert 0.05 < frac_any < 0.98 and 1.0 <= quads < 3.5


# ---------------------------------------------------------------- training-step pieces
def test_fused_adam_matches_torch():
    from c3d_hip.optim import FusedAdam
    torch.manual_seed(0)
    shapes = [(1000, 3), (1000, 1, 3), (1000, 15, 3), (1000, 1), (1003,)]   # incl. a size that is not a multiple of 4
    a = [torch.nn.Parameter(torch.randn(s, device="cuda")) for s in shapes]
    b = [torch.nn.Parameter(p.detach().clone()) for p in a]
    lrs = [1.6e-3, 2.5e-3, 1.25e-4, 0.05, 5e-3]
    oa = FusedAdam([{"params": [p], "lr": lr} for p, lr in zip(a, lrs)], lr=0.0, eps=1e-15)
    ob = torch.optim.Adam([{"params": [p], "lr": lr} for p, lr in zip(b, lrs)], lr=0.0, eps=1e-15)
    for it in range(5):
        for p, q in zip(a, b):
            g = torch.randn_like(p) * (10.0 ** (it - 2))
            p.grad, q.grad = g.clone(), g.clone()
        oa.step(); ob.step()
    for p, q in zip(a, b):
        assert torch.allclose(p, q, rtol=2e-5, atol=1e-7), (p - q).abs().max().item()
        assert torch.allclose(oa.state[p]["exp_avg_sq"], ob.state[q]["exp_avg_sq"], rtol=1e-6, atol=1e-12)
    # the one-launch update of all tensors (c3d_adam_step_multi, what FusedAdam.step issues) has the bits of one c3d_adam_step call per tensor; 20 tensors = two launches
    import c3d_hip as h
    lib = h.lib()
    shapes = [(257, 3), (1001,), (64, 15, 3), (5,)] * 5
    ps = [torch.randn(s, device="cuda") for s in shapes]
    gs = [torch.randn(s, device="cuda") for s in shapes]
    single = [(p.clone(), torch.zeros_like(p), torch.zeros_like(p)) for p in ps]
    multi = [(p.clone(), torch.zeros_like(p), torch.zeros_like(p)) for p in ps]
    for step in (1, 2, 3):
        for i, ((p, m, v), g) in enumerate(zip(single, gs)):
            h.check(lib.c3d_adam_step(h.ptr(p), h.ptr(g), h.ptr(m), h.ptr(v), p.numel(), 1e-3 * (i + 1), 0.9, 0.999, 1e-15, step, h.stream()), "adam")
        recs = [h.AdamTensor(h.ptr(p), h.ptr(g), h.ptr(m), h.ptr(v), p.numel(), step, 1e-3 * (i + 1), 0.9, 0.999, 1e-15) for i, ((p, m, v), g) in enumerate(zip(multi, gs))]
        h.check(lib.c3d_adam_step_multi((h.AdamTensor * len(recs))(*recs), len(recs), h.stream()), "adam multi")
    for (p, m, v), (q, n_, w) in zip(single, multi):
        assert torch.equal(p, q) and torch.equal(m, n_) and torch.equal(v, w)


def test_renderer_and_trainer_mirror_reduce_the_loss():
    """GaussianSplattingRenderer.render through the controller (the reference's call stack) and a few
    optimisation steps of GaussianSplatting3D: the loss must go down and parameters must stay finite."""
    from MVs_Algorithms.GaussianSplatting.main_3DGS import GSParams, GaussianSplatting3D, GaussianSplattingCameraController
    from MVs_Algorithms.GaussianSplatting.main_3DGS_renderer import GaussianSplattingRenderer
    np.random.seed(0); torch.manual_seed(0)
    H = W = 192
    poses = [[1.75, 0.0, az, 0.0, 0.0, 0.0] for az in (0.0, 90.0, 180.0, -90.0)]
    # targets: a denser, more opaque ball rendered with the same stack
    tgt = GaussianSplattingRenderer(sh_degree=3, device="cuda")
    tgt.initialize(None, num_pts=4000)
    with torch.no_grad():
        tgt.gaussians._opacity += 3.0
        tgt.gaussians._features_dc += 1.0
    tgt.gaussians.active_sh_degree = 3
    ctl = GaussianSplattingCameraController(tgt, W, H, 49.1, static_bg=[1.0, 1.0, 1.0], device="cuda")
    with torch.no_grad():
        imgs, masks, extra = ctl.render_all_pose(poses)
    assert imgs.shape == (4, 3, H, W) and masks.shape == (4, 1, H, W) and extra["radii"].shape == (4, 4000)
    assert 0.02 < masks.mean().item() < 0.9
    ref_images = [imgs[i].permute(1, 2, 0).cpu() for i in range(4)]       # ComfyUI IMAGE layout [H,W,3]
    ref_masks = [masks[i, 0].cpu() for i in range(4)]
    p = GSParams(training_iterations=30, batch_size=2, num_pts=3000, density_start_iter=10 ** 9, density_end_iter=-1, invert_bg_prob=1.0)
    tr = GaussianSplatting3D(p, None, device="cuda")
    tr.prepare_training(ref_images, ref_masks, poses, 49.1)
    losses = [tr.training_step(s, [s % 4, (s + 1) % 4]).item() for s in range(30)]
    assert np.isfinite(losses).all()
    assert np.mean(losses[-5:]) < 0.9 * np.mean(losses[:5]), losses
    for q in tr.params:
        assert torch.isfinite(q).all()


def _drop_in_pair(seed=8, n=60000, W=400, H=232, scale=0.012, cam=(-12.0, 70.0, 2.2)):
    """-> (plain(), rawp(), N, H, W): one differentiated forward + backward through the drop-in rasterizer on the plain and on the raw-parameter entry point, returning image,
    radii, depth, alpha and every gradient"""
    import diff_gaussian_rasterization as dgr
    sc = S.make_cloud(n, seed=seed, log_scale_mean=np.log(scale))
    raw = S.make_cloud(n, seed=seed, log_scale_mean=np.log(scale), activated=False)
    st = S.camera_settings(W, H, 49.1, *cam)
    rs = hip_settings(st, "cuda")
    gC = _dev(np.random.default_rng(3).normal(size=(3, H, W)).astype(np.float32), torch.float32)

    def plain():
        color, radii, depth, alpha, inp, m2d = hip_forward(sc, st, requires_grad=True)
        ((color * gC).sum() + alpha.sum() + 0.1 * depth.sum()).backward()
        return [color.detach(), radii, depth.detach(), alpha.detach(), m2d.grad] + [inp[k].grad for k in ("means3D", "opacities", "shs", "scales", "rotations")]

    def rawp():
        t = [torch.tensor(raw[k] if k != "shs" else raw["shs"][:, :1], dtype=torch.float32, device="cuda", requires_grad=True) for k in ("means3D", "shs")]
        f_rest = torch.tensor(raw["shs"][:, 1:], dtype=torch.float32, device="cuda", requires_grad=True)
        rest = [torch.tensor(raw[k], dtype=torch.float32, device="cuda", requires_grad=True) for k in ("opacities", "scales", "rotations")]
        m2d = torch.zeros_like(t[0], requires_grad=True)
        color, radii, depth, alpha = dgr.rasterize_gaussians_raw(t[0], m2d, t[1], f_rest, rest[0], rest[1], rest[2], rs)
        ((color * gC).sum() + alpha.sum() + 0.1 * depth.sum()).backward()
        return [color.detach(), radii, depth.detach(), alpha.detach(), m2d.grad, t[0].grad, t[1].grad, f_rest.grad] + [x.grad for x in rest]

    return plain, rawp, n, H, W


def _sync_reference(dgr, fn, key):
    """one run of fn on the wheel's synchronous path with nothing learnt for the shape -> (results, pair count)"""
    dgr._learnt.pop(key, None)
    was = dgr.sync_free(False)
    try:
        ref = fn()
    finally:
        dgr.sync_free(was)
    return ref, int(dgr.last_num_rendered)


@pytest.mark.parametrize("mode", ["verified", "unverified"])
def test_sync_free_drop_in_forward_equals_the_synchronous_path_bit_for_bit(mode):
    """After the first call of a (device, H, W) shape the drop-in rasterizer enqueues its whole forward at once (c3d_gs_forward_nosync / c3d_gs_forward_raw_nosync): launches
    sized for a learnt hint, buffers for four times that, the count left on the device -- and, in the default mode, forward() waits only for the count word the scan stores
    into pinned memory.  Same kernels, same arithmetic: images, radii and every gradient must have the same BITS as the synchronous path's, on both entry points."""
    import warnings
    import diff_gaussian_rasterization as dgr
    plain, rawp, N, H, W = _drop_in_pair()
    key = (torch.cuda.current_device(), H, W)
    was_mode = dgr.sync_free(mode)
    try:
        for fn in (plain, rawp):
            ref, D = _sync_reference(dgr, fn, key)
            dgr.flush()
            assert dgr._learnt[key][N] == D > 0                           # the synchronous call taught the shape its pair count
            first, cap = dgr._capacity_for(key, N)
            assert first >= int(dgr._HEADROOM * D) and cap == dgr._ROOM * first      # launches for 1.25 x the count, buffers for four times the launches
            first2, cap2 = dgr._capacity_for(key, N + N // 10)            # a model that has just densified: the estimate follows the point count, no synchronous call
            assert first2 > first and cap2 == dgr._ROOM * first2
            dgr.last_num_rendered = -1
            redone, beyond = dgr.redone_calls, dgr.beyond_hint_calls
            with warnings.catch_warnings():
                warnings.simplefilter("error")
                got = fn()
                assert dgr.pending_calls() == 1                         # the call went through the sync-free entry point: its final status words are on their way
                if mode == "verified":
                    assert int(dgr.last_num_rendered) == D              # ... and forward() has had its count already
                dgr.flush()
            assert dgr.pending_calls() == 0 and (dgr.redone_calls, dgr.beyond_hint_calls) == (redone, beyond)
            assert int(dgr.last_num_rendered) == D
            for a, b in zip(got, ref):
                assert torch.equal(a, b), fn.__name__
    finally:
        dgr.sync_free(was_mode)


@pytest.mark.parametrize("mode", ["verified", "unverified"])
def test_drop_in_forward_is_exact_when_the_learnt_capacity_is_half_the_need(mode):
    """VERDICT r5 item 1.  The reference never returns an incomplete image (main_3DGS_renderer.py:927-936 sizes its binning buffer from the exact count).  Force the learnt
    capacity -- the size of the LAUNCHES -- to half of what the view needs, under autograd and under torch.no_grad(): the sort workgroups stay and draw on, the ranges kernel
    strides, the clear follows the count.  Image, radii and ALL gradients bit-equal to sync_free(False), no warning, no exception, no second rendering; afterwards the hint
    has followed the count."""
    import warnings
    import diff_gaussian_rasterization as dgr
    plain, rawp, N, H, W = _drop_in_pair(seed=18, n=50000, W=360, H=200, scale=0.015, cam=(5.0, -40.0, 2.2))
    key = (torch.cuda.current_device(), H, W)
    slack, dgr._SLACK = dgr._SLACK, 0
    was_mode = dgr.sync_free(mode)
    try:
        for fn in (plain, rawp):
            ref, D = _sync_reference(dgr, fn, key)
            dgr._learnt[key][N] = int(D / (2 * dgr._HEADROOM))          # launch hint = half the need; buffers = 2 x the need
            first, cap = dgr._capacity_for(key, N)
            assert first <= D // 2 and D <= cap <= 2 * D
            redone, beyond = dgr.redone_calls, dgr.beyond_hint_calls
            with warnings.catch_warnings():
                warnings.simplefilter("error")
                got = fn()
                assert dgr.pending_calls() == 1
                dgr.flush()
            assert (dgr.redone_calls, dgr.beyond_hint_calls) == (redone, beyond + 1)      # the count exceeded the hint, the buffers held it: nothing rendered twice
            assert dgr._learnt[key][N] == D                         # ... and the host has learnt the count
            for a, b in zip(got, ref):
                assert torch.equal(a, b), fn.__name__
            with warnings.catch_warnings():
                warnings.simplefilter("error")
                got = fn()                                          # the next call fits its hint
                dgr.flush()
            assert (dgr.redone_calls, dgr.beyond_hint_calls) == (redone, beyond + 1)
            for a, b in zip(got, ref):
                assert torch.equal(a, b), fn.__name__
        # without autograd (parameters that require a gradient, grad mode off: not differentiated -> forward-only, same bits)
        sc = S.make_cloud(50000, seed=18, log_scale_mean=np.log(0.015))
        st = S.camera_settings(360, 200, 49.1, 5.0, -40.0, 2.2)
        D = dgr._learnt[key][N]
        dgr._learnt[key][N] = int(D / (2 * dgr._HEADROOM))
        beyond = dgr.beyond_hint_calls
        with torch.no_grad(), warnings.catch_warnings():
            warnings.simplefilter("error")
            color, radii, depth, alpha, _, _ = hip_forward(sc, st, requires_grad=True)
            assert dgr.pending_calls() == 1
            dgr.flush()
        assert dgr.beyond_hint_calls == beyond + 1 and not color.requires_grad
        was = dgr.sync_free(False)
        try:
            color0, radii0, depth0, alpha0, _, _ = hip_forward(sc, st, requires_grad=True)
        finally:
            dgr.sync_free(was)
        assert torch.equal(color, color0) and torch.equal(radii, radii0) and torch.equal(depth, depth0) and torch.equal(alpha, alpha0)
    finally:
        dgr._SLACK = slack
        dgr.sync_free(was_mode)


def test_a_view_beyond_its_buffers_is_rendered_again_at_the_exact_count():
    """the default mode's last line of defence: a pair count more than 6 x the largest seen does not fit the call's buffers.  forward() has waited for the count, so it knows --
    and renders the second half again at the exact count, on the geometry already projected, before anything leaves: image, radii and every gradient bit-equal to the
    synchronous path, under autograd and without (no caller ever receives a truncated image)."""
    import warnings
    import diff_gaussian_rasterization as dgr
    plain, rawp, N, H, W = _drop_in_pair(seed=18, n=50000, W=360, H=200, scale=0.015, cam=(5.0, -40.0, 2.2))
    key = (torch.cuda.current_device(), H, W)
    slack, dgr._SLACK = dgr._SLACK, 0
    was_mode = dgr.sync_free("verified")
    try:
        for fn in (plain, rawp):
            ref, D = _sync_reference(dgr, fn, key)
            dgr._learnt[key][N] = int(D / (2 * dgr._HEADROOM * dgr._ROOM))      # buffers for half the need
            redone = dgr.redone_calls
            with warnings.catch_warnings():
                warnings.simplefilter("error")
                got = fn()
                dgr.flush()
            assert dgr.redone_calls == redone + 1 and dgr._learnt[key][N] == D
            for a, b in zip(got, ref):
                assert torch.equal(a, b), fn.__name__
        sc = S.make_cloud(50000, seed=18, log_scale_mean=np.log(0.015))
        st = S.camera_settings(360, 200, 49.1, 5.0, -40.0, 2.2)
        dgr._learnt[key][N] = int(D / (2 * dgr._HEADROOM * dgr._ROOM))
        with torch.no_grad():
            color, radii, depth, alpha, _, _ = hip_forward(sc, st)
        assert dgr.redone_calls == redone + 2
        was = dgr.sync_free(False)
        try:
            with torch.no_grad():
                color0, radii0, depth0, alpha0, _, _ = hip_forward(sc, st)
        finally:
            dgr.sync_free(was)
        assert torch.equal(color, color0) and torch.equal(radii, radii0) and torch.equal(depth, depth0) and torch.equal(alpha, alpha0)
    finally:
        dgr._SLACK = slack
        dgr.sync_free(was_mode)


def test_unverified_mode_returns_nan_for_a_view_beyond_its_buffers_and_raises_one_call_late():
    """sync_free('unverified') -- the host never waits -- cannot render a view whose pair count exceeds its buffers.  Never an image that merely looks plausible: the planes are
    NaN, nothing is read or written out of bounds in either direction, the next examination raises, and the capacity has regrown -- the call after that is exact again."""
    import diff_gaussian_rasterization as dgr
    plain, _, N, H, W = _drop_in_pair(seed=18, n=50000, W=360, H=200, scale=0.015, cam=(5.0, -40.0, 2.2))
    key = (torch.cuda.current_device(), H, W)
    ref, D = _sync_reference(dgr, plain, key)
    slack, dgr._SLACK = dgr._SLACK, 0
    was_mode = dgr.sync_free("unverified")
    try:
        dgr._learnt[key][N] = int(D / (2 * dgr._HEADROOM * dgr._ROOM))      # buffers for half the need
        got = plain()
        torch.cuda.synchronize()
        assert torch.isnan(got[0]).all() and torch.isnan(got[2]).all() and torch.isnan(got[3]).all()
        assert torch.equal(got[1], ref[1])                            # radii do not depend on the binning
        with pytest.raises(RuntimeError, match="returned as NaN"):
            dgr.flush()
        assert dgr.pending_calls() == 0 and dgr._learnt[key][N] == D
        got = plain()
        dgr.flush()
        for a, b in zip(got, ref):
            assert torch.equal(a, b)
    finally:
        dgr._SLACK = slack
        dgr.sync_free(was_mode)


def test_a_last_call_that_could_not_be_rendered_is_reported_at_exit(tmp_path):
    """sync_free('unverified'): a caller that never issues another forward (and never calls flush()) still hears of it -- the pending status words are examined when the interpreter exits"""
    import subprocess
    import sys
    prog = tmp_path / "last_call.py"
    prog.write_text(
        "import sys, numpy as np, torch\n"
        "sys.path.insert(0, %r)\n"
        "import diff_gaussian_rasterization as dgr\n"
        "from c3d_hip import synthetic as S\n"
        "sc = S.make_cloud(50000, seed=18, log_scale_mean=np.log(0.015))\n"
        "st = S.camera_settings(360, 200, 49.1, 5.0, -40.0, 2.2)\n"
        "t = lambda a: torch.as_tensor(np.asarray(a, dtype=np.float32)).cuda()\n"
        "rs = dgr.GaussianRasterizationSettings(200, 360, st['tanfovx'], st['tanfovy'], t(st['bg']), 1.0, t(st['viewmatrix']).reshape(4, 4),\n"
        "                                       t(st['projmatrix']).reshape(4, 4), st['sh_degree'], t(st['campos']), False, False)\n"
        "def render():\n"
        "    with torch.no_grad():\n"
        "        return dgr.GaussianRasterizer(rs)(means3D=t(sc['means3D']), means2D=None, opacities=t(sc['opacities']), shs=t(sc['shs']), scales=t(sc['scales']),\n"
        "                                          rotations=t(sc['rotations']))\n"
        "render()\n"                                              # synchronous first call: learns the count
        "dgr.sync_free('unverified')\n"
        "key = (torch.cuda.current_device(), 200, 360)\n"
        "dgr._learnt[key][50000] //= 12; dgr._SLACK = 0\n"
        "render()\n"                                              # never waited for, buffers far too small, and nothing examines it
        "assert dgr.pending_calls() == 1\n"
        % os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "comfyui-3d-pack_amd"))
    r = subprocess.run([sys.executable, "-W", "always", str(prog)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    assert "RuntimeWarning" in r.stderr and "returned as NaN" in r.stderr, r.stderr[-2000:]


def test_forward_only_renders_have_the_same_bits_and_refuse_a_backward_pass():
    """C3D_GS_FLAG_FORWARD_ONLY: a render nobody will differentiate records no pair activity, runs no record-base scan and stores neither final_T nor
    n_contrib -- same image, depth, alpha and radii, bit for bit; a backward entry point that receives such settings says so instead of reading state that was never written."""
    import c3d_hip as h
    import diff_gaussian_rasterization as dgr
    lib = h.lib()
    sc = S.make_cloud(40000, seed=5, log_scale_mean=np.log(0.015))
    W, H = 333, 190
    st = S.camera_settings(W, H, 49.1, 12.0, 25.0, 2.4)
    outs = []
    for fwd_only in (False, True):
        t = {k: _dev(sc[k], torch.float32) for k in ("means3D", "opacities", "shs", "scales", "rotations")}
        rs = hip_settings(st, "cuda")
        keep = []
        cs = dgr._settings_struct(rs, keep, forward_only=fwd_only)
        N = 40000
        radii = torch.empty((N,), dtype=torch.int32, device="cuda")
        geom = torch.empty((lib.c3d_gs_geom_bytes(N),), dtype=torch.uint8, device="cuda")
        img = torch.full((lib.c3d_gs_image_bytes(H, W),), 0x5A, dtype=torch.uint8, device="cuda")
        nr = C.c_int64(0)
        h.check(lib.c3d_gs_forward_project(C.byref(cs), N, 16, h.ptr(t["means3D"]), h.ptr(t["shs"]), None, h.ptr(t["opacities"]), h.ptr(t["scales"]), h.ptr(t["rotations"]), None,
                                           h.ptr(radii), h.ptr(geom), C.byref(nr), h.stream()), "project")
        binning = torch.empty((lib.c3d_gs_binning_bytes(nr.value, H, W),), dtype=torch.uint8, device="cuda")
        color, depth, alpha = torch.empty((3, H, W), device="cuda"), torch.empty((1, H, W), device="cuda"), torch.empty((1, H, W), device="cuda")
        h.check(lib.c3d_gs_forward_render(C.byref(cs), N, 16, h.ptr(radii), h.ptr(geom), nr.value, h.ptr(binning), h.ptr(img), h.ptr(color), h.ptr(depth), h.ptr(alpha), h.stream()), "render")
        torch.cuda.synchronize()
        outs.append((color, depth, alpha, radii, img.clone()))
        if fwd_only:
            assert bool((img == 0x5A).all())                          # the per-pixel backward state was not written
            g = [torch.empty_like(t[k]) for k in ("means3D", "opacities", "shs", "scales", "rotations")]
            scratch = torch.empty((lib.c3d_gs_backward_scratch_bytes(N, nr.value),), dtype=torch.uint8, device="cuda")
            rc = lib.c3d_gs_backward(C.byref(cs), N, 16, h.ptr(t["means3D"]), h.ptr(t["shs"]), None, h.ptr(t["scales"]), h.ptr(t["rotations"]), None, h.ptr(radii), h.ptr(geom), nr.value,
                                     h.ptr(binning), h.ptr(img), h.ptr(color), None, None, h.ptr(torch.empty((N, 3), device="cuda")), h.ptr(torch.empty((N, 3), device="cuda")),
                                     h.ptr(g[1]), h.ptr(g[0]), None, h.ptr(g[2]), h.ptr(g[3]), h.ptr(g[4]), h.ptr(scratch), h.stream())
            assert rc != 0 and b"FORWARD_ONLY" in lib.c3d_last_error()
        else:
            assert not bool((img == 0x5A).all())
    for a, b in zip(outs[0][:4], outs[1][:4]):
        assert torch.equal(a, b)
    # the Python boundary: not differentiated -> forward-only; a backward pass forced onto such a call raises a clear error
    inp = {k: _dev(sc[k], torch.float32).requires_grad_() for k in ("means3D", "opacities", "shs", "scales", "rotations")}
    e = torch.empty(0, device="cuda")
    color, radii, depth, alpha = dgr._RasterizeGaussians.apply(inp["means3D"], torch.zeros_like(inp["means3D"]), inp["shs"], e, inp["opacities"], inp["scales"], inp["rotations"], e,
                                                               hip_settings(st, "cuda"), False)
    assert torch.equal(color.detach(), outs[0][0]) and torch.equal(radii, outs[0][3])
    with pytest.raises(RuntimeError, match="forward-only"):
        color.sum().backward()


def test_fused_activation_path_equals_accessor_path():
    """rasterize_gaussians_raw (exp / sigmoid / normalize / cat folded into the kernels) against the op-by-op accessor path of
    GaussianSplattingRenderer.render: same images, same gradients on the raw parameters."""
    from MVs_Algorithms.GaussianSplatting.main_3DGS_renderer import GaussianSplattingRenderer
    from shared_utils.camera_utils import MiniCam, OrbitCamera, orbit_camera
    raw = S.make_cloud(30000, seed=11, log_scale_mean=np.log(0.02), activated=False)
    W, H = 320, 200
    cam = OrbitCamera(W, H, fovy=49.1)
    mc = MiniCam(orbit_camera(-15.0, 50.0, 2.2), W, H, cam.fovy, cam.fovx, 0.01, 100, device="cuda")
    gC = _dev(np.random.default_rng(4).normal(size=(3, H, W)).astype(np.float32), torch.float32)
    outs = []
    for unfused in (False, True):
        r = GaussianSplattingRenderer(sh_degree=3, device="cuda")
        r.initialize({"xyz": raw["means3D"], "features": raw["shs"], "scaling_raw": raw["scales"], "rotation_raw": raw["rotations"] * 3.0,
                      "opacity_raw": raw["opacities"]})
        r.force_unfused = unfused
        out = r.render(mc, scaling_modifier=0.9)
        ((out["image"] * gC).sum() + out["alpha"].sum() + 0.1 * out["depth"].sum()).backward()
        g = r.gaussians
        outs.append((out, [t.grad.clone() for t in (g._xyz, g._features_dc, g._features_rest, g._opacity, g._scaling, g._rotation)],
                     out["viewspace_points"].grad.clone()))
    (o1, g1, v1), (o2, g2, v2) = outs
    assert torch.equal(o1["radii"], o2["radii"])
    assert (o1["image"] - o2["image"]).abs().mean().item() <= 1e-6 and (o1["alpha"] - o2["alpha"]).abs().max().item() <= 1e-5
    for a, b, name in zip(g1, g2, ("xyz", "f_dc", "f_rest", "opacity", "scaling", "rotation")):
        assert rel_err(a.cpu().numpy(), b.cpu().numpy()) <= 2e-4, name
    assert rel_err(v1.cpu().numpy(), v2.cpu().numpy()) <= 2e-4


@pytest.mark.parametrize("deg", [0, 1, 2])
def test_raw_parameter_paths_take_any_sh_storage_degree(deg):
    """A PLY of SH degree 0, 1 or 2 (mesh_processer/mesh_utils.py:346-350 of the reference loads any; LGM's converter writes degree 0: f_rest [N, K-1, 3] with
    K = (deg + 1)^2, empty for degree 0) takes the SAME fused paths as the degree-3 storage of the trainer -- rasterize_gaussians_raw, c3d_gs_render_views_raw,
    c3d_gs_train_views_raw -- and each is held to the op-by-op accessor path through the plain boundary (which the float64 restatement tests above cover)."""
    from MVs_Algorithms.GaussianSplatting.main_3DGS_renderer import GaussianSplattingRenderer
    from c3d_hip.gs_step import FusedViewStep
    from shared_utils.camera_utils import MiniCam, OrbitCamera, orbit_camera
    K = (deg + 1) ** 2
    raw = S.make_cloud(30000, seed=13, log_scale_mean=np.log(0.02), activated=False)
    W, H = 320, 200
    cam = OrbitCamera(W, H, fovy=49.1)
    cams = [MiniCam(orbit_camera(el, az, 2.2), W, H, cam.fovy, cam.fovx, 0.01, 100, device="cuda") for el, az in ((-15.0, 50.0), (20.0, -120.0), (45.0, 170.0))]
    rng = np.random.default_rng(4)
    gC = _dev(rng.normal(size=(3, H, W)).astype(np.float32), torch.float32)

    def make():
        r = GaussianSplattingRenderer(sh_degree=deg, device="cuda")
        r.initialize({"xyz": raw["means3D"], "features": raw["shs"][:, :K], "scaling_raw": raw["scales"], "rotation_raw": raw["rotations"], "opacity_raw": raw["opacities"]})
        g = r.gaussians
        assert g.max_sh_degree == deg and tuple(g._features_rest.shape[1:]) == (K - 1, 3)
        return r, [g._xyz, g._features_dc, g._features_rest, g._opacity, g._scaling, g._rotation]

    # (1) one view through autograd: fused raw entry point vs accessor path
    outs = []
    for unfused in (False, True):
        r, plist = make()
        r.force_unfused = unfused
        out = r.render(cams[0])
        ((out["image"] * gC).sum() + out["alpha"].sum() + 0.1 * out["depth"].sum()).backward()
        outs.append((out, [t.grad.clone() if t.grad is not None else torch.zeros_like(t) for t in plist]))
    (o1, g1), (o2, g2) = outs
    assert torch.equal(o1["radii"], o2["radii"])
    # the two paths run different instances of the projection kernel (other FMA contractions: last-bit differences in conic / opacity), so a splat that sits
    # exactly at the alpha = 1/255 threshold of a pixel may be taken by one and dropped by the other: <= 1/255 there, nothing elsewhere
    assert (o1["image"] - o2["image"]).abs().mean().item() <= 1e-6 and (o1["alpha"] - o2["alpha"]).abs().mean().item() <= 1e-6
    assert (o1["alpha"] - o2["alpha"]).abs().max().item() <= 1.05 / 255
    for a, b, name in zip(g1, g2, ("xyz", "f_dc", "f_rest", "opacity", "scaling", "rotation")):
        if a.numel():
            e = rel_err(a.cpu().numpy(), b.cpu().numpy())
            assert e <= GRAD_REL, (name, deg, e)          # a threshold flip shows in the gradients too: the north star's bound, not the 2e-4 of identical decisions
    # (2) the multi-view forward call: bit-equal to the per-view fused render
    r, plist = make()
    with torch.no_grad():
        per_view = [r.render(c) for c in cams]
        batch = r.render_views(cams, lanes=1, group=3)
    for i, pv in enumerate(per_view):
        for k in ("image", "depth", "alpha", "radii"):
            assert torch.equal(batch[k][i], pv[k]), (k, i)
    # (3) the fused training step against autograd over the accessor path, same loss
    import diff_gaussian_rasterization as dgr
    import math
    tcs = [_dev(rng.uniform(size=(3, H, W)).astype(np.float32), torch.float32) for _ in cams]
    tas = [_dev(rng.uniform(size=(1, H, W)).astype(np.float32), torch.float32) for _ in cams]
    r.force_unfused = True
    for c, tc, ta in zip(cams, tcs, tas):
        out = r.render(c)
        ((0.8 * (out["image"] - tc).abs().mean() + 3.0 * ((out["alpha"] - ta) ** 2).mean()) / len(cams)).backward()
    ref = [t.grad.clone() if t.grad is not None else torch.zeros_like(t) for t in plist]
    rs = [dgr.GaussianRasterizationSettings(H, W, math.tan(c.FoVx * 0.5), math.tan(c.FoVy * 0.5), r.bg_color, 1.0, c.world_view_transform, c.full_proj_transform,
                                            deg, c.camera_center, False, False) for c in cams]
    step = FusedViewStep(30000, H, W, "cuda", lanes=1)
    grads = [torch.empty_like(t) for t in plist]
    step.run(rs, [t.detach() for t in plist], grads, tcs, tas, None, w_l1=0.8, w_alpha_mse=3.0, scale=1.0 / len(cams), accumulate=False)
    for a, b, name in zip(grads, ref, ("xyz", "f_dc", "f_rest", "opacity", "scaling", "rotation")):
        if a.numel():
            e = rel_err(a.cpu().numpy(), b.cpu().numpy())
            assert e <= GRAD_REL, (name, deg, e)


@pytest.mark.parametrize("lanes", [1, 2, 4])
def test_fused_multi_view_step_matches_autograd(lanes):
    """c3d_gs_train_views_raw (V views, pixel loss and backward in one sync-free call, views dealt onto `lanes` HIP streams) against the
    per-view autograd path with the same loss; also the overflow / regrow path and run-to-run bit reproducibility."""
    from MVs_Algorithms.GaussianSplatting.main_3DGS_renderer import GaussianSplattingRenderer
    from c3d_hip.gs_step import FusedViewStep
    import diff_gaussian_rasterization as dgr
    raw = S.make_cloud(40000, seed=21, log_scale_mean=np.log(0.015), activated=False)
    W, H, V = 256, 160, 5          # 5 views: with lanes > 1 the per-Gaussian pass runs in two chunks (views 0-3, then 4)
    r = GaussianSplattingRenderer(sh_degree=3, device="cuda")
    r.initialize({"xyz": raw["means3D"], "features": raw["shs"], "scaling_raw": raw["scales"], "rotation_raw": raw["rotations"], "opacity_raw": raw["opacities"]})
    g = r.gaussians
    plist = [g._xyz, g._features_dc, g._features_rest, g._opacity, g._scaling, g._rotation]
    rs_list, cams = [], []
    for (el, az) in [(-20.0, 10.0), (15.0, 130.0), (40.0, -100.0), (0.0, 60.0), (-35.0, -30.0)]:
        st = S.camera_settings(W, H, 49.1, el, az, 2.2, bg=(1, 1, 1))
        rs = hip_settings(st, "cuda")
        rs_list.append(rs)
        cams.append(type("Cam", (), dict(image_height=H, image_width=W, FoVx=2 * np.arctan(st["tanfovx"]), FoVy=2 * np.arctan(st["tanfovy"]),
                                         world_view_transform=rs.viewmatrix, full_proj_transform=rs.projmatrix, camera_center=rs.campos))())
    rng = np.random.default_rng(0)
    tcs = [_dev(rng.uniform(size=(3, H, W)).astype(np.float32), torch.float32) for _ in range(V)]
    tas = [_dev(rng.uniform(size=(1, H, W)).astype(np.float32), torch.float32) for _ in range(V)]
    white = torch.ones(3, device="cuda")
    # reference: autograd over the renderer API
    total = 0.0
    for i in range(V):
        out = r.render(cams[i], bg_color=white)
        loss = 0.5 * (0.8 * (out["image"] - tcs[i]).abs().mean() + 0.3 * ((out["image"] - tcs[i]) ** 2).mean() + 3.0 * ((out["alpha"] - tas[i]) ** 2).mean())
        loss.backward()
        total += loss.item()
    ref = [p.grad.clone() for p in plist]
    for p in plist:
        p.grad = None
    # fused step, deliberately tiny capacity first to exercise the regrow path
    step = FusedViewStep(40000, H, W, "cuda", pair_capacity=5000, lanes=lanes)
    grads = [torch.zeros_like(p) for p in plist]
    lv = step.run(rs_list, [p.detach() for p in plist], grads, tcs, tas, None, w_l1=0.8, w_l2=0.3, w_alpha_mse=3.0, scale=0.5)
    assert step.capacity > 5000
    assert abs(lv.item() - total) <= 1e-4 * max(1.0, abs(total))
    for a, b, name in zip(grads, ref, ("xyz", "f_dc", "f_rest", "opacity", "scaling", "rotation")):
        assert rel_err(a.cpu().numpy(), b.cpu().numpy()) <= 5e-4, name
    # second call accumulates on top
    lv2 = step.run(rs_list, [p.detach() for p in plist], grads, tcs, tas, None, w_l1=0.8, w_l2=0.3, w_alpha_mse=3.0, scale=0.5)
    for a, b in zip(grads, ref):
        assert rel_err(a.cpu().numpy(), 2 * b.cpu().numpy()) <= 5e-4
    # fixed lane count -> fixed summation order -> identical bits from run to run
    runs = []
    for _ in range(2):
        gz = [torch.zeros_like(p) for p in plist]
        step.run(rs_list, [p.detach() for p in plist], gz, tcs, tas, None, w_l1=0.8, w_l2=0.3, w_alpha_mse=3.0, scale=0.5)
        runs.append(gz)
    for a, b in zip(*runs):
        assert torch.equal(a, b)


def test_fused_step_deferred_status():
    """defer_status: a fitted step does not wait for its own overflow / fault words; the next run() (or finish()) examines them.  Same gradients as the
    synchronous mode; an overflow between two steps is noticed one step late, loudly, and the capacity regrown."""
    import warnings
    from MVs_Algorithms.GaussianSplatting.main_3DGS_renderer import GaussianSplattingRenderer
    from c3d_hip.gs_step import FusedViewStep
    raw = S.make_cloud(30000, seed=5, log_scale_mean=np.log(0.015), activated=False)
    W, H = 192, 128
    r = GaussianSplattingRenderer(sh_degree=3, device="cuda")
    r.initialize({"xyz": raw["means3D"], "features": raw["shs"], "scaling_raw": raw["scales"], "rotation_raw": raw["rotations"], "opacity_raw": raw["opacities"]})
    g = r.gaussians
    plist = [q.detach() for q in (g._xyz, g._features_dc, g._features_rest, g._opacity, g._scaling, g._rotation)]
    rs = [hip_settings(S.camera_settings(W, H, 49.1, el, az, 2.2, bg=(1, 1, 1)), "cuda") for el, az in ((-20.0, 10.0), (15.0, 130.0), (40.0, -100.0))]
    rng = np.random.default_rng(0)
    tcs = [_dev(rng.uniform(size=(3, H, W)).astype(np.float32), torch.float32) for _ in rs]
    tas = [_dev(rng.uniform(size=(1, H, W)).astype(np.float32), torch.float32) for _ in rs]
    outs = []
    for defer in (False, True):
        step = FusedViewStep(30000, H, W, "cuda", lanes=2)
        step.defer_status = defer
        grads = [torch.empty_like(q) for q in plist]
        losses = [step.run(rs, plist, grads, tcs, tas, None, w_l1=0.8, w_alpha_mse=3.0, scale=1 / 3, accumulate=False) for _ in range(3)]
        assert (step._pending is not None) == defer               # the first run fits the capacity synchronously, later ones are deferred
        step.finish()
        assert step._pending is None
        outs.append(([l.item() for l in losses], [q.clone() for q in grads]))
    assert np.allclose(outs[0][0], outs[1][0], rtol=1e-6)            # the loss VALUE is summed with float atomics (one per workgroup): last-bit differences
    for a, b in zip(outs[0][1], outs[1][1]):
        assert torch.equal(a, b)
    # overflow noticed one step late: by the next run() (which then redoes itself with the regrown capacity) ...
    step.capacity = 2000
    step._alloc()
    with warnings.catch_warnings(record=True) as wlist:
        warnings.simplefilter("always")
        step.run(rs, plist, grads, tcs, tas, None, w_l1=0.8, w_alpha_mse=3.0, scale=1 / 3, accumulate=False)
        assert not wlist
        step.run(rs, plist, grads, tcs, tas, None, w_l1=0.8, w_alpha_mse=3.0, scale=1 / 3, accumulate=False)
    assert any("incomplete" in str(w.message) for w in wlist) and step.capacity > 2000
    step.finish()
    for a, b in zip(grads, outs[0][1]):
        assert torch.equal(a, b)


def test_fused_step_param_backward_by_gaussian_ranges():
    """run(param_chunks=K, after_chunk=...): the per-Gaussian backward pass range by range (c3d_gs_step_param_backward_range) -- what the multi-GPU
    step uses to start a range's gradient exchange underneath the next range's kernels.  Same bits as the one-launch pass; the callback sees
    every row exactly once, in order, and (here) checks that the rows it is handed are already final on the stream."""
    from MVs_Algorithms.GaussianSplatting.main_3DGS_renderer import GaussianSplattingRenderer
    from c3d_hip.gs_step import FusedViewStep
    N = 30001                                                           # not a multiple of the 256-Gaussian range granularity
    raw = S.make_cloud(N, seed=6, log_scale_mean=np.log(0.015), activated=False)
    W, H = 192, 128
    r = GaussianSplattingRenderer(sh_degree=3, device="cuda")
    r.initialize({"xyz": raw["means3D"], "features": raw["shs"], "scaling_raw": raw["scales"], "rotation_raw": raw["rotations"], "opacity_raw": raw["opacities"]})
    g = r.gaussians
    plist = [q.detach() for q in (g._xyz, g._features_dc, g._features_rest, g._opacity, g._scaling, g._rotation)]
    rs = [hip_settings(S.camera_settings(W, H, 49.1, el, az, 2.2, bg=(1, 1, 1)), "cuda") for el, az in ((-20.0, 10.0), (15.0, 130.0), (40.0, -100.0))]
    rng = np.random.default_rng(0)
    tcs = [_dev(rng.uniform(size=(3, H, W)).astype(np.float32), torch.float32) for _ in rs]
    tas = [_dev(rng.uniform(size=(1, H, W)).astype(np.float32), torch.float32) for _ in rs]
    step = FusedViewStep(N, H, W, "cuda", lanes=2)
    ref = [torch.empty_like(q) for q in plist]
    l0 = step.run(rs, plist, ref, tcs, tas, None, w_l1=0.8, w_alpha_mse=3.0, scale=1 / 3, accumulate=False)      # fits the capacity; one launch
    assert step._fitted
    for chunks in (1, 3, 7):
        grads = [torch.full_like(q, float("nan")) for q in plist]
        seen, snaps = [], []

        def after(g0, g1):
            seen.append((g0, g1))
            snaps.append([q[g0:g1].clone() for q in grads])             # stream-ordered copy: what a collective launched here would read
        l1 = step.run(rs, plist, grads, tcs, tas, None, w_l1=0.8, w_alpha_mse=3.0, scale=1 / 3, accumulate=False, param_chunks=chunks, after_chunk=after)
        assert len(seen) == chunks and seen[0][0] == 0 and seen[-1][1] == N and all(a[1] == b[0] for a, b in zip(seen[:-1], seen[1:]))
        assert all(a % 256 == 0 for a, _ in seen)
        assert torch.equal(l0, l1)
        for a, b in zip(grads, ref):
            assert torch.equal(a, b)
        for (g0, g1), snap in zip(seen, snaps):
            for a, b in zip(snap, ref):
                assert torch.equal(a, b[g0:g1])
    import c3d_hip
    lib = c3d_hip.lib()
    assert lib.c3d_gs_step_param_backward_range(None, 1, N, *([None] * 5), *([None] * 6), 100, 0, None, 2, 10, None) != 0      # argument checks come first
    assert b"NULL" in lib.c3d_last_error()


@pytest.mark.parametrize("lambda_ssim,offsets", [(0.0, False), (0.2, False), (0.2, True)])
def test_trainer_fused_step_equals_autograd_step(lambda_ssim, offsets):
    """GaussianSplatting3D.training_step with the reference's node defaults -- lambda_ssim 0.2, invert_bg_prob 0.5 (nodes.py:1177,1181) -- and
    without MS-SSIM: the fused library step and the per-view autograd path draw the same backgrounds, produce the same loss and leave the
    same parameters after an Adam step.  Masks are SOFT (rembg alpha / the bilinear resize of _fit): (c - ref) * mask, not c*mask - ref*mask^2."""
    from MVs_Algorithms.GaussianSplatting.main_3DGS import GSParams, GaussianSplatting3D
    H = W = 192                                                     # 5-scale MS-SSIM needs sides > 160
    poses = [[1.75, -10.0, az, 0.0, 0.0, 0.0] for az in (0.0, 120.0, -120.0)]
    rng = np.random.default_rng(1)
    refs = [torch.tensor(rng.uniform(size=(H, W, 3)).astype(np.float32)) for _ in poses]
    masks = [torch.tensor(np.clip(rng.uniform(size=(H, W)) * 1.6 - 0.3, 0, 1).astype(np.float32)) for _ in poses]
    results = []
    for variant in ("library", "halves", "autograd"):
        fused = variant != "autograd"
        np.random.seed(3); torch.manual_seed(3)
        p = GSParams(training_iterations=2, batch_size=3, lambda_ssim=lambda_ssim, num_pts=5000, density_start_iter=10 ** 9, density_end_iter=-1, invert_bg_prob=0.5,
                     lambda_offset=0.5 if offsets else 0.0, lambda_offset_opacity=0.3 if offsets else 0.0)
        tr = GaussianSplatting3D(p, None, device="cuda")
        tr.use_fused_step = fused
        tr.image_loss_in_torch = variant == "halves"        # library: L1 + alpha MSE + MS-SSIM inside c3d_gs_train_views_raw; halves: torch's loss between the two halves
        tr.ms_ssim_loss.use_hip = False                      # wherever torch computes the loss here it is the plain-torch MS-SSIM: an independent check of the HIP one
        tr.prepare_training(refs, masks, poses, 49.1)
        assert tr._can_fuse() == fused
        np.random.seed(17)                                           # the per-view background draws
        losses = [tr.training_step(s, [0, 1, 2]).item() for s in range(2)]
        results.append((losses, [q.detach().clone() for q in tr.params]))
    (l2, p2) = results[-1]
    for (l1, p1) in results[:-1]:
        assert abs(l1[0] - l2[0]) <= 1e-5 * max(1, abs(l2[0])) and abs(l1[1] - l2[1]) <= 1e-4 * max(1, abs(l2[1])), (l1, l2)
        for a, b in zip(p1, p2):
            assert (a - b).abs().max().item() <= 2e-4 * max(1.0, b.abs().max().item())


@pytest.mark.parametrize("shape,masked", [((2, 3, 201, 183), False), ((2, 3, 201, 183), True), ((1, 3, 1080, 1920), True), ((3, 1, 176, 320), False),
                                          ((2, 3, 186, 330), True)])      # 186 = 16 * 11 + 10, 330 = 32 * 10 + 10: the last tile of a row / column pools its whole halo (round 4: pooling inside the forward kernel)
def test_msssim_hip_matches_the_torch_restatement(shape, masked):
    """include/c3d_loss.h: value and d/dy of mean MS-SSIM(x, y) from the fused HIP kernels against the plain-torch restatement of the published
    algorithm (shared_utils/msssim.py, the stand-in for pytorch_msssim.MS_SSIM the reference calls at main_3DGS.py:192) -- odd sizes (the padded
    2 x 2 pooling), the level-0 transform x * mask, clamp(y) * mask with its chain rule, gradient accumulation, bit reproducibility."""
    import c3d_hip as h
    from shared_utils.msssim import MS_SSIM
    B, C, H, W = shape
    gen = torch.Generator(device="cpu").manual_seed(B * H + W)
    base = torch.rand((B, C, H, W), generator=gen)
    x = (0.6 * base + 0.4 * torch.rand((B, C, H, W), generator=gen)).cuda()
    y = (0.6 * base + 0.4 * torch.rand((B, C, H, W), generator=gen) * 1.3 - 0.1).cuda()      # some values outside [0, 1]: the clamp matters
    m = torch.rand((B, 1, H, W), generator=gen).cuda() if masked else None
    ms = MS_SSIM(data_range=1, size_average=True, channel=C)
    ms.use_hip = False
    yt = y.clone().requires_grad_(True)
    ref = ms(x * m, yt.clamp(0, 1) * m) if masked else ms(x, yt)
    ref.backward()
    lib = h.lib()
    ws = torch.empty((lib.c3d_msssim_workspace_bytes(B, C, H, W),), dtype=torch.uint8, device="cuda")
    grads = []
    for rep in range(2):
        g = torch.full_like(y, 7.0)
        val = torch.zeros(1, device="cuda")
        h.check(lib.c3d_msssim_value_grad(h.ptr(x), h.ptr(y), h.ptr(m) if masked else None, 1 if masked else 0, B, C, H, W, 1.0, 0, h.ptr(g), h.ptr(val), h.ptr(ws), h.stream()), "msssim")
        grads.append(g)
    assert torch.equal(grads[0], grads[1])                              # no atomics: identical bits
    assert abs(val.item() - ref.item()) <= 2e-5 * max(abs(ref.item()), 1e-3), (val.item(), ref.item())
    err = (grads[0] - yt.grad).abs()
    scale = yt.grad.abs().max().item()
    print("[msssim %s masked=%s] value %.6f (torch %.6f), grad max err %.2e of max %.2e, rel L2 %.2e"
          % (shape, masked, val.item(), ref.item(), err.max().item(), scale, (err.norm() / yt.grad.norm()).item()))
    assert (err.norm() / yt.grad.norm()).item() <= 1e-3 and err.max().item() <= 2e-3 * scale
    # accumulate with a scale
    g = torch.ones_like(y)
    h.check(lib.c3d_msssim_value_grad(h.ptr(x), h.ptr(y), h.ptr(m) if masked else None, 1 if masked else 0, B, C, H, W, -0.5, 1, h.ptr(g), None, h.ptr(ws), h.stream()), "msssim")
    assert torch.allclose(g, 1.0 - 0.5 * grads[0], rtol=1e-5, atol=1e-7 * max(scale, 1e-30) + 1e-12)
    # the autograd front end (what the trainers call): MS_SSIM(x, y) -> HIP, gradient through backward()
    ms_h = MS_SSIM(data_range=1, size_average=True, channel=C)
    yh = (y.clamp(0, 1) * m if masked else y).detach().clone().requires_grad_(True)
    v = ms_h(x * m if masked else x, yh)
    (3.0 * v).backward()
    yr = yh.detach().clone().requires_grad_(True)
    vr = ms(x * m if masked else x, yr)
    vr.backward()
    assert abs(v.item() - vr.item()) <= 2e-5 and ((yh.grad - 3.0 * yr.grad).norm() / (3.0 * yr.grad.norm())).item() <= 1e-3


def test_fused_forward_backward_halves_match_autograd():
    """c3d_gs_forward_views_raw + c3d_gs_backward_views_raw (the fused step split at the image, per-view backgrounds, caller-supplied
    dL/dcolor, dL/ddepth, dL/dalpha) against the per-view autograd path over the same renderer."""
    from MVs_Algorithms.GaussianSplatting.main_3DGS_renderer import GaussianSplattingRenderer
    from c3d_hip.gs_step import FusedViewStep
    raw = S.make_cloud(40000, seed=21, log_scale_mean=np.log(0.015), activated=False)
    W, H, V = 256, 160, 5
    r = GaussianSplattingRenderer(sh_degree=3, device="cuda")
    r.initialize({"xyz": raw["means3D"], "features": raw["shs"], "scaling_raw": raw["scales"], "rotation_raw": raw["rotations"], "opacity_raw": raw["opacities"]})
    g = r.gaussians
    plist = [g._xyz, g._features_dc, g._features_rest, g._opacity, g._scaling, g._rotation]
    rs_list, cams, bgs = [], [], []
    for i, (el, az) in enumerate([(-20.0, 10.0), (15.0, 130.0), (40.0, -100.0), (0.0, 60.0), (-35.0, -30.0)]):
        bg = (1.0, 1.0, 1.0) if i % 2 == 0 else (0.0, 0.0, 0.0)
        st = S.camera_settings(W, H, 49.1, el, az, 2.2, bg=bg)
        rs = hip_settings(st, "cuda")
        rs_list.append(rs); bgs.append(rs.bg)
        cams.append(type("Cam", (), dict(image_height=H, image_width=W, FoVx=2 * np.arctan(st["tanfovx"]), FoVy=2 * np.arctan(st["tanfovy"]),
                                         world_view_transform=rs.viewmatrix, full_proj_transform=rs.projmatrix, camera_center=rs.campos))())
    rng = np.random.default_rng(0)
    gC = _dev(rng.normal(size=(V, 3, H, W)).astype(np.float32), torch.float32)
    gD = _dev(rng.normal(size=(V, 1, H, W)).astype(np.float32), torch.float32)
    gA = _dev(rng.normal(size=(V, 1, H, W)).astype(np.float32), torch.float32)
    imgs = []
    for i in range(V):
        out = r.render(cams[i], bg_color=bgs[i])
        imgs.append((out["image"].detach().clone(), out["alpha"].detach().clone(), out["depth"].detach().clone(), out["radii"].clone()))
        # render() clamps the image; the fused halves hand out the unclamped colour, so differentiate through the same clamp on both sides
        ((out["image"] * gC[i]).sum() + (out["depth"] * gD[i]).sum() + (out["alpha"] * gA[i]).sum()).backward()
    ref = [p.grad.clone() for p in plist]
    for p in plist:
        p.grad = None
    for lanes in (1, 4):
        step = FusedViewStep(40000, H, W, "cuda", pair_capacity=5000, lanes=lanes)      # tiny capacity: the regrow path of forward()
        color, depth, alpha, radii = step.forward(rs_list, [p.detach() for p in plist], want_depth=True, want_radii=True)
        assert step.capacity > 5000
        for i in range(V):
            # the step projects all its views in one kernel (k_preprocess_views), the renderer one view per launch: the same statements, but two
            # compilations of them (float32 contraction / ordering may differ in the last bit), hence "equal to a few ulp" and not torch.equal
            for got, want in ((color[i].clamp(0, 1), imgs[i][0]), (alpha[i], imgs[i][1]), (depth[i], imgs[i][2])):
                assert torch.allclose(got, want, rtol=1e-5, atol=1e-5), float((got - want).abs().max())
            assert int((radii[i] != imgs[i][3]).sum()) <= 2
            print("[halves] view %d lanes %d: max |dcolor| %.1e, bit-equal %s" % (i, lanes, float((color[i].clamp(0, 1) - imgs[i][0]).abs().max()), torch.equal(color[i].clamp(0, 1), imgs[i][0])))
        dcolor = gC * ((color >= 0) & (color <= 1))                  # d clamp
        grads = [torch.empty_like(p) for p in plist]
        step.backward(grads, dcolor, gA, gD, accumulate=False)
        for a, b, name in zip(grads, ref, ("xyz", "f_dc", "f_rest", "opacity", "scaling", "rotation")):
            assert rel_err(a.cpu().numpy(), b.cpu().numpy()) <= 5e-4, (name, lanes)
        step.backward(grads, dcolor, gA, gD, accumulate=True)        # ... and on top
        for a, b in zip(grads, ref):
            assert rel_err(a.cpu().numpy(), 2 * b.cpu().numpy()) <= 5e-4
        rg, g2 = step.read_view(V - 1)
        assert int((rg != imgs[V - 1][3]).sum()) <= 2 and torch.isfinite(g2).all()
        # the backward half clears the "record written" bytes itself.  Stale marks -- here every byte behind the forward state is set to 0x41, as a call in
        # between on these slices could leave them: valid bytes all "written", the records 12.08 each -- would make the per-Gaussian pass add records nobody wrote this time.
        import c3d_hip as h
        lib = h.lib()
        step.backward(grads, dcolor, gA, gD, accumulate=False)
        once = [t.clone() for t in grads]
        slice_bytes = lib.c3d_gs_step_workspace_bytes(40000, H, W, step.capacity, 1)
        fwd_only = lib.c3d_gs_render_workspace_bytes(40000, H, W, step.capacity, 1)
        for v in range(V):      # a slice = forward state | dL/dcolour plane | gradient records | valid bytes | per-view hand-over arrays | loss partials
            step.workspace[v * slice_bytes + fwd_only + 12 * H * W:(v + 1) * slice_bytes].fill_(0x41)
        step.backward(grads, dcolor, gA, gD, accumulate=False)
        for a, b in zip(grads, once):
            assert torch.equal(a, b)


def test_fused_densify_statistics_equal_the_torch_form():
    """c3d_gs_step_accumulate_densify_stats (round 5: the statistics of the step's last view accumulated by one kernel straight from the workspace) against
    c3d_gs_step_read_view + GaussianModel.add_densification_stats (reference main_3DGS.py:210-213, main_3DGS_renderer.py:767-769): same denominators and radii exactly, the
    gradient norms to rounding (sqrt(x^2 + y^2) in one kernel against torch's norm reduction)"""
    from MVs_Algorithms.GaussianSplatting.main_3DGS_renderer import GaussianSplattingRenderer
    from c3d_hip.gs_step import FusedViewStep
    raw = S.make_cloud(30000, seed=5, log_scale_mean=np.log(0.02), activated=False)
    W, H, V = 200, 136, 3
    r = GaussianSplattingRenderer(sh_degree=3, device="cuda")
    r.initialize({"xyz": raw["means3D"], "features": raw["shs"], "scaling_raw": raw["scales"], "rotation_raw": raw["rotations"], "opacity_raw": raw["opacities"]})
    g = r.gaussians
    plist = [g._xyz, g._features_dc, g._features_rest, g._opacity, g._scaling, g._rotation]
    rs = [hip_settings(S.camera_settings(W, H, 49.1, el, az, 2.2, bg=(1, 1, 1)), "cuda") for el, az in ((-10.0, 20.0), (25.0, 140.0), (5.0, -80.0))]
    rng = np.random.default_rng(1)
    tc = [_dev(rng.uniform(size=(3, H, W)).astype(np.float32), torch.float32) for _ in range(V)]
    ta = [_dev(rng.uniform(size=(1, H, W)).astype(np.float32), torch.float32) for _ in range(V)]
    step = FusedViewStep(30000, H, W, "cuda", views=V)
    grads = [torch.empty_like(p) for p in plist]
    step.run(rs, [p.detach() for p in plist], grads, tc, ta, None, w_l1=0.8, w_alpha_mse=3.0, scale=1.0 / V, accumulate=False)
    for view in (V - 1, 0):
        acc0 = _dev(rng.uniform(size=(30000, 1)).astype(np.float32), torch.float32)
        den0 = _dev(rng.integers(0, 5, size=(30000, 1)).astype(np.float32), torch.float32)
        mr0 = _dev(rng.integers(0, 30, size=(30000,)).astype(np.float32), torch.float32)
        radii, vg = step.read_view(view)
        assert int((radii > 0).sum()) > 10000 and float(vg.abs().max()) > 0
        g.xyz_gradient_accum, g.denom, g.max_radii2D = acc0.clone(), den0.clone(), mr0.clone()
        g.add_densification_stats(vg, radii > 0, radii)
        a, d, m = acc0.clone(), den0.clone(), mr0.clone()
        step.accumulate_densify_stats(view, a, d, m)
        assert torch.equal(d, g.denom) and torch.equal(m, g.max_radii2D)
        assert torch.allclose(a, g.xyz_gradient_accum, rtol=1e-6, atol=0.0)
        assert not torch.equal(a, acc0)


def test_trainer_densify_prune_schedule():
    """The reference's default schedule densifies (main_3DGS.py:209-224): statistics from the step's last view, clone/split/prune at the
    interval, opacity reset -- through the fused step (statistics read back with c3d_gs_step_read_view) and through the autograd path.
    N changes mid-training; buffers, optimizer state and the fused workspace follow."""
    from MVs_Algorithms.GaussianSplatting.main_3DGS import GSParams, GaussianSplatting3D
    H = W = 160
    poses = [[1.75, -10.0, az, 0.0, 0.0, 0.0] for az in (0.0, 120.0, -120.0)]
    rng = np.random.default_rng(4)
    refs = [torch.tensor(rng.uniform(size=(H, W, 3)).astype(np.float32)) for _ in poses]
    masks = [torch.tensor((rng.uniform(size=(H, W)) > 0.4).astype(np.float32)) for _ in poses]
    counts = []
    for fused in (True, False):
        np.random.seed(3); torch.manual_seed(3)
        p = GSParams(training_iterations=7, batch_size=3, lambda_ssim=0.0, num_pts=4000, density_start_iter=1, density_end_iter=100,
                     densification_interval=2, opacity_reset_interval=4, densify_grad_threshold=2e-6, invert_bg_prob=1.0)
        tr = GaussianSplatting3D(p, None, device="cuda")
        tr.use_fused_step = fused
        tr.prepare_training(refs, masks, poses, 49.1)
        ns, losses = [], []
        for s in range(7):
            losses.append(tr.training_step(s, [0, 1, 2]).item())
            g = tr.renderer.gaussians
            ns.append(g._xyz.shape[0])
            n = ns[-1]
            for q in (g._features_dc, g._features_rest, g._opacity, g._scaling, g._rotation, g.xyz_gradient_accum, g.denom, g.max_radii2D, g.init_xyz):
                assert q.shape[0] == n
            for grp in tr.optimizer.param_groups:
                st = tr.optimizer.state[grp["params"][0]]
                assert st["exp_avg"].shape == grp["params"][0].shape
        assert all(np.isfinite(losses))
        assert ns[0] == 4000 and ns[1] == 4000                  # step 0 outside the window, step 1 collects statistics only
        assert ns[2] != 4000 and tr.last_densify["cloned"] + tr.last_densify["split"] > 0
        assert float(g.get_opacity.detach().max()) <= 0.6                # step 4 reset the opacities to <= 0.01; two Adam steps since
        out = tr.renderer.render(_minicam_for(tr, poses[0]), bg_color=torch.ones(3, device="cuda"))
        assert out["image"].shape == (3, H, W) and torch.isfinite(out["image"]).all()
        counts.append(ns)
    # same schedule, same random draws, gradients equal to ~1e-4: the two paths grow alike
    for a, b in zip(*counts):
        assert abs(a - b) <= 0.02 * b


def test_device_densify_equals_the_torch_form():
    """include/c3d_densify.h (classification + single-pass prefix sums + one source-index kernel + ONE gather launch for 6 parameters, 12 moments and
    init_xyz) against GaussianModel.densify_and_prune's torch form -- itself held to a run of the reference's densify_and_prune by
    tests/test_ref_conventions.py -- on the same model state and the same normal samples: same points in the same order, same values bit for bit,
    same Adam moments, same counts; with and without the scale criterion, and when nothing / everything is hot."""
    import copy
    from MVs_Algorithms.GaussianSplatting.main_3DGS_renderer import GaussianModel
    from MVs_Algorithms.GaussianSplatting.main_3DGS import GSParams
    rng = np.random.default_rng(12)
    N = 20011
    raw = S.make_cloud(N, seed=5, log_scale_mean=np.log(0.02), activated=False)

    def build():
        m = GaussianModel(3, device="cuda")
        m.create_from_tensors(torch.tensor(raw["means3D"]), torch.tensor(raw["shs"]), torch.tensor(raw["scales"]), torch.tensor(raw["rotations"]), torch.tensor(raw["opacities"]))
        m.training_setup(GSParams())
        for grp in m.optimizer.param_groups:            # moments as after a few Adam steps
            q = grp["params"][0]
            g_ = torch.Generator(device="cpu").manual_seed(len(grp["name"]))
            m.optimizer.state[q] = {"step": 3, "exp_avg": torch.randn(q.shape, generator=g_).cuda(), "exp_avg_sq": torch.rand(q.shape, generator=g_).cuda()}
        m.init_xyz = m._xyz.detach().clone()
        return m

    grad = torch.tensor(rng.uniform(0, 4e-4, size=(N, 1)).astype(np.float32)).cuda()
    den = torch.tensor(rng.integers(0, 3, size=(N, 1)).astype(np.float32)).cuda()          # zeros: 0 / 0 -> NaN -> 0
    for max_grad, max_screen in ((2e-4, 1), (2e-4, 0), (1.0, 1), (0.0, 1)):
        res = []
        noise = None
        for device_path in (False, True):
            m = build()
            m.use_device_densify = device_path
            m.xyz_gradient_accum, m.denom = grad.clone(), den.clone()
            if noise is None:
                g = m.xyz_gradient_accum / m.denom
                g[g.isnan()] = 0
                hot = g.norm(dim=-1) >= max_grad
                n_split = int((hot & ~(m.get_scaling.detach().max(dim=1).values <= m.percent_dense * 4)).sum())
                noise = torch.randn((2 * n_split, 3), generator=torch.Generator(device="cpu").manual_seed(1)).cuda()
            info = m.densify_and_prune(max_grad, min_opacity=0.3, extent=4, max_screen_size=max_screen, noise=noise.clone())
            res.append((info, m))
        (ia, a), (ib, b) = res
        assert ia == ib, (ia, ib)
        assert ib["points"] == b._xyz.shape[0] and (max_grad >= 1.0 or ib["cloned"] + ib["split"] > 0)
        for qa, qb in zip((a._xyz, a._features_dc, a._features_rest, a._opacity, a._scaling, a._rotation, a.init_xyz, a.xyz_gradient_accum, a.denom, a.max_radii2D),
                          (b._xyz, b._features_dc, b._features_rest, b._opacity, b._scaling, b._rotation, b.init_xyz, b.xyz_gradient_accum, b.denom, b.max_radii2D)):
            assert qa.shape == qb.shape and torch.equal(qa.detach(), qb.detach())
        for ga, gb in zip(a.optimizer.param_groups, b.optimizer.param_groups):
            sa, sb = a.optimizer.state[ga["params"][0]], b.optimizer.state[gb["params"][0]]
            assert sa["step"] == sb["step"] and torch.equal(sa["exp_avg"], sb["exp_avg"]) and torch.equal(sa["exp_avg_sq"], sb["exp_avg_sq"])


def _minicam_for(tr, pose):
    from shared_utils.camera_utils import MiniCam, orbit_camera
    ctl = tr.cam_controller
    radius, elev, azim, cx, cy, cz = pose
    return MiniCam(orbit_camera(elev, azim, radius, target=np.array([cx, cy, cz], dtype=np.float32)), tr.ref_size_W, tr.ref_size_H, ctl.cam.fovy, ctl.cam.fovx,
                   ctl.cam.near, ctl.cam.far, ctl.projection_matrix, device=tr.device)


def test_reduce_ranks_fixed_order_sum():
    """c3d_reduce_ranks_f32: the local half of the all-gather gradient exchange -- rank-ordered sum of the gathered copies in one pass,
    bit-equal to adding the copies one after another."""
    import c3d_hip as h
    world, n = 8, 4 * 50001
    g = torch.randn(world * n, device="cuda")
    dst = torch.empty(n, device="cuda")
    h.check(h.lib().c3d_reduce_ranks_f32(h.ptr(dst), h.ptr(g), world, n, 0.125, h.stream(dst.device)), "c3d_reduce_ranks_f32")
    ref = g[:n].clone()
    for r in range(1, world):
        ref += g[r * n:(r + 1) * n]
    assert torch.equal(dst, ref * 0.125)
    # FlatGrads: views alias one 16-byte aligned buffer in parameter order
    from c3d_hip.parallel import FlatGrads
    ps = [torch.zeros(1001, *s, device="cuda") for s in ((3,), (1, 3), (15, 3), (1,), (3,), (4,))]
    fg = FlatGrads(ps)
    for i, v in enumerate(fg.views):
        assert v.shape == ps[i].shape and v.is_contiguous() and v.data_ptr() % 16 == 0
        v.fill_(float(i + 1))
    assert float(fg.flat.sum()) == sum((i + 1) * p.numel() for i, p in enumerate(ps))


def test_views_of_very_unequal_size_in_one_launch_equal_their_per_view_renders():
    """the workgroups of a multi-view sort pass serve view (linear block id) % V and, once their view has no tile left, draw tickets of views that have (k_onesweep):
    eight views whose pair counts differ by more than an order of magnitude -- close-ups next to far shots -- through ONE launch per stage, bit-identical to
    the per-view path"""
    from MVs_Algorithms.GaussianSplatting.main_3DGS_renderer import GaussianSplattingRenderer
    from MVs_Algorithms.GaussianSplatting.main_3DGS import GaussianSplattingCameraController
    from shared_utils.camera_utils import MiniCam, orbit_camera
    raw = S.make_cloud(120000, seed=21, log_scale_mean=np.log(0.012), activated=False)
    r = GaussianSplattingRenderer(sh_degree=3, device="cuda")
    r.initialize({"xyz": raw["means3D"], "features": raw["shs"], "scaling_raw": raw["scales"], "rotation_raw": raw["rotations"], "opacity_raw": raw["opacities"]})
    W, H = 400, 304
    ctl = GaussianSplattingCameraController(r, W, H, 49.1, static_bg=[0.1, 0.1, 0.1])
    poses = [[rad, el, az, 0.0, 0.0, 0.0] for rad, el, az in ((1.25, 0.0, 0.0), (9.0, 10.0, 40.0), (1.3, -20.0, 90.0), (7.0, 30.0, 130.0), (12.0, 0.0, 180.0), (1.4, 45.0, -140.0),
                                                            (5.0, -35.0, -90.0), (2.5, 5.0, -30.0))]
    import diff_gaussian_rasterization as dgr
    counts = []
    with torch.no_grad():
        per_view = []
        for p in poses:
            per_view.append(ctl.render_at_pose(p))
            dgr.flush()
            counts.append(int(dgr.last_num_rendered))
        cams = [MiniCam(orbit_camera(el, az, rad, target=np.zeros(3, dtype=np.float32)), W, H, ctl.cam.fovy, ctl.cam.fovx, ctl.cam.near, ctl.cam.far, ctl.projection_matrix, device="cuda")
                for rad, el, az, *_ in poses]
        out = r.render_views(cams, ctl.static_bg, lanes=1, group=8)
    assert max(counts) > 2.5 * min(counts) and min(counts) > 0, counts
    for i, pv in enumerate(per_view):
        for k in ("image", "depth", "alpha"):
            assert torch.equal(out[k][i], pv[k]), (k, i, counts)
        assert torch.equal(out["radii"][i], pv["radii"])


@pytest.mark.parametrize("lanes,group,res", [(1, 8, (200, 136)), (2, 2, (200, 136)), (3, 1, (200, 136)), (2, 3, (251, 143))])
def test_render_views_equals_per_view_render(lanes, group, res):
    """c3d_gs_render_views_raw (a whole orbit in one call: `group` views per launch of every stage of the chain, `lanes` groups in flight) gives exactly what
    the per-view render() gives -- one group of all five views, groups of 2 + 2 + 1 on two streams, single views on three, 3 + 2; the camera controller takes
    that path when autograd is off."""
    from MVs_Algorithms.GaussianSplatting.main_3DGS_renderer import GaussianSplattingRenderer
    from MVs_Algorithms.GaussianSplatting.main_3DGS import GaussianSplattingCameraController
    raw = S.make_cloud(30000, seed=8, log_scale_mean=np.log(0.02), activated=False)
    r = GaussianSplattingRenderer(sh_degree=3, device="cuda")
    r.initialize({"xyz": raw["means3D"], "features": raw["shs"], "scaling_raw": raw["scales"], "rotation_raw": raw["rotations"], "opacity_raw": raw["opacities"]})
    W, H = res                     # (251, 143): odd tile grid (16 x 9 tiles, partial tiles on both borders, odd supertile counts)
    ctl = GaussianSplattingCameraController(r, W, H, 49.1, static_bg=[0.2, 0.5, 0.9])
    poses = [[2.2, el, az, 0.0, 0.0, 0.0] for el, az in ((-30.0, 0.0), (0.0, 75.0), (30.0, 150.0), (60.0, -120.0), (10.0, -45.0))]
    with torch.no_grad():
        per_view = [ctl.render_at_pose(p) for p in poses]
        r._view_render_key = None
        cams = []
        from shared_utils.camera_utils import MiniCam, orbit_camera
        for radius, el, az, cx, cy, cz in poses:
            cams.append(MiniCam(orbit_camera(el, az, radius, target=np.array([cx, cy, cz], dtype=np.float32)), W, H, ctl.cam.fovy, ctl.cam.fovx, ctl.cam.near, ctl.cam.far,
                                ctl.projection_matrix, device="cuda"))
        out = r.render_views(cams, ctl.static_bg, lanes=lanes, group=group)
        images, masks, extra = ctl.render_all_pose(poses)            # autograd off -> batched path
    for i, pv in enumerate(per_view):
        for k in ("image", "depth", "alpha"):
            assert torch.equal(out[k][i], pv[k]), (k, i)
        assert torch.equal(out["radii"][i], pv["radii"]) and torch.equal(out["visibility_filter"][i], pv["visibility_filter"])
        assert torch.equal(images[i], pv["image"]) and torch.equal(masks[i], pv["alpha"])
    assert float(out["alpha"].max()) > 0.5
    # with autograd on the controller falls back to the per-view loop and keeps the graph
    images2, _, extra2 = ctl.render_all_pose(poses[:2])
    assert images2.requires_grad and "viewspace_points" in extra2
    # the workspace's size picks the schedule: with fewer slices than lanes * group the groups get narrower (same images); less than one slice is refused
    with torch.no_grad():
        r.render_views(cams, ctl.static_bg, lanes=lanes, group=group)              # (the controller call above left a default object behind: make this one current)
    vr = r._view_render
    assert vr.lanes == lanes and vr.group == group and vr._fitted
    full = vr.workspace
    one = vr.workspace.numel() // vr.slices
    with torch.no_grad():
        for keep in sorted({1, max(1, vr.slices - 1)}):
            vr.workspace = full[:one * keep]
            out1 = r.render_views(cams, ctl.static_bg, lanes=lanes, group=group)
            for k in ("image", "depth", "alpha"):
                assert torch.equal(out1[k], out[k]), (k, keep)
        vr.workspace = full[:one - 256]
        with pytest.raises(RuntimeError, match="less than one slice"):
            r.render_views(cams, ctl.static_bg, lanes=lanes, group=group)
        vr.workspace = full


@pytest.mark.parametrize("lambda_ssim", [0.0, 0.2])
def test_trainer_longer_run_with_densification(lambda_ssim):
    """60 steps of the trainer at 100k points / 256^2 with the densification schedule running (fused step: in-kernel loss for lambda_ssim = 0,
    forward / torch MS-SSIM / backward halves otherwise): the loss goes down, N changes several times, nothing goes non-finite, buffers keep following N."""
    from MVs_Algorithms.GaussianSplatting.main_3DGS import GSParams, GaussianSplatting3D
    H = W = 256
    rng = np.random.default_rng(11)
    poses = [[1.75, float(el), float(az), 0.0, 0.0, 0.0] for el in (-20, 20) for az in (0, 90, 180, 270)]
    yy, xx = np.meshgrid(np.linspace(-1, 1, H), np.linspace(-1, 1, W), indexing="ij")
    disk = ((xx ** 2 + yy ** 2) < 0.45).astype(np.float32)
    refs = [torch.tensor(np.stack([disk * (0.3 + 0.1 * i), disk * 0.5, disk * (0.9 - 0.1 * i)], -1).astype(np.float32)) for i in range(len(poses))]
    masks = [torch.tensor(disk) for _ in poses]
    np.random.seed(5); torch.manual_seed(5)
    p = GSParams(training_iterations=60, batch_size=4, lambda_ssim=lambda_ssim, num_pts=100000, density_start_iter=5, density_end_iter=55,
                 densification_interval=10, opacity_reset_interval=30, densify_grad_threshold=5e-7, invert_bg_prob=1.0)
    tr = GaussianSplatting3D(p, None, device="cuda")
    tr.prepare_training(refs, masks, poses, 49.1)
    assert tr._can_fuse()                                            # round 2: also with MS-SSIM
    rs = np.random.RandomState(0)
    losses, ns = [], []
    for s in range(60):
        losses.append(tr.training_step(s, [int(i) for i in rs.randint(0, len(poses), 4)]).item())
        ns.append(tr.renderer.gaussians._xyz.shape[0])
    assert np.isfinite(losses).all()
    assert np.mean(losses[-10:]) < np.mean(losses[:10])
    assert len(set(ns)) >= 3 and ns[-1] != 100000
    g = tr.renderer.gaussians
    for t in (g._xyz, g._features_dc, g._features_rest, g._opacity, g._scaling, g._rotation):
        assert torch.isfinite(t).all() and t.shape[0] == ns[-1]


@pytest.mark.parametrize("streams,V", [(4, 33), (2, 19), (8, 7)])
def test_render_views_on_several_streams_equals_the_one_stream_call(streams, V):
    """FusedViewRender(streams=S): the views in S contiguous parts (>= 8 views each), one library call per part on a HIP stream of its own, forked from and joined into the
    caller's stream -- bit for bit the one-stream call, radii included, through the first call's capacity fit and through a forced overflow (all parts regrow and go again)."""
    from c3d_hip.gs_step import FusedViewRender
    import diff_gaussian_rasterization as dgr
    raw = S.make_cloud(40000, seed=11, log_scale_mean=np.log(0.02), activated=False)
    W, H = 320, 200
    t = lambda x: torch.as_tensor(np.asarray(x, dtype=np.float32)).cuda()
    sh = t(raw["shs"])
    params = [t(raw["means3D"]), sh[:, :1].contiguous(), sh[:, 1:].contiguous(), t(raw["opacities"]), t(raw["scales"]), t(raw["rotations"])]
    rs = []
    for i in range(V):
        st = S.camera_settings(W, H, 49.1, -30.0 + 7.0 * i, 20.0 * i, 2.0 + 0.05 * i, bg=(0.1 * (i % 5), 0.3, 0.8), sh_degree=3)
        rs.append(dgr.GaussianRasterizationSettings(H, W, st["tanfovx"], st["tanfovy"], t(st["bg"]), 1.0, t(st["viewmatrix"]).reshape(4, 4), t(st["projmatrix"]).reshape(4, 4), 3,
                                                    t(st["campos"]), False, False))
    one = FusedViewRender(40000, H, W, "cuda", group=4)
    many = FusedViewRender(40000, H, W, "cuda", group=4, streams=streams)
    S_, per = many._plan(V)
    assert S_ == max(1, min(streams, V // 8)) and per * S_ >= V
    with torch.no_grad():
        ref = one.run(rs, params, want_radii=True)
        for attempt in range(2):      # the first call fits the capacity (and rebuilds the parts), the second runs on the fitted buffers
            out = many.run(rs, params, want_radii=True)
            for a, b in zip(out, ref):
                assert torch.equal(a, b)
        assert many._fitted and (S_ == 1 or len(many._parts) == S_)
        need = many.capacity
        many.capacity = need // 4      # too small: every part reports the overflow, the call regrows all of them and renders again
        many._realloc()
        out = many.run(rs, params, want_radii=True)
        for a, b in zip(out, ref):
            assert torch.equal(a, b)
        assert many.capacity > need // 4
        # work queued on the caller's stream before and after the call is ordered with it
        x = torch.zeros(8, device="cuda")
        p2 = [q.clone() for q in params]
        p2[0].mul_(1.0)      # (a write the parts must wait for)
        out2 = many.run(rs, p2, want_radii=False)
        y = out2[0].sum() + x.sum()
        assert torch.isfinite(y) and torch.equal(out2[0], ref[0])


def test_orbit_of_seventeen_poses_through_the_controller_takes_two_streams_and_equals_the_per_view_renders():
    """The product default of the orbit path: GaussianSplattingCameraController.render_all_pose without autograd -> render_views(streams=4) -> at 17 cameras two parts
    (9 + 8 views) on two HIP streams; every image, mask and radius equals the per-view render() of the same pose bit for bit."""
    from MVs_Algorithms.GaussianSplatting.main_3DGS_renderer import GaussianSplattingRenderer
    from MVs_Algorithms.GaussianSplatting.main_3DGS import GaussianSplattingCameraController
    raw = S.make_cloud(20000, seed=21, log_scale_mean=np.log(0.025), activated=False)
    r = GaussianSplattingRenderer(sh_degree=3, device="cuda")
    r.initialize({"xyz": raw["means3D"], "features": raw["shs"], "scaling_raw": raw["scales"], "rotation_raw": raw["rotations"], "opacity_raw": raw["opacities"]})
    ctl = GaussianSplattingCameraController(r, 240, 160, 49.1, static_bg=[0.1, 0.2, 0.3])
    poses = [[2.0 + 0.03 * i, -40.0 + 6.0 * i, 21.0 * i, 0.0, 0.0, 0.0] for i in range(17)]
    with torch.no_grad():
        images, masks, extra = ctl.render_all_pose(poses)
        vr = r._view_render
        assert vr.streams == 4 and vr._plan(len(poses)) == (2, 9)
        for i, p in enumerate(poses):
            pv = ctl.render_at_pose(p)
            assert torch.equal(images[i], pv["image"]) and torch.equal(masks[i], pv["alpha"]) and torch.equal(extra["radii"][i], pv["radii"]), i
        images2, _, _ = ctl.render_all_pose(poses)      # fitted buffers, same parts
        assert torch.equal(images2, images) and len(vr._parts) == 2
