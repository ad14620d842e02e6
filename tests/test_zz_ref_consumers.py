"""The five OTHER consumers of the two rasterizers, from runs of the reference's own files.

tests/golden/ref_consumers.npz (tests/golden/make_golden_ref_consumers.py, which imports the reference's modules in the build container) holds, for
  flex     FlexiCubesRenderer.get_orbit_camera + render_mesh       /root/reference/MVs_Algorithms/FlexiCubes/flexicubes_renderer.py:28-74
  bake     color_func_to_albedo                                    /root/reference/mesh_processer/mesh_utils.py:521-568
  lgm      LGM GaussianRenderer.render                             /root/reference/Gen_3D_Modules/LGM/core/gs.py:26-97
  tgs      TGS GS3DRenderer.forward_single_view (SH and rgb heads) /root/reference/Gen_3D_Modules/TriplaneGaussian/models/renderer.py:203-306
  trellis  TRELLIS render()                                        /root/reference/Gen_3D_Modules/TRELLIS/trellis/renderers/gaussian_render.py:50-144
every call the function made into `nvdiffrast.torch` / `diff_gaussian_rasterization` (inputs, keyword arguments, outputs of the CPU oracle behind the name) and
what the function returned.  The GPU tests replay the recorded calls through the HIP drop-ins -- same tensors, same argument patterns as the real call sites --
hold every op output to the recorded one, and rebuild the function's result from the HIP outputs (the few torch lines between the ops and the return statement
are quoted with their reference line numbers).  Those replays are OP-LEVEL: an op that takes a `rast` is fed the recorded one, so a disagreement of the HIP
rasterizer could not propagate.  Round 5 adds END-TO-END replays of the two mesh consumers (`*_end_to_end_on_hip`): the HIP rasterizer's own `rast` flows into the
interpolate / antialias calls behind it exactly as in the reference's function bodies -- ids first (all but <= 1e-5 of the pixels), then the functions' results.
The Gaussian consumers make one rasterizer call per image: their replays are end-to-end as they stand.  The CPU test re-runs the generator in --check mode where the
reference is mounted."""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = os.path.join(HERE, "golden", "ref_consumers.npz")
IMG_L1 = 1e-4


def _z():
    return np.load(GOLD)


def test_consumer_fixture_is_what_the_reference_code_produces():
    """--check: regenerate by running the reference's five files in this container and compare bit for bit (skipped where /root/reference is not mounted);
    everywhere: the recorded op sequences are the ones the call sites are documented to issue"""
    z = _z()
    seq = lambda p: [str(z["%s_c%d_op" % (p, i)]) for i in range(int(z[p + "_ncalls"]))]
    assert seq("flex") == ["rasterize", "antialias", "interpolate", "interpolate", "interpolate", "antialias", "rasterize", "antialias", "interpolate", "antialias"]
    assert seq("bake") == ["rasterize", "interpolate", "interpolate"]
    assert seq("lgm") == ["gs"] * 4 and seq("tgs_sh") == ["gs"] * 2 and seq("tgs_rgb") == ["gs"] * 2 and seq("trellis") == ["gs"] * 3
    assert int(z["lgm_c0_st_sh_degree"]) == 0 and "lgm_c0_in_colors_precomp" in z.files and "lgm_c0_in_shs" not in z.files
    assert "tgs_sh_c0_in_shs" in z.files and "tgs_sh_c1_in_colors_precomp" in z.files          # the mask pass renders ones over black
    assert "trellis_c2_in_colors_precomp" in z.files                                             # convert_SHs_python: colours evaluated by the reference's eval_sh
    if not os.path.isdir("/root/reference"):
        pytest.skip("/root/reference not mounted: the committed fixture stands")
    r = subprocess.run([sys.executable, os.path.join(HERE, "golden", "make_golden_ref_consumers.py"), "--check"], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]


# ------------------------------------------------------------------------------------------------------------------------------ GPU replays
def T(a, dtype=torch.float32):
    return torch.tensor(np.asarray(a), dtype=dtype, device="cuda")


def _need_gpu():
    if not torch.cuda.is_available():
        pytest.fail("-m gpu tests need a HIP device (no CPU fallback exists)")
    import c3d_hip
    c3d_hip.lib()


def replay_dr(z, p, i, rast_override=None):
    """call i of section p through the HIP `nvdiffrast.torch`; -> list of output tensors.  Ops that take a `rast` are fed the RECORDED one (same winners)."""
    import nvdiffrast.torch as dr
    op = str(z["%s_c%d_op" % (p, i)])
    a = lambda j: z["%s_c%d_in%d" % (p, i, j)]
    if op == "rasterize":
        res = tuple(int(x) for x in a(2))
        out = dr.rasterize(dr.RasterizeCudaContext(), T(a(0)), T(a(1), torch.int32), res)
        orast, odb = z["%s_c%d_out0" % (p, i)], z["%s_c%d_out1" % (p, i)]
        r = out[0].cpu().numpy()
        same = r[..., 3] == orast[..., 3]
        assert (~same).sum() <= max(2, int(5e-4 * same.size)), (p, i, int((~same).sum()))
        assert np.abs(r[same][:, :3] - orast[same][:, :3]).mean() <= 1e-5 and np.abs(r[same][:, :3] - orast[same][:, :3]).max() <= 1e-3
        assert np.abs(out[1].cpu().numpy()[same] - odb[same]).max() <= 1e-3 * max(1.0, float(np.abs(odb).max()))
        return list(out)
    if op == "interpolate":
        kw = {}
        k_db = "%s_c%d_kw_rast_db" % (p, i)
        if k_db in z.files and z[k_db].size:
            kw["rast_db"] = T(z[k_db])
        k_da = "%s_c%d_kw_diff_attrs" % (p, i)
        if k_da in z.files and z[k_da].size:
            kw["diff_attrs"] = str(z[k_da]) if z[k_da].dtype.kind in "US" else [int(x) for x in np.atleast_1d(z[k_da])]
        out, da = dr.interpolate(T(a(0)), T(a(1)), T(a(2), torch.int32), **kw)
        assert np.abs(out.cpu().numpy() - z["%s_c%d_out0" % (p, i)]).max() <= 2e-5 * max(1.0, float(np.abs(a(0)).max())), (p, i)
        return [out, da]
    if op == "antialias":
        out = dr.antialias(T(a(0)), T(a(1)), T(a(2)), T(a(3), torch.int32))
        assert np.abs(out.cpu().numpy() - z["%s_c%d_out0" % (p, i)]).mean() <= IMG_L1, (p, i)
        assert np.abs(out.cpu().numpy() - z["%s_c%d_out0" % (p, i)]).max() <= 2e-3, (p, i)
        return [out]
    raise AssertionError("unexpected op %r" % op)


def replay_gs(z, p, i, autocast=False):
    """rasterizer call i of section p through the HIP `diff_gaussian_rasterization`, exactly as the call site builds it; -> (color, radii, depth, alpha)"""
    import diff_gaussian_rasterization as dgr
    st = lambda k: z["%s_c%d_st_%s" % (p, i, k)]
    opt = lambda k: T(z["%s_c%d_in_%s" % (p, i, k)]) if ("%s_c%d_in_%s" % (p, i, k)) in z.files else None
    rs = dgr.GaussianRasterizationSettings(image_height=int(st("image_height")), image_width=int(st("image_width")), tanfovx=float(st("tanfovx")), tanfovy=float(st("tanfovy")),
                                           bg=T(st("bg")), scale_modifier=float(st("scale_modifier")), viewmatrix=T(st("viewmatrix")), projmatrix=T(st("projmatrix")),
                                           sh_degree=int(st("sh_degree")), campos=T(st("campos")), prefiltered=bool(st("prefiltered")), debug=bool(st("debug")))
    means3D = opt("means3D")
    call = lambda: dgr.GaussianRasterizer(raster_settings=rs)(means3D=means3D, means2D=torch.zeros_like(means3D), shs=opt("shs"), colors_precomp=opt("colors_precomp"),
                                                               opacities=opt("opacities"), scales=opt("scales"), rotations=opt("rotations"), cov3D_precomp=opt("cov3D_precomp"))
    if autocast:       # TriplaneGaussian/models/renderer.py:261,295 wrap the call in torch.autocast(device_type, dtype=torch.float32)
        with torch.autocast(device_type="cuda", dtype=torch.float32):
            color, radii, depth, alpha = call()
    else:
        color, radii, depth, alpha = call()
    o = [z["%s_c%d_out%d" % (p, i, j)] for j in range(4)]
    assert np.abs(color.cpu().numpy() - o[0]).mean() <= IMG_L1, (p, i)
    assert (radii.cpu().numpy() != o[1]).sum() <= 2, (p, i)
    assert np.abs(alpha.cpu().numpy() - o[3]).mean() <= IMG_L1 and np.abs(depth.cpu().numpy() - o[2]).mean() <= IMG_L1 * max(1.0, float(np.abs(o[2]).max())), (p, i)
    return color, radii, depth, alpha


def close(got, want, mean=IMG_L1, mx=5e-3):
    g = got.detach().cpu().numpy() if torch.is_tensor(got) else np.asarray(got)
    assert g.shape == want.shape, (g.shape, want.shape)
    assert np.abs(g - want).mean() <= mean and np.abs(g - want).max() <= mx, (float(np.abs(g - want).mean()), float(np.abs(g - want).max()))


@pytest.mark.gpu
def test_flexicubes_render_mesh_calls_replayed_on_hip():
    _need_gpu()
    z = _z()
    outs = [replay_dr(z, "flex", i) for i in range(int(z["flex_ncalls"]))]
    # render_mesh #1 (return_types mask, depth, normal, vertex_normal; white_bg False): calls 0-5
    rast = T(z["flex_c0_out0"])
    alpha_index = rast[..., -1:] > 0                                         # flexicubes_renderer.py:49-51
    close(outs[1][0], z["flex_res_a_mask"])                                  # :56  img = dr.antialias(alpha, rast, v_pos_clip, faces)
    img = outs[2][0].clone()                                                 # :58-66  depth: interpolate z, clamp to [depth_min, depth_max], normalise, zero outside
    dmin, dmax = -5.5, -0.5
    m = torch.clamp(img[alpha_index], min=dmin, max=dmax)
    img[alpha_index] = (m - dmin) / (dmax - dmin)
    img[~alpha_index] = 0
    close(img, z["flex_res_a_depth"], mx=2e-3)
    close(outs[3][0], z["flex_res_a_normal"], mx=2e-3)                       # :67-69  per-face normals through an (i, i, i) index buffer
    close(outs[5][0], z["flex_res_a_vertex_normal"])                         # :70-72  antialias((vn + 1) / 2)
    # render_mesh #2 (mask, vertex_normal; white_bg True): calls 6-9, img = lerp(ones, img, alpha)  (:73-75)
    alpha = (T(z["flex_c6_out0"])[..., -1:] > 0).float()
    close(torch.lerp(torch.ones_like(outs[7][0]), outs[7][0], alpha), z["flex_res_b_mask"])
    close(torch.lerp(torch.ones_like(outs[9][0]), outs[9][0], alpha), z["flex_res_b_vertex_normal"])
    # the call site passes int64 faces through `.int()` and a fresh RasterizeCudaContext per call: what the recorded dtypes say
    assert z["flex_c0_in1"].dtype == np.int32 and z["flex_c3_in2"].dtype == np.int32


def _ids_agree(rast, recorded, what):
    """winners of the HIP rasterizer against the recorded ones: all but <= 1e-5 of the pixels (at least 3: a triangle edge through a pixel centre falls either way);
    -> bool mask [B,H,W] of the pixels within one pixel of a disagreement (antialias spreads a winner's colour to its neighbours)"""
    ids, want = rast[..., 3].cpu().numpy(), recorded[..., 3]
    bad = ids != want
    print("[f4 end-to-end] %s: %d of %d pixels have another winner" % (what, int(bad.sum()), bad.size))
    assert bad.sum() <= max(3, int(1e-5 * bad.size)), (what, int(bad.sum()))
    near = bad.copy()
    near[:, 1:] |= bad[:, :-1]; near[:, :-1] |= bad[:, 1:]; near[:, :, 1:] |= bad[:, :, :-1]; near[:, :, :-1] |= bad[:, :, 1:]
    return near


def _close_e2e(got, want, near, mean=IMG_L1, mx=5e-3):
    g = got.detach().cpu().numpy()
    assert g.shape == want.shape, (g.shape, want.shape)
    d = np.abs(g - want)
    assert d.mean() <= mean, float(d.mean())                       # the whole image, disagreeing winners included
    assert d[~near].max() <= mx, float(d[~near].max())             # everywhere but next to a pixel that went to another triangle


@pytest.mark.gpu
def test_flexicubes_render_mesh_end_to_end_on_hip():
    """FlexiCubesRenderer.render_mesh (flexicubes_renderer.py:41-74) with the HIP ops feeding each other: rasterize -> alpha -> antialias / interpolate (+ clamp, normalise)
    / interpolate over per-face normals / interpolate + antialias, twice (black and white background); inputs = what the reference's run handed to its first op."""
    _need_gpu()
    import nvdiffrast.torch as dr
    z = _z()
    a = lambda i, j: z["flex_c%d_in%d" % (i, j)]
    for first, tag, white in ((0, "a", False), (6, "b", True)):
        vclip, faces = T(a(first, 0)), T(a(first, 1), torch.int32)
        res = tuple(int(x) for x in a(first, 2))
        rast, _ = dr.rasterize(dr.RasterizeCudaContext(), vclip, faces, res)               # :44-45
        near = _ids_agree(rast, z["flex_c%d_out0" % first], "flex render_mesh #%s" % tag)[..., None]
        alpha_index = rast[..., -1:] > 0                                                     # :47-49
        alpha = alpha_index.float()
        bgmix = (lambda img: torch.lerp(torch.ones_like(img), img, alpha)) if white else (lambda img: img)      # :73-75
        mask = dr.antialias(alpha, rast, vclip, faces)                                       # :54
        _close_e2e(bgmix(mask), z["flex_res_%s_mask" % tag], near)
        if not white:
            img, _ = dr.interpolate(T(a(2, 0)), rast, faces)                                 # :57-63  depth
            dmin, dmax = -5.5, -0.5
            img = img.clone()
            img[alpha_index] = (torch.clamp(img[alpha_index], min=dmin, max=dmax) - dmin) / (dmax - dmin)
            img[~alpha_index] = 0
            _close_e2e(img, z["flex_res_a_depth"], near, mx=2e-3)
            nrm, _ = dr.interpolate(T(a(3, 0)), rast, T(a(3, 2), torch.int32))               # :65-66  per-face normals through (i, i, i)
            _close_e2e(nrm, z["flex_res_a_normal"], np.broadcast_to(near, nrm.shape), mx=2e-3)
        iv, ia_ = (4, 5) if not white else (8, 9)
        vn, _ = dr.interpolate(T(a(iv, 0)), rast, faces)                                     # :68-69  vertex normals, antialiased
        vn = dr.antialias((vn + 1) * 0.5, rast, vclip, faces)
        _close_e2e(bgmix(vn), z["flex_res_%s_vertex_normal" % tag], np.broadcast_to(near, vn.shape))


@pytest.mark.gpu
def test_color_func_to_albedo_end_to_end_on_hip():
    """color_func_to_albedo (mesh_utils.py:521-568) up to the padding: rasterize the UV layout, interpolate positions and the coverage attribute on the HIP rast, query the
    generator's colour field on the covered texels"""
    _need_gpu()
    import nvdiffrast.torch as dr
    z = _z()
    a = lambda i, j: z["bake_c%d_in%d" % (i, j)]
    h = w = int(a(0, 2)[0])
    rast, _ = dr.rasterize(dr.RasterizeCudaContext(), T(a(0, 0)), T(a(0, 1), torch.int32), (h, w))      # :537
    near = _ids_agree(rast, z["bake_c0_out0"], "color_func_to_albedo")
    xyzs, _ = dr.interpolate(T(a(1, 0)), rast, T(a(1, 2), torch.int32))                                   # :538
    mask, _ = dr.interpolate(T(a(2, 0)), rast, T(a(2, 2), torch.int32))                                   # :539
    xyzs, mask = xyzs.view(-1, 3), (mask > 0).view(-1)                                                    # :542-543
    albedo = torch.zeros(h * w, 3, device="cuda")
    albedo[mask] = torch.sigmoid(xyzs[mask] * 3.0 + torch.tensor([0.3, -0.2, 0.1], device="cuda"))       # the generator's colour field
    _close_e2e(albedo.view(1, h, w, 3), z["bake_res_albedo"][None], np.broadcast_to(near[..., None], (1, h, w, 3)), mx=1e-4)


@pytest.mark.gpu
def test_color_func_to_albedo_calls_replayed_on_hip():
    _need_gpu()
    z = _z()
    outs = [replay_dr(z, "bake", i) for i in range(3)]
    h = w = int(z["bake_c0_in2"][0])
    xyzs, mask = outs[1][0].view(-1, 3), (outs[2][0] > 0).view(-1)            # mesh_utils.py:540-545
    assert torch.equal(mask.cpu(), torch.tensor(z["bake_c0_out0"][..., 3].reshape(-1) > 0))      # interpolated ones > 0 exactly where a triangle won the texel
    albedo = torch.zeros(h * w, 3, device="cuda")
    albedo[mask] = torch.sigmoid(xyzs[mask] * 3.0 + torch.tensor([0.3, -0.2, 0.1], device="cuda"))   # the generator's colour field, queried as :555-563
    close(albedo.view(h, w, 3), z["bake_res_albedo"], mx=1e-4)
    assert float(mask.float().mean()) > 0.5


@pytest.mark.gpu
def test_lgm_gaussian_renderer_calls_replayed_on_hip():
    _need_gpu()
    z = _z()
    outs = [replay_gs(z, "lgm", i) for i in range(4)]
    S = int(z["lgm_c0_st_image_height"])
    images = torch.stack([o[0].clamp(0, 1) for o in outs], dim=0).view(2, 2, 3, S, S)            # LGM/core/gs.py:85-91
    alphas = torch.stack([o[3] for o in outs], dim=0).view(2, 2, 1, S, S)
    close(images, z["lgm_res_image"]); close(alphas, z["lgm_res_alpha"])


@pytest.mark.gpu
@pytest.mark.parametrize("head", ["sh", "rgb"])
def test_tgs_forward_single_view_calls_replayed_on_hip(head):
    _need_gpu()
    z = _z()
    p = "tgs_" + head
    rgb = replay_gs(z, p, 0, autocast=True)
    msk = replay_gs(z, p, 1, autocast=True)
    close(rgb[0].permute(1, 2, 0), z[p + "_res_comp_rgb"])                   # TriplaneGaussian/models/renderer.py:272-275
    close(msk[0].permute(1, 2, 0), z[p + "_res_comp_mask"])                  # :305
    assert rgb[0].dtype == torch.float32


@pytest.mark.gpu
def test_trellis_render_calls_replayed_on_hip():
    _need_gpu()
    z = _z()
    a, b, c = (replay_gs(z, "trellis", i) for i in range(3))
    close(a[0], z["trellis_res_a_render"]); close(b[0], z["trellis_res_b_render"]); close(c[0], z["trellis_res_c_render"])
    assert (a[1].cpu().numpy() != z["trellis_res_a_radii"]).sum() <= 2
    assert float(z["trellis_c1_st_scale_modifier"]) == pytest.approx(0.8)
