"""Regenerates tests/golden/ref_consumers.npz by RUNNING THE REFERENCE'S OWN CODE of the five other consumers of the two rasterizers
 on the CPU in this container, over tests/fake_dr.py / tests/fake_dgr.py (the CPU oracles behind the wheels' names):

  flex   /root/reference/MVs_Algorithms/FlexiCubes/flexicubes_renderer.py:28-74      FlexiCubesRenderer.get_orbit_camera + render_mesh (mask, depth, normal, vertex_normal; white_bg)
  bake   /root/reference/mesh_processer/mesh_utils.py:521-568                       color_func_to_albedo: rasterize in UV space with `ft`, interpolate positions / ones with `f`
  lgm    /root/reference/Gen_3D_Modules/LGM/core/gs.py:11-97                        GaussianRenderer.render: [B, N, 14] Gaussians, precomputed colours, sh_degree 0, B x V views
  tgs    /root/reference/Gen_3D_Modules/TriplaneGaussian/models/renderer.py:203-306  GS3DRenderer.forward_single_view: SH colours under torch.autocast(float32) + the white-on-black mask pass
  trellis /root/reference/Gen_3D_Modules/TRELLIS/trellis/renderers/gaussian_render.py:50-144   render(): SH features, optional precomputed covariance / override colour

Every call these functions make into `nvdiffrast.torch` / `diff_gaussian_rasterization` is RECORDED with its inputs and outputs, together with what the function
returned.  tests/test_zz_ref_consumers.py replays the recorded calls through the HIP drop-ins on the GPU box (where /root/reference does not exist) and rebuilds the
functions' results from them.  What this pins: the argument patterns, shapes, dtypes and op order of the real call sites -- not restatements of them.

Third-party pieces the files import and this image lacks, and how they are stood in for (none of them is the code under test):
  kiui.cam.orbit_camera    restated from the published formula: x = r cos e sin a, y = -r sin e, z = r cos e cos a, look-at, OpenGL convention
  kiui.op.uv_padding       identity (texture dilation by nearest neighbour: after the ops, outside the path)
  easydict.EasyDict        dict with attribute access
  TGS utils (typing / base / ops / networks), TRELLIS representations: inert stubs; the methods are called on stand-in objects that carry the fields they read

  python tests/golden/make_golden_ref_consumers.py [--check]
"""
import math
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), HERE]
import make_golden_ref_py as G  # noqa: E402

REF = G.REF
OUT = os.path.join(HERE, "ref_consumers.npz")
REC = []          # recorded op calls of the section being generated


def _np(x):
    if torch.is_tensor(x):
        return x.detach().cpu().numpy().copy()      # a copy: the call sites go on to modify op outputs in place (render_mesh's depth normalisation)
    return x


def _orbit_camera(elevation, azimuth, radius=1.0, is_degree=True, target=None, opengl=True):
    """kiui.cam.orbit_camera as published (0.2.x): camera-to-world 4x4"""
    if is_degree:
        elevation, azimuth = np.deg2rad(elevation), np.deg2rad(azimuth)
    x = radius * np.cos(elevation) * np.sin(azimuth)
    y = -radius * np.sin(elevation)
    z = radius * np.cos(elevation) * np.cos(azimuth)
    target = np.zeros(3, dtype=np.float32) if target is None else target
    campos = np.array([x, y, z]) + target

    def nrm(v):
        return v / (np.linalg.norm(v) + 1e-20)
    fwd = nrm(campos - target)                       # opengl: camera looks along -z
    right = nrm(np.cross(np.array([0, 1, 0], dtype=np.float32), fwd))
    up = nrm(np.cross(fwd, right))
    T = np.eye(4, dtype=np.float32)
    T[:3, :3] = np.stack([right, up, fwd], axis=1)
    T[:3, 3] = campos
    return T


def install():
    """stubs + the two recording stand-ins"""
    G._install_stubs()
    G._cpu_redirect()
    import fake_dr
    import fake_dgr
    sys.modules["kiui.cam"].orbit_camera = _orbit_camera
    sys.modules["kiui.op"].uv_padding = lambda image, mask, padding=None, **k: image
    ed = types.ModuleType("easydict")

    class EasyDict(dict):
        __getattr__ = dict.__getitem__
        __setattr__ = dict.__setitem__
    ed.EasyDict = EasyDict
    sys.modules["easydict"] = ed

    # ---- nvdiffrast.torch: every op call recorded with inputs and outputs
    def rec_dr(name, fn):
        def wrapped(*a, **k):
            out = fn(*a, **k)
            args = [_np(x) for x in a if not isinstance(x, fake_dr.RasterizeCudaContext)]
            outs = [_np(o) for o in (out if isinstance(out, tuple) else (out,))]
            REC.append({"op": name, "args": args, "kwargs": {kk: _np(v) for kk, v in k.items()}, "outs": outs})
            return out
        return wrapped
    if not getattr(fake_dr, "_recording", False):
        for nm in ("rasterize", "interpolate", "texture", "antialias"):
            setattr(fake_dr, nm, rec_dr(nm, getattr(fake_dr, nm)))
        fake_dr._recording = True
    nv = types.ModuleType("nvdiffrast"); nv.torch = fake_dr; nv.__path__ = []
    sys.modules["nvdiffrast"], sys.modules["nvdiffrast.torch"] = nv, fake_dr

    # ---- diff_gaussian_rasterization: the rasterizer call recorded with settings, tensors and outputs
    if not getattr(fake_dgr, "_recording", False):
        orig = fake_dgr.GaussianRasterizer.__call__

        def call(self, means3D, means2D, opacities, shs=None, colors_precomp=None, scales=None, rotations=None, cov3D_precomp=None):
            out = orig(self, means3D, means2D, opacities, shs=shs, colors_precomp=colors_precomp, scales=scales, rotations=rotations, cov3D_precomp=cov3D_precomp)
            rs = self.raster_settings
            REC.append({"op": "gs", "settings": {kk: (_np(v) if torch.is_tensor(v) else v) for kk, v in rs._asdict().items()},
                        "args": {"means3D": _np(means3D), "opacities": _np(opacities), "shs": _np(shs), "colors_precomp": _np(colors_precomp), "scales": _np(scales),
                                 "rotations": _np(rotations), "cov3D_precomp": _np(cov3D_precomp)},
                        "autocast": bool(torch.is_autocast_enabled("cpu")) if hasattr(torch, "is_autocast_enabled") else False,
                        "outs": [_np(o) for o in out]})
            return out
        fake_dgr.GaussianRasterizer.__call__ = call
        fake_dgr.RECORD = False
        fake_dgr._recording = True
    sys.modules["diff_gaussian_rasterization"] = fake_dgr
    return fake_dr, fake_dgr


def dump(prefix, out, result):
    """flatten the recorded calls + the function's result into `out`"""
    out[prefix + "_ncalls"] = np.asarray(len(REC))
    for i, c in enumerate(REC):
        p = "%s_c%d_" % (prefix, i)
        out[p + "op"] = np.asarray(c["op"])
        if c["op"] == "gs":
            for k, v in c["settings"].items():
                out[p + "st_" + k] = np.asarray(v)
            for k, v in c["args"].items():
                if v is not None:
                    out[p + "in_" + k] = np.asarray(v)
            out[p + "autocast"] = np.asarray(c["autocast"])
        else:
            for j, v in enumerate(c["args"]):
                out[p + "in%d" % j] = np.asarray(v if v is not None else np.zeros(0))
                out[p + "none%d" % j] = np.asarray(v is None)
            for k, v in c["kwargs"].items():
                out[p + "kw_" + k] = np.asarray(v if v is not None else np.zeros(0))
        for j, v in enumerate(c["outs"]):
            out[p + "out%d" % j] = np.asarray(v)
    for k, v in result.items():
        out[prefix + "_res_" + k] = np.asarray(_np(v))
    del REC[:]


# ---------------------------------------------------------------------------------------------------------------- scenes
def sphere_mesh(rng, n_lat=8, n_lon=12, r=0.8):
    th = np.linspace(0.15, np.pi - 0.15, n_lat)
    ph = np.linspace(0, 2 * np.pi, n_lon, endpoint=False)
    T, P = np.meshgrid(th, ph, indexing="ij")
    d = np.stack([np.sin(T) * np.cos(P), np.cos(T), np.sin(T) * np.sin(P)], -1).reshape(-1, 3)
    v = (d * r * (1 + 0.1 * rng.normal(size=(d.shape[0], 1)))).astype(np.float32)
    vt = np.stack([0.05 + 0.9 * P / (2 * np.pi), 0.05 + 0.9 * T / np.pi], -1).reshape(-1, 2).astype(np.float32)
    f = []
    for i in range(n_lat - 1):
        for j in range(n_lon - 1):                       # no seam column: the chart stays a proper parametrisation
            a, b = i * n_lon + j, i * n_lon + j + 1
            f += [[a, a + n_lon, b], [b, a + n_lon, b + n_lon]]
    return v, np.asarray(f, np.int32), vt, d.astype(np.float32)


def gaussians(rng, n, sh_k=4):
    xyz = (rng.normal(size=(n, 3)) * 0.25).astype(np.float32)
    opacity = rng.uniform(0.2, 0.95, size=(n, 1)).astype(np.float32)
    scales = np.exp(rng.normal(np.log(0.05), 0.4, size=(n, 3))).astype(np.float32)
    q = rng.normal(size=(n, 4)).astype(np.float32); q /= np.linalg.norm(q, axis=1, keepdims=True)
    rgb = rng.uniform(0.05, 0.95, size=(n, 3)).astype(np.float32)
    shs = (rng.normal(size=(n, sh_k, 3)) * 0.3).astype(np.float32)
    return xyz, opacity, scales, q, rgb, shs


def look_camera(elev, azim, radius, fovy_deg, H, W, znear=0.1, zfar=100.0):
    """world_view_transform / full_proj_transform / camera_center in the convention the rasterizer receives (row-vector storage)"""
    c2w = _orbit_camera(elev, azim, radius)
    w2c = np.linalg.inv(c2w)
    w2c[1:3, :3] *= -1; w2c[:3, 3] *= -1                 # OpenGL -> COLMAP, as shared_utils/camera_utils.py:199-203
    view = torch.tensor(w2c, dtype=torch.float32).transpose(0, 1)
    fovy = np.deg2rad(fovy_deg); fovx = 2 * np.arctan(np.tan(fovy / 2) * W / H)
    ty, tx = math.tan(fovy / 2), math.tan(fovx / 2)
    P = torch.zeros(4, 4)
    P[0, 0] = 1 / tx; P[1, 1] = 1 / ty; P[3, 2] = 1.0; P[2, 2] = zfar / (zfar - znear); P[2, 3] = -(zfar * znear) / (zfar - znear)
    full = view @ P.transpose(0, 1)
    return types.SimpleNamespace(FoVx=float(fovx), FoVy=float(fovy), image_height=H, image_width=W, height=H, width=W, world_view_transform=view,
                                 full_proj_transform=full, camera_center=torch.tensor(-c2w[:3, 3], dtype=torch.float32))


# ---------------------------------------------------------------------------------------------------------------- the five consumers
def gen_flex(out):
    pkg = types.ModuleType("FlexiCubes"); pkg.__path__ = [os.path.join(REF, "MVs_Algorithms/FlexiCubes")]
    sys.modules["FlexiCubes"] = pkg
    util = G._load("FlexiCubes.util", "MVs_Algorithms/FlexiCubes/util.py")
    pkg.util = util
    fr = G._load("FlexiCubes.flexicubes_renderer", "MVs_Algorithms/FlexiCubes/flexicubes_renderer.py")
    rng = np.random.default_rng(5)
    v, f, vt, vn = sphere_mesh(rng)
    fi = f.astype(np.int64)
    fnrm = np.cross(v[fi[:, 1]] - v[fi[:, 0]], v[fi[:, 2]] - v[fi[:, 0]]); fnrm /= np.linalg.norm(fnrm, axis=1, keepdims=True) + 1e-20
    mesh = types.SimpleNamespace(vertices=torch.tensor(v), faces=torch.tensor(f.astype(np.int64)), nrm=torch.tensor(fnrm.astype(np.float32)), v_nrm=torch.tensor(vn))
    r = fr.FlexiCubesRenderer(True)
    res = [40, 56]
    mv, mvp = r.get_orbit_camera(azimuth=35.0, elevation=20.0, fovy=45, iter_res=res, cam_radius=3.0, device="cpu")
    with torch.no_grad():
        o1 = r.render_mesh(mesh, mv.unsqueeze(0), mvp.unsqueeze(0), res, return_types=["mask", "depth", "normal", "vertex_normal"], white_bg=False)
        o2 = r.render_mesh(mesh, mv.unsqueeze(0), mvp.unsqueeze(0), res, return_types=["mask", "vertex_normal"], white_bg=True)
    result = {"mv": mv, "mvp": mvp, "v": v, "f": f, "fnrm": fnrm.astype(np.float32), "vn": vn}
    result.update({"a_" + k: x for k, x in o1.items()}); result.update({"b_" + k: x for k, x in o2.items()})
    dump("flex", out, result)


def gen_bake(out):
    pkg = types.ModuleType("mesh_processer"); pkg.__path__ = [os.path.join(REF, "mesh_processer")]
    saved = {k: sys.modules.get(k) for k in ("mesh_processer", "mesh_processer.mesh_utils")}
    sys.modules["mesh_processer"] = pkg
    su = types.ModuleType("shared_utils"); su.__path__ = [os.path.join(REF, "shared_utils")]
    sys.modules["shared_utils"] = su
    G._load("shared_utils.sh_utils", "shared_utils/sh_utils.py")
    mu = G._load("mesh_processer.mesh_utils", "mesh_processer/mesh_utils.py")
    rng = np.random.default_rng(6)
    v, f, vt, vn = sphere_mesh(rng)
    mesh = types.SimpleNamespace(v=torch.tensor(v), f=torch.tensor(f), vt=torch.tensor(vt), ft=torch.tensor(f))
    rgb = lambda xyz: torch.sigmoid(xyz * 3.0 + torch.tensor([0.3, -0.2, 0.1]))          # a smooth colour field: the "network" the reference queries
    import fake_dr
    with torch.no_grad():
        albedo = mu.color_func_to_albedo(mesh, rgb, texture_resolution=48, padding=2, batch_size=500, device="cpu", force_cuda_rast=True, glctx=fake_dr.RasterizeCudaContext())
    dump("bake", out, {"albedo": albedo, "v": v, "f": f, "vt": vt})
    for k, m in saved.items():
        if m is not None:
            sys.modules[k] = m


def gen_lgm(out):
    pkg = types.ModuleType("LGM"); pkg.__path__ = [os.path.join(REF, "Gen_3D_Modules/LGM")]
    core = types.ModuleType("LGM.core"); core.__path__ = [os.path.join(REF, "Gen_3D_Modules/LGM/core")]
    sys.modules["LGM"], sys.modules["LGM.core"] = pkg, core
    opts = G._load("LGM.core.options", "Gen_3D_Modules/LGM/core/options.py")
    gs = G._load("LGM.core.gs", "Gen_3D_Modules/LGM/core/gs.py")
    opt = opts.Options(); opt.output_size = 40
    r = gs.GaussianRenderer(opt)
    rng = np.random.default_rng(7)
    B, V, N = 2, 2, 300
    g = np.zeros((B, N, 14), np.float32)
    for b in range(B):
        xyz, op, sc, q, rgb, _ = gaussians(rng, N)
        g[b] = np.concatenate([xyz, op, sc, q, rgb], 1)
    cams = [look_camera(el, az, 1.5, opt.fovy, 40, 40) for el, az in ((0.0, 0.0), (20.0, 120.0), (-25.0, 240.0), (45.0, 30.0))]
    # LGM's own camera convention: cam_view = w2c^T with its fixed projection; built here from the same quantities the rasterizer receives
    cam_view = torch.stack([c.world_view_transform for c in cams]).view(B, V, 4, 4)
    cam_view_proj = torch.stack([c.world_view_transform @ r.proj_matrix for c in cams]).view(B, V, 4, 4)
    cam_pos = torch.stack([c.camera_center for c in cams]).view(B, V, 3)
    with torch.no_grad():
        res = r.render(torch.tensor(g), cam_view, cam_view_proj, cam_pos, scale_modifier=1)
    dump("lgm", out, {"image": res["image"], "alpha": res["alpha"]})


def _typing_stub():
    m = types.ModuleType("tgs_typing_stub")
    import typing
    names = {k: getattr(typing, k) for k in ("Any", "Callable", "Dict", "Iterable", "List", "Literal", "NamedTuple", "NewType", "Optional", "Sized", "Tuple", "Type", "TypeVar", "Union")}
    names["Tensor"] = torch.Tensor

    class _Sub:
        def __getitem__(self, item):
            return torch.Tensor
    for k in ("Float", "Int", "Bool", "Num", "Shaped", "Integer"):
        names[k] = _Sub()
    names["DictConfig"] = dict
    for k, v in names.items():
        setattr(m, k, v)
    m.__all__ = list(names)
    return m


def gen_tgs(out):
    base = "Gen_3D_Modules/TriplaneGaussian"
    for name, path in (("tgs", base), ("tgs.utils", base + "/utils"), ("tgs.models", base + "/models")):
        p = types.ModuleType(name); p.__path__ = [os.path.join(REF, path)]
        sys.modules[name] = p
    sys.modules["tgs.utils.typing"] = _typing_stub()
    import dataclasses

    class BaseModule(torch.nn.Module):              # what the class bodies of renderer.py need of threestudio-style BaseModule: a dataclass `Config` to derive from
        @dataclasses.dataclass
        class Config:
            pass
    ub = G._Stub("tgs.utils.base")
    ub.BaseModule = BaseModule
    sys.modules["tgs.utils.base"] = ub
    for nm in ("tgs.utils.ops", "tgs.models.networks"):
        sys.modules[nm] = G._Stub(nm)
    import importlib.util
    spec = importlib.util.spec_from_file_location("tgs.models.renderer", os.path.join(REF, base, "models/renderer.py"))
    mod = importlib.util.module_from_spec(spec); mod.__package__ = "tgs.models"
    sys.modules["tgs.models.renderer"] = mod
    spec.loader.exec_module(mod)
    rng = np.random.default_rng(8)
    N, H, W = 260, 36, 48
    xyz, op, sc, q, rgb, shs = gaussians(rng, N, sh_k=4)
    T = torch.tensor
    for tag, use_rgb, deg in (("sh", False, 1), ("rgb", True, 0)):
        gsm = types.SimpleNamespace(xyz=T(xyz), opacity=T(op), scaling=T(sc), rotation=T(q), shs=T(rgb[:, None, :]) if use_rgb else T(shs))
        me = types.SimpleNamespace(device=torch.device("cpu"), cfg=types.SimpleNamespace(scaling_modifier=1.0, sh_degree=deg),
                                   gs_net=types.SimpleNamespace(cfg=types.SimpleNamespace(use_rgb=use_rgb)))
        cam = look_camera(15.0, 60.0, 1.6, 40.0, H, W)
        with torch.no_grad():
            ret = mod.GS3DRenderer.forward_single_view(me, gsm, cam, torch.tensor([1.0, 1.0, 1.0]), ret_mask=True)
        dump("tgs_" + tag, out, {"comp_rgb": ret["comp_rgb"], "comp_mask": ret["comp_mask"]})


def gen_trellis(out):
    base = "Gen_3D_Modules/TRELLIS/trellis"
    for name, path in (("trellis", base), ("trellis.renderers", base + "/renderers"), ("trellis.representations", base + "/representations")):
        p = types.ModuleType(name); p.__path__ = [os.path.join(REF, path)]
        sys.modules[name] = p
    rg = G._Stub("trellis.representations.gaussian"); rg.Gaussian = object
    sys.modules["trellis.representations.gaussian"] = rg
    import importlib.util
    for nm, rel in (("trellis.renderers.sh_utils", "renderers/sh_utils.py"), ("trellis.renderers.gaussian_render", "renderers/gaussian_render.py")):
        spec = importlib.util.spec_from_file_location(nm, os.path.join(REF, base, rel))
        mod = importlib.util.module_from_spec(spec); mod.__package__ = "trellis.renderers"
        sys.modules[nm] = mod
        spec.loader.exec_module(mod)
    mod = sys.modules["trellis.renderers.gaussian_render"]
    rng = np.random.default_rng(9)
    N, H, W = 280, 40, 40
    xyz, op, sc, q, rgb, shs = gaussians(rng, N, sh_k=4)
    T = torch.tensor
    pc = types.SimpleNamespace(get_xyz=T(xyz), get_opacity=T(op), get_scaling=T(sc), get_rotation=T(q), get_features=T(shs), active_sh_degree=1, max_sh_degree=1)
    cam = look_camera(-10.0, 200.0, 1.7, 40.0, H, W)
    ed = sys.modules["easydict"].EasyDict
    bg = torch.tensor([0.0, 0.0, 0.0])
    with torch.no_grad():
        r1 = mod.render(cam, pc, ed({"convert_SHs_python": False, "compute_cov3D_python": False, "debug": False}), bg, scaling_modifier=1.0)
        r2 = mod.render(cam, pc, ed({"convert_SHs_python": False, "compute_cov3D_python": False, "debug": False}), bg, scaling_modifier=0.8, override_color=T(rgb))
        r3 = mod.render(cam, pc, ed({"convert_SHs_python": True, "compute_cov3D_python": False, "debug": False}), bg)
    dump("trellis", out, {"a_render": r1["render"], "a_radii": r1["radii"], "b_render": r2["render"], "c_render": r3["render"]})


def generate():
    assert os.path.isdir(REF), "needs /root/reference"
    install()
    out = {}
    for fn in (gen_flex, gen_bake, gen_lgm, gen_tgs, gen_trellis):
        del REC[:]
        fn(out)
    return {k: np.asarray(v) for k, v in out.items()}


def main():
    out = generate()
    if "--check" in sys.argv:
        ref = np.load(OUT)
        bad = [k for k in out if k not in ref.files or out[k].shape != ref[k].shape or not np.array_equal(out[k], ref[k])]
        bad += [k for k in ref.files if k not in out]
        if bad:
            print("MISMATCH:", bad[:20])
            sys.exit(1)
        print("ok: %d arrays identical to the committed fixture" % len(out))
        return
    np.savez_compressed(OUT, **out)
    print("wrote", OUT, "with", len(out), "arrays,", os.path.getsize(OUT), "bytes")


if __name__ == "__main__":
    main()
