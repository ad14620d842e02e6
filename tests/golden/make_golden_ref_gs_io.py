"""Regenerates tests/golden/ref_gs_io.npz by RUNNING THE REFERENCE'S OWN model construction and PLY wire-format code on the CPU:

  GaussianModel.create_from_pcd (main_3DGS_renderer.py:407-433), .to_ply (:475-484), .create_from_ply (:486-498), .get_points_cloud (:468-473)
  mesh_processer/mesh_utils.py  construct_list_of_gs_attributes (:333-344), calculate_max_sh_degree_from_gs_ply (:346-350),
                                write_gs_ply (:352-360), read_gs_ply (:362-390), switch_vector_axis / switch_ply_axis_and_scale (:446-472)

Replaced third-party packages (absent from this image):
  * plyfile            -> comfyui-3d-pack_amd/c3d_hip/ply.py, this repo's reader / writer with the slice of the plyfile API the reference
                          uses.  The layout logic that runs on top of it (attribute order, channel-major SH flattening, reshapes) is the
                          reference's; the container class is shared by both sides of the comparison.
  * simple_knn._C.distCUDA2 -> oracle/knn_oracle.py (exact 3-nearest-neighbour mean squared distance on the CPU);
  * kornia's quaternion_to_axis_angle / axis_angle_to_quaternion (used by switch_ply_axis_and_scale) -> the mirror's own conversions,
    shared by both sides and therefore not pinned; kiui.op.inverse_sigmoid restated.

  python tests/golden/make_golden_ref_gs_io.py [--check]
"""
import importlib.util
import os
import sys
import types
from typing import NamedTuple

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), HERE]
import make_golden_ref_py as G  # noqa: E402

OUT = os.path.join(HERE, "ref_gs_io.npz")
PKG = os.path.join(ROOT, "comfyui-3d-pack_amd")


def _load_file(name, path):
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    sys.modules[name] = mod
    spec.loader.exec_module(mod)
    return mod


class PointCloud(NamedTuple):       # mesh_processer/mesh.py:903-906 (that module needs cv2 / trimesh / kiui at import)
    points: np.ndarray
    colors: np.ndarray
    normals: np.ndarray


def mirror_quaternion_conversions():
    src = open(os.path.join(PKG, "mesh_processer", "mesh_utils.py")).read()
    ns = {"torch": torch, "np": np}
    start, end = src.index("def _quat_to_axis_angle"), src.index("def switch_ply_axis_and_scale")
    exec(compile(src[start:end], "mirror_quat", "exec"), ns)
    return ns["_quat_to_axis_angle"], ns["_axis_angle_to_quat"]


def reference_modules():
    G._install_stubs()
    G._cpu_redirect()
    ply = _load_file("plyfile", os.path.join(PKG, "c3d_hip", "ply.py"))
    sys.modules["kiui.op"].inverse_sigmoid = lambda x: torch.log(x / (1 - x))
    q2a, a2q = mirror_quaternion_conversions()
    wrap = lambda f: (lambda t: torch.from_numpy(np.asarray(f(t.detach().cpu().numpy()), dtype=np.float32)))      # kornia works on tensors
    sys.modules["kornia.geometry.conversions"].quaternion_to_axis_angle = wrap(q2a)
    sys.modules["kornia.geometry.conversions"].axis_angle_to_quaternion = wrap(a2q)
    from oracle import knn_oracle
    sys.modules["simple_knn._C"].distCUDA2 = lambda pts: torch.from_numpy(knn_oracle.dist2_mean3(pts.detach().cpu().numpy()))
    sys.modules["mesh_processer.mesh"].PointCloud = PointCloud
    su = types.ModuleType("shared_utils"); su.__path__ = [os.path.join(G.REF, "shared_utils")]
    sys.modules["shared_utils"] = su
    G._load("shared_utils.sh_utils", "shared_utils/sh_utils.py")
    pkg = types.ModuleType("ref_mp"); pkg.__path__ = [os.path.join(G.REF, "mesh_processer")]
    sys.modules["ref_mp"] = pkg
    sys.modules["ref_mp.mesh"] = sys.modules["mesh_processer.mesh"]
    mu = G._load("ref_mp.mesh_utils", "mesh_processer/mesh_utils.py")
    sys.modules["mesh_processer.mesh_utils"] = mu
    _, _, ren = G.reference_modules()
    return ren, mu, ply


def ply_matrix(plydata):
    el = plydata.elements[0]
    names = [p.name for p in el.properties]
    return np.asarray(names), np.stack([np.asarray(el[n], dtype=np.float32) for n in names], axis=1), np.asarray([p.dtype if hasattr(p, "dtype") else "" for p in el.properties])


def generate():
    ren, mu, ply = reference_modules()
    rng = np.random.default_rng(41)
    N = 70
    out = {}
    pts = rng.normal(size=(N, 3)) * 0.4
    pts[5] = pts[4]                                                   # a duplicate: distance 0 clamps to 1e-7
    cols = rng.uniform(0, 1, (N, 3))
    out["pcd_points"], out["pcd_colors"] = pts, cols
    for deg in (3, 1):
        g = ren.GaussianModel(deg)
        g.create_from_pcd(PointCloud(points=pts, colors=cols, normals=np.zeros_like(pts)), 7.5)
        pre = "deg%d_" % deg
        for k in ("xyz", "features_dc", "features_rest", "scaling", "rotation", "opacity"):
            out[pre + "pcd_" + k] = getattr(g, "_" + k).detach().numpy().copy()
        out[pre + "pcd_init_xyz"], out[pre + "pcd_max_radii2D"] = g.init_xyz.numpy().copy(), g.max_radii2D.numpy().copy()
        out[pre + "pcd_spatial_lr_scale"] = np.float64(g.spatial_lr_scale)
        with torch.no_grad():                                         # a trained-looking model: every property distinct
            g._features_rest.add_(torch.from_numpy(rng.normal(size=tuple(g._features_rest.shape)).astype(np.float32)))
            g._rotation.add_(torch.from_numpy(rng.normal(size=(N, 4)).astype(np.float32)))
            g._scaling.add_(torch.from_numpy(rng.normal(size=(N, 3)).astype(np.float32) * 0.3))
            g._opacity.add_(torch.from_numpy(rng.normal(size=(N, 1)).astype(np.float32)))
        for k in ("xyz", "features_dc", "features_rest", "scaling", "rotation", "opacity"):
            out[pre + "model_" + k] = getattr(g, "_" + k).detach().numpy().copy()
        pd = g.to_ply()
        names, mat, dts = ply_matrix(pd)
        out[pre + "ply_names"], out[pre + "ply_data"], out[pre + "ply_dtypes"] = names, mat, dts
        out[pre + "ply_max_sh_degree"] = np.int64(mu.calculate_max_sh_degree_from_gs_ply(pd)[0])
        for k, v in zip(("xyz", "features_dc", "features_extra", "opacities", "scales", "rots"), mu.read_gs_ply(pd)):
            out[pre + "read_" + k] = v
        g2 = ren.GaussianModel(deg)
        g2.create_from_ply(pd)
        for k in ("xyz", "features_dc", "features_rest", "scaling", "rotation", "opacity"):
            out[pre + "fromply_" + k] = getattr(g2, "_" + k).detach().numpy()
        out[pre + "fromply_active_sh_degree"] = np.int64(g2.active_sh_degree)
        pc = g.get_points_cloud()
        out[pre + "cloud_points"], out[pre + "cloud_colors"] = np.asarray(pc.points), np.asarray(pc.colors)
        for tag, axis, scale, inv in (("swapA", [2, 0, 1], [1.0, 1.0, 1.0], 0), ("swapB", [0, 2, 1], [1.0, -1.0, 2.0], 1)):
            sw = mu.switch_ply_axis_and_scale(pd, axis, scale, inv)
            n2, m2, _ = ply_matrix(sw)
            out[pre + tag + "_names"], out[pre + tag + "_data"] = n2, m2
    return {k: np.asarray(v) for k, v in out.items()}


def main():
    out = generate()
    if "--check" in sys.argv:
        ref = np.load(OUT)
        bad = [k for k in out if k not in ref.files or out[k].shape != ref[k].shape or not np.array_equal(out[k], ref[k])]
        bad += [k for k in ref.files if k not in out]
        if bad:
            print("MISMATCH:", bad)
            sys.exit(1)
        print("ok: %d arrays identical to the committed fixture" % len(out))
        return
    np.savez_compressed(OUT, **out)
    print("wrote", OUT, "with", len(out), "arrays")


if __name__ == "__main__":
    main()
