"""Regenerates tests/golden/ref_mesh_class.npz by RUNNING THE REFERENCE'S OWN `Mesh` container methods that the mesh path uses
(/root/reference/mesh_processer/mesh.py: auto_normal :471-494, aabb :450-457, auto_size :460-469, set_new_albedo :442-447) on the CPU.
Asset I/O (obj / glb / ply loaders, xatlas) is outside the hot path and not run.  Third-party imports of that module absent from this
image are stubbed (cv2, trimesh, kiui); kiui.op.safe_normalize / dot are restated (x / sqrt(clamp(sum x^2, eps)); sum(x*y, -1, keepdim)),
kiui.typing is given the names of `typing` plus Tensor.

  python tests/golden/make_golden_ref_mesh_class.py [--check]
"""
import os
import sys
import types
import typing

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), HERE]
import make_golden_ref_py as G  # noqa: E402
import make_golden_ref_render as GR  # noqa: E402

OUT = os.path.join(HERE, "ref_mesh_class.npz")


def reference_mesh_module():
    G._install_stubs()
    G._cpu_redirect()
    for name in ("torchvision", "torchvision.transforms", "torchvision.transforms.functional", "PIL", "PIL.Image"):
        if name not in sys.modules:
            try:
                __import__(name)
            except Exception:
                m = G._Stub(name); m.__path__ = []
                sys.modules[name] = m
    kt = sys.modules["kiui.typing"]
    names = [n for n in typing.__all__]
    for n in names:
        setattr(kt, n, getattr(typing, n))
    kt.Tensor, kt.ndarray = torch.Tensor, np.ndarray
    kt.__all__ = names + ["Tensor", "ndarray"]
    sys.modules["kiui.op"].safe_normalize = lambda x, eps=1e-20: x / torch.sqrt(torch.clamp(torch.sum(x * x, -1, keepdim=True), min=eps))
    sys.modules["kiui.op"].dot = lambda x, y: torch.sum(x * y, -1, keepdim=True)
    su = types.ModuleType("shared_utils"); su.__path__ = [os.path.join(G.REF, "shared_utils")]
    sys.modules["shared_utils"] = su
    G._load("shared_utils.sh_utils", "shared_utils/sh_utils.py")
    G._load("shared_utils.image_utils", "shared_utils/image_utils.py")
    del sys.modules["mesh_processer.mesh"]
    return G._load("ref_mesh_module", "mesh_processer/mesh.py")


def generate():
    mm = reference_mesh_module()
    sc = GR.scene()
    rng = np.random.default_rng(3)
    v = sc["v"].copy()
    v[7] = v[8]                                                       # a degenerate neighbourhood
    f = np.concatenate([sc["f"], [[0, 0, 0]]]).astype(np.int32)       # and a degenerate face
    T = torch.from_numpy
    out = {"v": v, "f": f}
    m = mm.Mesh(v=T(v.copy()), f=T(f.copy()), device=torch.device("cpu"))
    m.auto_normal()
    out["vn"], out["fn"] = m.vn.numpy().copy(), m.fn.numpy().copy()
    lo, hi = m.aabb()
    out["aabb_min"], out["aabb_max"] = lo.numpy().copy(), hi.numpy().copy()
    m.auto_size(bound=0.9)
    out["sized_v"], out["ori_center"], out["ori_scale"] = m.v.numpy().copy(), np.asarray(m.ori_center), np.float64(m.ori_scale)
    m.set_new_albedo(6, 10)
    out["gray_albedo"] = m.albedo.numpy().copy()
    tex = rng.uniform(0, 1, (8, 8, 3)).astype(np.float32)
    out["tex"] = tex
    m2 = mm.Mesh(v=T(v.copy()), f=T(f.copy()), albedo=T(tex.copy()), device=torch.device("cpu"))
    m2.set_new_albedo(12, 10)
    out["resized_albedo"] = m2.albedo.numpy().copy()
    return {k: np.asarray(val) for k, val in out.items()}


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
