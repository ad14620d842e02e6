"""Regenerates tests/golden/mesh_hy_raster.npz by RUNNING REFERENCE CODE in this container:

  * the reference's Python wrapper `custom_rasterizer/render.py` (`rasterize` :19-23, `interpolate` :26-31), imported from where it
    lies under /root/reference, on top of
  * the CPU half of its `custom_rasterizer_kernel` (lib/custom_rasterizer_kernel/rasterizer.cpp:94-133) compiled by
    oracle/ref_build.py into oracle/_ref/.

That rasterizer is the stand-alone replacement Hunyuan3D's texture renderer uses for `dr.rasterize` / `dr.interpolate`
(differentiable_renderer/mesh_render.py:165-191): same inputs (clip-space pos [1,V,4], tri [T,3], resolution), outputs
(face id + 1, three perspective-correct barycentrics), so it pins the conventions of the mesh oracle that the two share -- see
tests/test_ref_pin.py for what is and is not comparable.  Needs /root/reference; the vectors it writes are what travels.
Run from the repo root:  python tests/golden/make_golden_mesh.py
"""
import importlib.util
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT]
from oracle import ref_build  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))
REF_WRAPPER = "/root/reference/Gen_3D_Modules/Hunyuan3D_V2/hy3dgen/texgen/custom_rasterizer/custom_rasterizer/render.py"


def reference_wrapper():
    """The reference's render.py with `custom_rasterizer_kernel` resolved to the module compiled from the reference's own source."""
    assert ref_build.build(), "needs /root/reference"
    sys.modules["custom_rasterizer_kernel"] = ref_build.load()
    spec = importlib.util.spec_from_file_location("hy_render_ref", REF_WRAPPER)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def soup(T, seed):
    """T independent triangles with per-vertex w in [0.6, 2.5] (perspective matters), z/w inside (-1, 1)."""
    r = np.random.default_rng(seed)
    xy = r.uniform(-0.9, 0.9, (T, 1, 2)) + r.normal(0, 0.25, (T, 3, 2))
    z = np.clip(r.uniform(-0.7, 0.7, (T, 1, 1)) + r.normal(0, 0.08, (T, 3, 1)), -0.95, 0.95)
    w = r.uniform(0.6, 2.5, (T, 3, 1))
    pos = np.concatenate([np.concatenate([xy, z], -1) * w, w], -1).reshape(1, T * 3, 4).astype(np.float32)
    return pos, np.arange(T * 3, dtype=np.int32).reshape(T, 3)


def sphere(nu, nv, seed):
    """A closed UV sphere seen through a perspective camera: shared vertices, front and back faces overlap in depth order."""
    r = np.random.default_rng(seed)
    th = np.linspace(0, np.pi, nv + 1)[1:-1]; ph = np.linspace(0, 2 * np.pi, nu, endpoint=False)
    v = [[0, 1, 0]] + [[np.sin(t) * np.cos(p), np.cos(t), np.sin(t) * np.sin(p)] for t in th for p in ph] + [[0, -1, 0]]
    v = np.asarray(v, np.float64) * 0.8 + r.normal(0, 0.01, (len(v), 3))
    f = []
    ring = lambda i, j: 1 + i * nu + (j % nu)
    for j in range(nu):
        f.append([0, ring(0, j + 1), ring(0, j)])
        f.append([len(v) - 1, ring(nv - 2, j), ring(nv - 2, j + 1)])
    for i in range(nv - 2):
        for j in range(nu):
            f.append([ring(i, j), ring(i, j + 1), ring(i + 1, j)])
            f.append([ring(i + 1, j), ring(i, j + 1), ring(i + 1, j + 1)])
    a, b = 0.4, -0.3
    Ry = np.array([[np.cos(a), 0, np.sin(a)], [0, 1, 0], [-np.sin(a), 0, np.cos(a)]])
    Rx = np.array([[1, 0, 0], [0, np.cos(b), -np.sin(b)], [0, np.sin(b), np.cos(b)]])
    cam = v @ (Rx @ Ry).T + np.array([0.1, -0.05, -2.6])
    n, fa, t = 0.5, 10.0, np.tan(np.radians(25.0))
    clip = np.stack([cam[:, 0] / t, cam[:, 1] / t, -(fa + n) / (fa - n) * cam[:, 2] - 2 * fa * n / (fa - n), -cam[:, 2]], -1)
    return clip[None].astype(np.float32), np.asarray(f, np.int32)


CASES = [("soup64", lambda: soup(40, 0), (64, 64)), ("soup_odd", lambda: soup(30, 2), (37, 53)), ("sphere", lambda: sphere(12, 8, 5), (64, 80))]


def main():
    ref = reference_wrapper()
    out = {}
    for name, make, res in CASES:
        pos, tri = make()
        attr = np.random.default_rng(7).normal(size=(1, pos.shape[1], 5)).astype(np.float32)
        findices, bary = ref.rasterize(torch.from_numpy(pos), torch.from_numpy(tri), res)
        interp = ref.interpolate(torch.from_numpy(attr), findices, bary, torch.from_numpy(tri))
        out.update({name + "_pos": pos, name + "_tri": tri, name + "_attr": attr, name + "_res": np.asarray(res, np.int32),
                    name + "_findices": findices.numpy().astype(np.int32), name + "_bary": bary.numpy().astype(np.float32),
                    name + "_interp": interp.numpy().astype(np.float32)})
        print(name, res, "covered", int((findices > 0).sum()))
    np.savez_compressed(os.path.join(HERE, "mesh_hy_raster.npz"), **out)


if __name__ == "__main__":
    main()
    print("wrote", sorted(os.listdir(HERE)))
