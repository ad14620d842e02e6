"""Regenerates tests/golden/*.npz from the oracle (float64 run, stored as float32 inputs + float64-derived
outputs rounded to float32).  The reference ships no golden vectors for this path, so
these freeze OUR oracle: they catch drift, they do not pin parity with the un-vendored CUDA wheel.
Run from the repo root:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "comfyui-3d-pack_amd")]
from c3d_hip import synthetic as S  # noqa: E402
from oracle import gs_oracle as O  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))


def gs_small():
    sc = S.make_small_scene(N=96, seed=21, scale=0.06)
    st = S.camera_settings(64, 48, 49.1, -25, 40, 2.0, bg=(1.0, 1.0, 1.0))
    rng = np.random.default_rng(99)
    gC = rng.normal(size=(3, 48, 64)).astype(np.float32)
    gD = rng.normal(size=(1, 48, 64)).astype(np.float32)
    gA = rng.normal(size=(1, 48, 64)).astype(np.float32)
    kw = dict(shs=sc["shs"], scales=sc["scales"], rotations=sc["rotations"])
    color, radii, depth, alpha, state = O.forward(sc["means3D"], sc["opacities"], st, dtype=np.float64, **kw)
    g = O.backward(state, gC, gD, gA)
    out = dict(sc)
    out.update({"st_" + k: np.asarray(v) for k, v in st.items() if k not in ("prefiltered", "debug")})
    out.update(color=color.astype(np.float32), depth=depth.astype(np.float32), alpha=alpha.astype(np.float32), radii=radii,
               gC=gC, gD=gD, gA=gA)
    out.update({"grad_" + k: g[k].astype(np.float32) for k in ("means3D", "opacities", "shs", "scales", "rotations", "means2D")})
    np.savez_compressed(os.path.join(HERE, "gs_small.npz"), **out)


if __name__ == "__main__":
    gs_small()
    print("wrote", os.listdir(HERE))
