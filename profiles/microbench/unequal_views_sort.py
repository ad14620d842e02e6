"""Round 5: do unequal views serialise the tile sort now that a workgroup's home view is its XCD?  Eight forward views of the BASELINE cloud in ONE launch per stage:
(a) eight equal views (radius 2.2), (b) eight views whose pair counts differ by several times (radii 1.4 ... 5), (c) equal views again at the radius whose pair count is
(b)'s mean.  If stealing works, (b)'s tile sort costs about what (c)'s does; without it, what its LARGEST view would cost alone on one XCD (about (largest / mean) times (c))."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "comfyui-3d-pack_amd"))
import c3d_hip as _h
from c3d_hip import synthetic as S
from c3d_hip.gs_step import FusedViewRender
import diff_gaussian_rasterization as dgr
dev = torch.device("cuda", 0)
N, W, H, deg, V = 1_000_000, 1920, 1080, 3, 8
raw = S.make_cloud(N, seed=1234, sh_degree=deg, activated=False)
from MVs_Algorithms.GaussianSplatting.main_3DGS_renderer import GaussianSplattingRenderer
r = GaussianSplattingRenderer(sh_degree=deg, device=dev)
r.initialize({"xyz": raw["means3D"], "features": raw["shs"], "scaling_raw": raw["scales"], "rotation_raw": raw["rotations"], "opacity_raw": raw["opacities"]})
g = r.gaussians
plist = [q.detach() for q in (g._xyz, g._features_dc, g._features_rest, g._opacity, g._scaling, g._rotation)]
t = lambda x: torch.as_tensor(np.asarray(x, dtype=np.float32)).to(dev)


def settings(radii):
    out = []
    for i, rad in enumerate(radii):
        st = S.camera_settings(W, H, 49.1, -30.0, 22.5 * i, rad, bg=(1.0, 1.0, 1.0), sh_degree=deg)
        out.append(dgr.GaussianRasterizationSettings(H, W, st["tanfovx"], st["tanfovy"], t(st["bg"]), 1.0, t(st["viewmatrix"]).reshape(4, 4), t(st["projmatrix"]).reshape(4, 4), deg, t(st["campos"]), False, False))
    return out


def pairs(rs):
    with torch.no_grad():
        dgr.GaussianRasterizer(rs)(means3D=plist[0], means2D=None, opacities=torch.sigmoid(plist[3]), shs=torch.cat([plist[1], plist[2]], 1), scales=torch.exp(plist[4]),
                                   rotations=torch.nn.functional.normalize(plist[5]))
    dgr.flush()
    return int(dgr.last_num_rendered)


def run(name, radii, cap):
    st = settings(radii)
    D = [pairs(s) for s in st]
    vr = FusedViewRender(N, H, W, dev, lanes=1, group=8, pair_capacity=cap)
    vr._fitted = True
    with torch.no_grad():
        for _ in range(3):
            vr.run(st, plist)
        torch.cuda.synchronize()
        _h.prof_enable(True)
        for _ in range(10):
            vr.run(st, plist)
        torch.cuda.synchronize()
    p = _h.prof_read(); _h.prof_enable(False)
    ms = {k: v[0] / v[1] for k, v in p.items()}
    print("%-44s pairs per view (M): %s  mean %.2f max %.2f | per 8-view launch: tile sort %.3f ms, emit %.3f, depth sort %.3f, compositing %.3f" %
          (name, " ".join("%.1f" % (d / 1e6) for d in D), np.mean(D) / 1e6, max(D) / 1e6, ms["gs_tile_sort"], ms["gs_emit"], ms["gs_depth_sort"], ms["gs_composite_fwd"]), flush=True)
    return np.mean(D)


cap = 16_000_000
run("(a) eight equal views, radius 2.2", [2.2] * 8, cap)
m = run("(b) unequal: radii 1.4 ... 5", [1.4, 5.0, 1.6, 4.0, 1.8, 3.0, 2.2, 2.6], cap)
for rad in (1.9, 2.0, 2.1):
    run("(c) eight equal views, radius %.1f" % rad, [rad] * 8, cap)
