"""debug (round 5): how many tiles of the multi-view sort passes are processed by a workgroup of another view (work stealing in k_onesweep)?  Needs the `dbg` library variant."""
import ctypes as C
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "comfyui-3d-pack_amd"))
import c3d_hip as _h
from c3d_hip import synthetic as S
from c3d_hip.gs_step import FusedViewStep
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
settings = []
for (rad, e, az) in S.orbit_poses_64()[:V]:
    st = S.camera_settings(W, H, 49.1, e, az, rad, bg=(1.0, 1.0, 1.0), sh_degree=deg)
    settings.append(dgr.GaussianRasterizationSettings(H, W, st["tanfovx"], st["tanfovy"], t(st["bg"]), 1.0, t(st["viewmatrix"]).reshape(4, 4), t(st["projmatrix"]).reshape(4, 4), deg, t(st["campos"]), False, False))
tc = [torch.rand(3, H, W, device=dev) for _ in range(V)]; ta = [torch.rand(1, H, W, device=dev) for _ in range(V)]
o = FusedViewStep(N, H, W, dev, lanes=1, views=V)
grads = [torch.empty_like(q) for q in plist]
lib = _h.lib()
lib.c3d_dbg_sort_steals.argtypes = [C.c_void_p, C.c_int]
for it in range(4):
    lib.c3d_dbg_sort_steals(None, 1)
    o.run(settings, plist, grads, tc, ta, None, w_l1=0.8, w_alpha_mse=3.0, scale=1.0 / V, accumulate=False)
    torch.cuda.synchronize()
    out = (C.c_ulonglong * 8)()
    lib.c3d_dbg_sort_steals(out, 0)
    print("step %d capacity %d: tiles at home %d, stolen %d (by quarter of the victim's tile range: %s), workgroups that found nothing %d, looks %d" % (it, o.capacity, out[0], out[1], list(out[4:8]), out[2], out[3]), flush=True)
