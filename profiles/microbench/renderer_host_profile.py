"""Host-side profile of the per-view autograd loop through GaussianSplattingRenderer.render (bench.py --render-path fused): where does the host spend its ~1 ms per view?
  python profiles/microbench/renderer_host_profile.py [verified|unverified]"""
import cProfile
import os
import pstats
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "comfyui-3d-pack_amd")]
os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")
import diff_gaussian_rasterization as dgr
from c3d_hip import synthetic as S
from MVs_Algorithms.GaussianSplatting.main_3DGS_renderer import GaussianSplattingRenderer

mode = sys.argv[1] if len(sys.argv) > 1 else "verified"
dgr.sync_free(mode)
dev = torch.device("cuda", 0)
N, W, H = 1_000_000, 1920, 1080
raw = S.make_cloud(N, seed=1234, activated=False)
r = GaussianSplattingRenderer(sh_degree=3, device=dev)
r.initialize({"xyz": raw["means3D"], "features": raw["shs"], "scaling_raw": raw["scales"], "rotation_raw": raw["rotations"], "opacity_raw": raw["opacities"]})
t = lambda x: torch.as_tensor(np.asarray(x, dtype=np.float32)).to(dev)
cams = []
for (rad, e, az) in S.orbit_poses_64()[:8]:
    st = S.camera_settings(W, H, 49.1, e, az, rad, bg=(1.0, 1.0, 1.0))
    vm, pm, cp = t(st["viewmatrix"]).reshape(4, 4), t(st["projmatrix"]).reshape(4, 4), t(st["campos"])
    cams.append(type("Cam", (), dict(image_height=H, image_width=W, FoVx=2 * np.arctan(st["tanfovx"]), FoVy=2 * np.arctan(st["tanfovy"]), world_view_transform=vm, full_proj_transform=pm, camera_center=cp))())
white = torch.ones(3, device=dev)
tc, ta = torch.rand(3, H, W, device=dev), torch.rand(1, H, W, device=dev)


def step():
    for c in cams:
        out = r.render(c, bg_color=white)
        loss = (out["image"] - tc).abs().mean() * 0.8 + 3.0 * ((out["alpha"] - ta) ** 2).mean()
        (loss / 8).backward()
    for q in (r.gaussians._xyz, r.gaussians._features_dc, r.gaussians._features_rest, r.gaussians._opacity, r.gaussians._scaling, r.gaussians._rotation):
        q.grad = None


for _ in range(3):
    step()
torch.cuda.synchronize()
for reps in (5, 30, 30):
    t0 = time.perf_counter()
    for _ in range(reps):
        step()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print("[%s] %d steps: host enqueue %.3f ms per view, wall %.3f ms per view" % (mode, reps, (t1 - t0) / (8 * reps) * 1e3, (t2 - t0) / (8 * reps) * 1e3))
pr = cProfile.Profile()
pr.enable()
for _ in range(5):
    step()
pr.disable()
torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("cumulative").print_stats(28)
