"""Where does the host time of a DiffRastMesh step go?  (bench.py --workload mesh is host bound: 4.2 ms to enqueue 8 views against 3.2 ms of kernels.)
cProfile over a few steps of the same loop as bench.main_mesh; prints the top entries by cumulative and by own time."""
import cProfile
import io
import os
import pstats
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "comfyui-3d-pack_amd")]
from c3d_hip import synthetic as S
from mesh_processer.mesh import Mesh
from MVs_Algorithms.DiffRastMesh.diff_mesh_renderer import DiffRastRenderer
from shared_utils.camera_utils import OrbitCamera, orbit_camera

dev = torch.device("cuda:0")
H = W = 1024
v, f, vt, vn = S.make_uv_sphere(500, 500, radius=0.7, displacement=0.05)
t = lambda x, dt=torch.float32: torch.tensor(x, dtype=dt, device=dev)
mesh = Mesh(v=t(v), f=t(f, torch.int32), vt=t(vt), ft=t(f, torch.int32), device=dev)
mesh.auto_normal()
mesh.albedo = torch.sigmoid(torch.randn((1024, 1024, 3), generator=torch.Generator(device="cpu").manual_seed(7))).to(dev)
r = DiffRastRenderer(mesh, True).to(dev)
r.train_geo = True
cam = OrbitCamera(W, H, fovy=49.1)
poses = [orbit_camera(e, az, 2.0) for e in (-20.0, 20.0) for az in np.arange(16) * 22.5][:8]
with torch.no_grad():
    targets = [r.render(p, cam.perspective, H, W)["image"].clone() * 0.9 for p in poses]


def step():
    for p, tg in zip(poses, targets):
        out = r.render(p, cam.perspective, H, W)
        loss = ((out["image"] - tg) ** 2).mean() + 0.1 * ((out["alpha"] - 0.5) ** 2).mean()
        (loss / 8).backward()
    r.raw_albedo.grad = None; r.v_offsets.grad = None


for _ in range(3):
    step()
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(5):
    step()
pr.disable()
torch.cuda.synchronize()
for key in ("cumulative", "tottime"):
    s = io.StringIO()
    pstats.Stats(pr, stream=s).sort_stats(key).print_stats(22)
    print("\n".join(line[:170] for line in s.getvalue().splitlines()[4:34]))
