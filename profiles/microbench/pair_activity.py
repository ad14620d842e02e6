"""How many (quadrant, splat) pairs of the BASELINE view does the backward compositing kernel walk?  Reads the activity bytes the recording forward
pass leaves in the binning buffer (gs_internal.h: gs_pair_activity) -- telemetry for DESIGN.md section 4d, not a test."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "comfyui-3d-pack_amd"), os.path.join(ROOT, "tests")]
import diff_gaussian_rasterization as dgr
from c3d_hip import synthetic as S
from helpers import hip_settings

N, W, H = 1_000_000, 1920, 1080
dev = "cuda"
from MVs_Algorithms.GaussianSplatting.main_3DGS_renderer import GaussianSplattingRenderer
raw = S.make_cloud(N, seed=1234, sh_degree=3, activated=False)
renderer = GaussianSplattingRenderer(sh_degree=3, device=dev)
renderer.initialize({"xyz": raw["means3D"], "features": raw["shs"], "scaling_raw": raw["scales"], "rotation_raw": raw["rotations"], "opacity_raw": raw["opacities"]})
r, e, az = S.orbit_poses_64()[0]
st = S.camera_settings(W, H, 49.1, e, az, r, bg=(1.0, 1.0, 1.0), sh_degree=3)
t = lambda x: torch.as_tensor(np.asarray(x, dtype=np.float32)).to(dev)
rs = dgr.GaussianRasterizationSettings(H, W, st["tanfovx"], st["tanfovy"], t(st["bg"]), 1.0, t(st["viewmatrix"]).reshape(4, 4), t(st["projmatrix"]).reshape(4, 4), 3, t(st["campos"]), False, False)
cam = type("Cam", (), dict(image_height=H, image_width=W, FoVx=2 * np.arctan(st["tanfovx"]), FoVy=2 * np.arctan(st["tanfovy"]), world_view_transform=rs.viewmatrix,
                           full_proj_transform=rs.projmatrix, camera_center=rs.campos))()
out = renderer.render(cam, bg_color=torch.ones(3, device=dev))
fn = out["image"].grad_fn
while fn is not None and not hasattr(fn, "saved_tensors"):
    fn = fn.next_functions[0][0]
node = fn
seen = set()
stack = [out["image"].grad_fn]
binning = None
while stack and binning is None:
    f_ = stack.pop()
    if f_ is None or id(f_) in seen:
        continue
    seen.add(id(f_))
    if "Rasterize" in type(f_).__name__:
        binning = f_.saved_tensors[-2]
        node_saved = f_.saved_tensors
        break
    stack.extend(x[0] for x in f_.next_functions)
assert binning is not None
D = dgr.last_num_rendered
img = None
for f_ in [x for x in [fn] if x is not None]:
    pass
# binning buffer layout (gs_internal.h: gs_carve_binning): tkey[0] | tkey[1] | tval[0] | tval[1] | ranges[tiles] | ...; image buffer: final_T[P] | n_contrib[P]
d_al = (4 * D + 255) // 256 * 256
keys = [binning[i * d_al:i * d_al + 4 * D].view(torch.int32).cpu().numpy() for i in range(2)]
res = 0 if np.all(np.diff(keys[0]) >= 0) else 1
assert np.all(np.diff(keys[res]) >= 0)
act_all = binning[(1 - res) * d_al:(1 - res) * d_al + D].cpu().numpy()
gx, gy = (W + 15) // 16, (H + 15) // 16
rng = binning[4 * d_al:4 * d_al + 8 * gx * gy].view(torch.int32).cpu().numpy().reshape(-1, 2)
image_buf = node_saved[-1]
ncon = image_buf[(4 * W * H + 255) // 256 * 256:][:4 * W * H].view(torch.int32).cpu().numpy().reshape(H, W)
pad = np.zeros((gy * 16, gx * 16), np.int32); pad[:H, :W] = ncon
upto = pad.reshape(gy, 16, gx, 16).max(axis=(1, 3)).reshape(-1)
mask = np.zeros(D, bool)
for t_ in range(gx * gy):
    mask[rng[t_, 0]:rng[t_, 0] + min(upto[t_], rng[t_, 1] - rng[t_, 0])] = True
act = np.where(mask, act_all, 0)
assert act.max() < 16
pop = np.array([bin(i).count("1") for i in range(16)])[act]
print("pairs D = %d; reached by some pixel of their tile (k < upto): %d; with >= 1 blended quadrant: %d (%.1f %% of D); blended (quadrant, splat) pairs: %d = %.2f per pair with any" % (
    D, int(mask.sum()), int((act > 0).sum()), 100.0 * (act > 0).mean(), int(pop.sum()), pop.sum() / max((act > 0).sum(), 1)))
print("histogram of blended quadrants per pair:", np.bincount(pop, minlength=5).tolist())
