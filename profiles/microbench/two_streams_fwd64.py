"""Experiment (round 6): do the binning chains of one group of views (latency / memory bound) hide under the compositing of another group (VALU bound)?
BASELINE config 2 -- 64 orbit cameras forward-only, 1 M Gaussians, 1080p -- rendered (a) as bench.py does: four groups of 16 views one after the other on one stream, and
(b) as two independent halves of 32 views on two streams (each half = two groups of 16 on its stream, own workspace), nothing between them but the GPU's scheduler.
Existing entry point (c3d_gs_render_views_raw), no library change; the images of (b) are compared bit for bit with (a).
  python profiles/microbench/two_streams_fwd64.py [group [N W H V log_scale_mean]]      (default: 16 1000000 1920 1080 64 log(0.004))"""
import ctypes as C
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "comfyui-3d-pack_amd")]
os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")
import c3d_hip as _h
import diff_gaussian_rasterization as dgr
from c3d_hip import synthetic as S
from c3d_hip.gs_step import FusedViewRender, FusedViewStep

group = int(sys.argv[1]) if len(sys.argv) > 1 else 16
dev = torch.device("cuda", 0)
N, W, H, NV = (int(x) for x in sys.argv[2:6]) if len(sys.argv) > 5 else (1_000_000, 1920, 1080, 64)
lsm = float(sys.argv[6]) if len(sys.argv) > 6 else float(np.log(0.004))
raw = S.make_cloud(N, seed=1234, activated=False, log_scale_mean=lsm)
t = lambda x: torch.as_tensor(np.asarray(x, dtype=np.float32)).to(dev)
sh = t(raw["shs"])
params = [t(raw["means3D"]), sh[:, :1].contiguous(), sh[:, 1:].contiguous(), t(raw["opacities"]), t(raw["scales"]), t(raw["rotations"])]
settings = []
for (r, e, az) in (S.orbit_poses_64() * 4)[:NV]:
    st = S.camera_settings(W, H, 49.1, e, az, r, bg=(1.0, 1.0, 1.0), sh_degree=3)
    settings.append(dgr.GaussianRasterizationSettings(H, W, st["tanfovx"], st["tanfovy"], t(st["bg"]), 1.0, t(st["viewmatrix"]).reshape(4, 4), t(st["projmatrix"]).reshape(4, 4),
                                                      3, t(st["campos"]), False, False))


def enqueue(vr, rs, out):
    """FusedViewRender.run without its host synchronisation (the capacity was fitted by a warm-up call)"""
    V = len(rs)
    keep = []
    views = FusedViewStep._settings(rs, keep, params)
    arr = lambda x: (C.c_void_p * V)(*[x[i].data_ptr() for i in range(V)])
    _h.check(_h.lib().c3d_gs_render_views_raw(views, V, N, *[_h.ptr(p) for p in params], arr(out[0]), arr(out[1]), arr(out[2]), None, vr.capacity, vr.lanes, _h.ptr(vr.workspace),
                                              vr.workspace.numel(), _h.ptr(vr.status), _h.stream(dev)), "c3d_gs_render_views_raw")
    return keep


one = FusedViewRender(N, H, W, dev, group=group)
with torch.no_grad():
    ref = one.run(settings, params)
    ref = one.run(settings, params)
def make(S_, grp):
    per = NV // S_
    rs_ = [FusedViewRender(N, H, W, dev, pair_capacity=one.capacity, group=grp) for _ in range(S_)]
    st_ = [torch.cuda.Stream(dev) for _ in range(S_)]
    f32 = dict(dtype=torch.float32, device=dev)
    ou_ = [(torch.empty((per, 3, H, W), **f32), torch.empty((per, 1, H, W), **f32), torch.empty((per, 1, H, W), **f32)) for _ in range(S_)]

    def fn():
        keep = []
        for h in range(S_):
            with torch.cuda.stream(st_[h]):
                keep.append(enqueue(rs_[h], settings[per * h:per * h + per], ou_[h]))
        return keep
    return fn, ou_, per


f32 = dict(dtype=torch.float32, device=dev)
out1 = (torch.empty((NV, 3, H, W), **f32), torch.empty((NV, 1, H, W), **f32), torch.empty((NV, 1, H, W), **f32))


def seq():
    return enqueue(one, settings, out1)


def timeit(name, fn, reps=5):
    torch.cuda.synchronize()
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    kk = [fn() for _ in range(reps)]
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    print("%-44s %.3f ms per %d views = %.0f Mpx/s" % (name, dt * 1e3, NV, NV * W * H / dt / 1e6), flush=True)


timeit("one stream, groups of %d" % group, seq)
for S_, grp in ((2, 16), (2, 8), (4, 16), (4, 8), (4, 4), (8, 8), (8, 4), (3, 8)):
    if NV % S_ or NV // S_ < 1:
        continue
    fn, ou_, per = make(S_, grp)
    timeit("%d streams, %d views each in groups of %d" % (S_, per, grp), fn)
    timeit("one stream, groups of %d" % group, seq, 3)
    torch.cuda.synchronize()
    same = all(torch.equal(ou_[h][k], ref[k][per * h:per * h + per]) for h in range(S_) for k in range(3))
    if not same:
        print("   IMAGES DIFFER")
    del fn, ou_
    torch.cuda.empty_cache()
print("done")
