"""Experiment (round 5): does an 8-view fused step gain from running as two (or four) half-steps on concurrent HIP streams?
The V-wide chain runs its ~20 launches one after the other; the compositing kernels are VALU-bound, the sorts / scans latency-bound, so kernels of
different half-steps could share the CUs.  Same BASELINE workload as `bench.py` (1 M Gaussians, SH 3, 1920x1080, 8 views per step, L1 + alpha loss);
every variant writes complete gradients (the halves into their own buffers, summed on the main stream inside the timed region).
usage: python profiles/microbench/two_streams_step.py [steps]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "comfyui-3d-pack_amd"))
import c3d_hip  # noqa: E402
from c3d_hip import synthetic as S  # noqa: E402
from c3d_hip.gs_step import FusedViewStep  # noqa: E402
from c3d_hip.parallel import FlatGrads  # noqa: E402
import diff_gaussian_rasterization as dgr  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
SSIM = float(os.environ.get("C3D_EXP_SSIM", "0"))      # round 6: 0.2 = BASELINE config 3's loss (masked, + 0.2 (1 - MS-SSIM)) in the upper block; the joined block is skipped then
dev = torch.device("cuda", 0)
N, W, H, deg, V = 1_000_000, 1920, 1080, 3, 8
raw = S.make_cloud(N, seed=1234, sh_degree=deg, activated=False)
from MVs_Algorithms.GaussianSplatting.main_3DGS_renderer import GaussianSplattingRenderer  # noqa: E402
renderer = GaussianSplattingRenderer(sh_degree=deg, device=dev)
renderer.initialize({"xyz": raw["means3D"], "features": raw["shs"], "scaling_raw": raw["scales"], "rotation_raw": raw["rotations"], "opacity_raw": raw["opacities"]})
gm = renderer.gaussians
plist = [q.detach() for q in (gm._xyz, gm._features_dc, gm._features_rest, gm._opacity, gm._scaling, gm._rotation)]
t = lambda x: torch.as_tensor(np.asarray(x, dtype=np.float32)).to(dev)
settings = []
for (r, e, az) in S.orbit_poses_64()[:V]:
    st = S.camera_settings(W, H, 49.1, e, az, r, bg=(1.0, 1.0, 1.0), sh_degree=deg)
    settings.append(dgr.GaussianRasterizationSettings(H, W, st["tanfovx"], st["tanfovy"], t(st["bg"]), 1.0, t(st["viewmatrix"]).reshape(4, 4),
                                                      t(st["projmatrix"]).reshape(4, 4), deg, t(st["campos"]), False, False))
tc = [torch.rand(3, H, W, device=dev) for _ in range(V)]
ta = [torch.rand(1, H, W, device=dev) for _ in range(V)]


def variant(parts, concurrent):
    """parts: view counts of the half-steps"""
    objs, flats, streams, ranges = [], [], [], []
    v0 = 0
    for p in parts:
        o = FusedViewStep(N, H, W, dev, lanes=1, views=p)
        o.defer_status = True
        objs.append(o)
        flats.append(FlatGrads([torch.empty_like(q).requires_grad_(False) for q in plist]))
        streams.append(torch.cuda.Stream(dev) if concurrent else None)
        ranges.append((v0, v0 + p))
        v0 += p
    main = torch.cuda.current_stream(dev)

    def step():
        if concurrent:
            e = torch.cuda.Event()
            e.record(main)
        for o, f, s, (a, b) in zip(objs, flats, streams, ranges):
            def go():
                o.run(settings[a:b], plist, f.views, tc[a:b], ta[a:b], (ta[a:b] if SSIM else None), w_l1=0.8, w_alpha_mse=3.0, scale=1.0 / V, accumulate=False, w_ssim=SSIM)
            if s is None:
                go()
            else:
                s.wait_event(e)
                with torch.cuda.stream(s):
                    go()
        if concurrent:
            for s in streams:
                main.wait_stream(s)
        for f in flats[1:]:
            flats[0].flat.add_(f.flat)
    for _ in range(4):
        step()
    for o in objs:
        o.finish()
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    for o in objs:
        o.finish()
    torch.cuda.synchronize(dev)
    ms = (time.perf_counter() - t0) / steps * 1e3
    del objs, flats
    torch.cuda.empty_cache()
    return ms


for rep in range(2):
    for name, parts, conc in (("one 8-view step", [8], False), ("4 + 4 views, one stream", [4, 4], False), ("4 + 4 views, two streams", [4, 4], True),
                              ("2+2+2+2 views, four streams", [2, 2, 2, 2], True), ("5 + 3 views, two streams", [5, 3], True)):
        ms = variant(parts, conc)
        print("%-32s %.3f ms per step  %.0f Mpx/s" % (name, ms, V * W * H / ms / 1e3), flush=True)


if SSIM:
    sys.exit(0)


def joined(parts, concurrent=True, steps_=None):
    """the half-steps write the slices of ONE 8-view workspace (each call on its own stream, per-Gaussian pass left out: accumulate flag 2), the streams join,
    and ONE per-Gaussian pass over all views runs on the main stream -- only existing entry points"""
    import ctypes as C
    import c3d_hip as _h
    lib = _h.lib()
    o = FusedViewStep(N, H, W, dev, lanes=1, views=V)
    f = FlatGrads([torch.empty_like(q) for q in plist])
    for _ in range(2):
        o.run(settings, plist, f.views, tc, ta, None, w_l1=0.8, w_alpha_mse=3.0, scale=1.0 / V, accumulate=False)      # fits the capacity
    cap = o.capacity
    vs = lib.c3d_gs_step_workspace_bytes(N, H, W, cap, 1) - lib.c3d_msssim_workspace_bytes(1, 3, H, W)
    assert o.workspace.numel() >= V * vs
    keep = []
    views = o._settings(settings, keep, plist)
    words = torch.zeros(4 * len(parts), dtype=torch.int32, device=dev)
    streams = [torch.cuda.Stream(dev) if concurrent else None for _ in parts]
    main = torch.cuda.current_stream(dev)
    pp = [_h.ptr(_h.f32c(q)) for q in plist]
    gp = [_h.ptr(g) for g in f.views]
    loss = _h.GsLoss(0.8, 0.0, 3.0, 1.0 / V, 0.0)
    tcp = (C.c_void_p * V)(*[x.data_ptr() for x in tc])
    tap = (C.c_void_p * V)(*[x.data_ptr() for x in ta])
    vsz = C.sizeof(_h.GsSettings)

    def step():
        words.zero_()
        if concurrent:
            e = torch.cuda.Event(); e.record(main)
        v0 = 0
        for i, (p, s) in enumerate(zip(parts, streams)):
            if s is not None:
                s.wait_event(e)
            with torch.cuda.stream(s if s is not None else main):
                sub = C.cast(C.addressof(views) + v0 * vsz, C.POINTER(_h.GsSettings))
                _h.check(lib.c3d_gs_train_views_raw(sub, p, N, *pp, C.cast(C.addressof(tcp) + 8 * v0, C.POINTER(C.c_void_p)), C.cast(C.addressof(tap) + 8 * v0, C.POINTER(C.c_void_p)), None,
                                                    C.byref(loss), *gp, words[4 * i + 2:].data_ptr(), cap, 1, 2, o.workspace.data_ptr() + v0 * vs, words[4 * i:].data_ptr(),
                                                    _h.stream(dev)), "train_views")
            v0 += p
        if concurrent:
            for s in streams:
                main.wait_stream(s)
        _h.check(lib.c3d_gs_step_param_backward_range(views, V, N, pp[0], pp[1], pp[2], pp[4], pp[5], *gp, cap, 0, o.workspace.data_ptr(), 0, N, _h.stream(dev)), "a8")
    for _ in range(4):
        step()
    torch.cuda.synchronize(dev)
    st = words.tolist()
    assert all(st[4 * i] == 0 for i in range(len(parts))), st
    ref = f.flat.clone()
    o.run(settings, plist, f.views, tc, ta, None, w_l1=0.8, w_alpha_mse=3.0, scale=1.0 / V, accumulate=False)
    o.finish(); torch.cuda.synchronize(dev)
    same = bool((ref.view(torch.int32) == f.flat.view(torch.int32)).all())
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize(dev)
    return (time.perf_counter() - t0) / steps * 1e3, same


for rep in range(2):
    for name, parts, conc in (("joined: [8], one stream", [8], False), ("joined: 4 + 4, one stream", [4, 4], False), ("joined: 4 + 4, two streams", [4, 4], True),
                              ("joined: 3 + 3 + 2, three streams", [3, 3, 2], True), ("joined: 2+2+2+2, four streams", [2, 2, 2, 2], True)):
        ms, same = joined(parts, conc)
        print("%-34s %.3f ms per step  %.0f Mpx/s   gradients bit-identical to the one-call step: %s" % (name, ms, V * W * H / ms / 1e3, same), flush=True)
