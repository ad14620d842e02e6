"""Timing of the wider nvdiffrast surface on one MI355X: mip-mapped texture fetch vs the plain linear fetch, the pyramid
build, and depth peeling, at the sizes of BASELINE config 5 (1024x1024 pixels, 499k-triangle sphere) with a 2048^2 x 3 texture minified
~3 texels / pixel.  Prints one JSON object; run under `rocprofv3 --kernel-trace` for the per-kernel table
(profiles/r01k_mesh_ext_kernel_stats.csv).  Usage (repo root):  python profiles/microbench/mesh_ext_timing.py [--iters 20]
"""
import argparse
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "comfyui-3d-pack_amd")]


def timed(fn, iters, warmup=3):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=20)
    a = ap.parse_args()
    import nvdiffrast.torch as dr
    from c3d_hip import synthetic as S
    dev = "cuda"
    H = W = 1024
    Ht = Wt = 2048
    g = torch.Generator(device="cpu").manual_seed(3)
    tex = torch.randn((1, Ht, Wt, 3), generator=g).to(dev).requires_grad_(True)
    ys, xs = torch.meshgrid(torch.arange(H, dtype=torch.float32), torch.arange(W, dtype=torch.float32), indexing="ij")
    c, s_ = np.cos(0.3), np.sin(0.3)
    k = 3.0 / Wt                                                     # 3 texels per pixel -> level ~1.58
    uv = torch.stack([(c * xs - s_ * ys) * k, (s_ * xs + c * ys) * k], -1)[None].to(dev).requires_grad_(True)
    da = torch.tensor([c * k, -s_ * k, s_ * k, c * k], dtype=torch.float32).expand(1, H, W, 4).contiguous().to(dev)
    gy = torch.randn((1, H, W, 3), generator=g).to(dev)
    res = {}

    def fwd_bwd(f):
        def run():
            tex.grad = None; uv.grad = None
            (f() * gy).sum().backward()
        return run
    with torch.no_grad():
        res["texture_linear_fwd_ms"] = timed(lambda: dr.texture(tex, uv, filter_mode="linear"), a.iters)
        mip = dr.texture_construct_mip(tex)
        res["mip_build_fwd_ms"] = timed(lambda: dr.texture_construct_mip(tex), a.iters)
        res["texture_mip_fwd_prebuilt_ms"] = timed(lambda: dr.texture(tex, uv, uv_da=da, mip=mip), a.iters)
        res["texture_mip_fwd_incl_build_ms"] = timed(lambda: dr.texture(tex, uv, uv_da=da), a.iters)
    res["texture_linear_fwd_bwd_ms"] = timed(fwd_bwd(lambda: dr.texture(tex, uv, filter_mode="linear")), a.iters)
    res["texture_mip_fwd_bwd_incl_build_ms"] = timed(fwd_bwd(lambda: dr.texture(tex, uv, uv_da=da)), a.iters)
    # coarse levels: every pixel of a tile hits the same few texels (bias 6 -> 32x32 level)
    bias = torch.full((1, H, W), 6.0, device=dev)
    res["texture_mip_coarse_fwd_bwd_incl_build_ms"] = timed(fwd_bwd(lambda: dr.texture(tex, uv, mip_level_bias=bias)), a.iters)

    v, f, vt, vn = S.make_uv_sphere(500, 500, radius=0.7, displacement=0.05)
    pos, _, _ = S.mesh_clip_positions(v, -20.0, 35.0, 2.0, W, H)
    tpos = torch.tensor(pos, dtype=torch.float32, device=dev)
    ttri = torch.tensor(f, dtype=torch.int32, device=dev)
    ctx = dr.RasterizeCudaContext()
    with torch.no_grad():
        res["rasterize_ms"] = timed(lambda: dr.rasterize(ctx, tpos, ttri, (H, W)), a.iters)

        def peel(n):
            def run():
                with dr.DepthPeeler(ctx, tpos, ttri, (H, W)) as p:
                    for _ in range(n):
                        p.rasterize_next_layer()
            return run
        res["depth_peel_1_layer_ms"] = timed(peel(1), a.iters)
        res["depth_peel_4_layers_ms"] = timed(peel(4), a.iters)
        with dr.DepthPeeler(ctx, tpos, ttri, (H, W)) as p:
            cov = [int((p.rasterize_next_layer()[0][..., 3] > 0).sum()) for _ in range(4)]
    res["depth_peel_covered_pixels"] = cov
    res = {k_: (round(v_, 4) if isinstance(v_, float) else v_) for k_, v_ in res.items()}
    res["config"] = {"pixels": [H, W], "texture": [Ht, Wt, 3], "texels_per_pixel": 3.0, "triangles": int(f.shape[0]), "iters": a.iters}
    print(json.dumps(res))


if __name__ == "__main__":
    main()
