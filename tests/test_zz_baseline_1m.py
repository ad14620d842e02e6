"""The BENCHED kernels under the float64 oracle at BASELINE size.

bench.py's numbers come from the fused multi-view entry points -- c3d_gs_train_views_raw (k_preprocess_views, k_composite_fwd_w<true>,
k_composite_bwd<true, false> with the pixel loss in its prologue, k_bwd_views_geom / k_bwd_views_sh) and c3d_gs_render_views_raw (projection
stream, view lanes) -- not from the drop-in kernels tests/test_gs_hip.py holds to the oracle.  Here those entry points run on the 1,000,000-Gaussian
cloud (seed 1234, SH degree 3, RAW parameters) at 1920 x 1080 on EIGHT of the 64 orbit cameras (two per elevation -30 / 0 / 30 / 60) and are compared
with the float64 build of oracle/gs_oracle.c:

  (b) c3d_gs_render_views_raw: image / depth / alpha of every camera (L1 <= 1e-4), radii (exact up to ceil(3 sigma) on a float32/float64 boundary);
  (a) c3d_gs_train_views_raw, lanes 4 and 1: loss value and the six RAW-parameter gradients against the SUM over the eight views of the float64
      oracle's gradients, chained through exp / sigmoid / normalize in float64;
  (c) the same call with config 3's full loss (mask, 0.8 L1 + 3 MSE(alpha) + 0.2 (1 - MS-SSIM), the HIP MS-SSIM inside the call) against the oracle
      plus the torch restatement of MS-SSIM in float64 on the CPU;
  (d) the tail: a float32 rasterizer and a float64 one take a few discrete decisions differently -- per pixel and splat alpha >= 1/255 and T < 1e-4, per
      Gaussian ceil(3 sigma) and the colour clamp max(0, SH + 0.5).  The fragile set is found two ways: the ORACLE ITSELF run in float32 against its
      float64 build (pixels whose n_contrib differs, Gaussians whose radius or clamp pattern differs), and the kernel's images against the float64
      ones (a flipped decision moves a pixel by far more than rounding noise); pixels within rounding of a kink of the loss (|clamp(C) - t| at C = t,
      render()'s clamp at 0 and 1) join them.  Gaussians blended into a fragile pixel (oracle/gs_oracle.c:
      gs_oracle_taint) or fragile themselves are "tainted"; the test prints how the rows beyond TAIL x the element-wise tolerance split between tainted
      and untainted Gaussians, holds the untainted ones to a hard bound, and replaces the old `hard = 1e9` by numbers.

  (e) round 5: the float32 build of the oracle runs the same eight views forward AND backward through the same loss; its gradients against the float64 build's
      are printed beside the kernel's (relative L2, max-norm, tail rows and how many of them the two tails share), and the kernel's max-norm / relative-L2 error
      must stay within 1.5 x the float32 oracle's own wherever it exceeds 1e-3 (relative L2: everywhere): the tail is what float32 does to this algorithm, not what this implementation adds.

Reference call sites: /root/reference/MVs_Algorithms/GaussianSplatting/main_3DGS.py:158-207 (the training step), shared_utils/camera_utils.py:253-274
(the orbit loop).  The oracle takes ~3.5 s per view (forward + backward) on the GPU box's host cores."""
import os

import numpy as np
import pytest
import torch

from c3d_hip import synthetic as S
from oracle import gs_oracle as O
from helpers import grad_report, hip_settings, oracle_forward

pytestmark = pytest.mark.gpu
N, W, H = 1_000_000, 1920, 1080
POSES = [(-30.0, 45.0), (-30.0, 202.5), (0.0, 0.0), (0.0, 157.5), (30.0, 22.5), (30.0, 270.0), (60.0, 90.0), (60.0, 315.0)]   # (elevation, azimuth) of the 64-camera orbit
RAW_NAMES = ("xyz", "f_dc", "f_rest", "opacity", "scaling", "rotation")
ALPHA_FLIP, COLOR_FLIP = 1e-5, 3e-5          # a pixel whose alpha / colour is off by more than this took a different per-splat decision (rounding noise: ~1e-6)
LOSS_KINK = 3e-6                             # |clamp(C) - target|, |C|, |C - 1| below this: float32 and float64 may sit on different sides of a kink of the loss
TAIL = 30.0                                  # rows beyond TAIL x the element-wise tolerance are "the tail" the test attributes
HARD = 5000.0                                # no entry at all may be off by more than this many tolerances (was 1e9 in round 2; measured: 2210, on a tainted row) ...
UNTAINTED_HARD = 30.0                        # ... and none of a Gaussian no fragile pixel touches by more than this (measured: <= 5.8)


def _chain_to_raw(og, raw):
    """float64 oracle gradients w.r.t. the ACTIVATED tensors -> w.r.t. GaussianModel's raw parameters (main_3DGS_renderer.py:294-321)"""
    op = raw["opacities"].astype(np.float64)
    sg = 1.0 / (1.0 + np.exp(-op))
    r = raw["rotations"].astype(np.float64)
    n = np.maximum(np.linalg.norm(r, axis=1, keepdims=True), 1e-12)
    q = r / n
    g = og["rotations"]
    return {"xyz": og["means3D"], "f_dc": og["shs"][:, :1], "f_rest": og["shs"][:, 1:], "opacity": og["opacities"] * sg * (1 - sg),
            "scaling": og["scales"] * np.exp(raw["scales"].astype(np.float64)), "rotation": (g - q * (q * g).sum(1, keepdims=True)) / n}


def _pixel_loss(oc, oa, tc, ta, scale, w_l1, w_a, mask=None):
    """value and pixel gradients of scale * [w_l1 mean|clamp(C) m - t m| + w_a mean (A - ta)^2] in float64 (main_3DGS.py:184-191)"""
    P = oc.shape[1] * oc.shape[2]
    m = 1.0 if mask is None else mask
    cc = np.clip(oc, 0.0, 1.0)
    d = (cc - tc) * m
    val = scale * (w_l1 * np.abs(d).mean() + w_a * ((oa - ta) ** 2).mean())
    dC = scale * w_l1 * np.sign(d) * m * ((oc >= 0) & (oc <= 1)) / (3.0 * P)
    dA = scale * 2.0 * w_a * (oa - ta) / P
    return val, dC, dA


@pytest.fixture(scope="module")
def full(oracle_built):
    if not torch.cuda.is_available():
        pytest.fail("-m gpu tests need a HIP device (no CPU fallback exists)")
    from c3d_hip.gs_step import FusedViewRender, FusedViewStep
    nt = os.cpu_count() or 8
    raw = S.make_cloud(N, seed=1234, activated=False)
    act = S.make_cloud(N, seed=1234, activated=True)
    dev = lambda a: torch.tensor(np.ascontiguousarray(a), dtype=torch.float32, device="cuda")
    plist = [dev(raw["means3D"]), dev(raw["shs"][:, :1]), dev(raw["shs"][:, 1:]), dev(raw["opacities"]), dev(raw["scales"]), dev(raw["rotations"])]
    sts = [S.camera_settings(W, H, 49.1, el, az, 2.2, bg=(1, 1, 1)) for el, az in POSES]
    rs = [hip_settings(st, "cuda") for st in sts]
    V = len(rs)
    # ---- (b) forward-only entry point: eight lanes, projection stream
    ren = FusedViewRender(N, H, W, "cuda", lanes=2, group=4)
    color, depth, alpha, radii = ren.run(rs, plist, want_radii=True)
    color, depth, alpha, radii = color.cpu().numpy(), depth.cpu().numpy(), alpha.cpu().numpy(), radii.cpu().numpy()
    del ren
    rng = np.random.default_rng(77)
    tcs = [rng.uniform(size=(3, H, W)).astype(np.float32) for _ in range(V)]
    tas = [rng.uniform(size=(1, H, W)).astype(np.float32) for _ in range(V)]
    out = dict(views=[], raw=raw)
    og_sum = None
    og32_sum = None
    og_full = None
    loss_sum, loss_full = 0.0, 0.0
    tainted = np.zeros((N,), bool)
    from shared_utils.msssim import MS_SSIM
    msssim = MS_SSIM(data_range=1, size_average=True, channel=3)
    for v in range(V):
        oc, orad, od, oa, ost = oracle_forward(act, sts[v], dtype=np.float64, nthreads=nt)
        # the oracle against itself: which decisions does float32 take differently?
        oc32, orad32, _, oa32, ost32 = oracle_forward(act, sts[v], dtype=np.float32, nthreads=nt)
        flags32 = ost32.image_state()["n_contrib"].reshape(H, W) != ost.image_state()["n_contrib"].reshape(H, W)
        g32, g64 = ost32.geometry(), ost.geometry()
        frag_g = (orad32 != orad) | ((g32["rgb"] == 0) != (g64["rgb"] == 0)).any(1) | (g32["tiles_touched"] != g64["tiles_touched"])
        # (e) the float32 build of the SAME oracle through the SAME loss, backward included (VERDICT r4, next-round 4 i): what a float32 rasterizer of this
        # algorithm -- the reference's CUDA kernels are float32 too -- leaves against float64, measured instead of argued.  Its loss gradient comes from its own
        # float32 image, as the kernel's does; the per-view gradients are summed in float64 (the kernel sums them in float32: the comparison favours the oracle).
        _, dC32, dA32 = _pixel_loss(oc32.astype(np.float64), oa32[0].astype(np.float64), tcs[v].astype(np.float64), tas[v][0].astype(np.float64), 1.0 / V, 0.8, 3.0)
        og32 = O.backward(ost32, dC32.astype(np.float32), None, dA32.astype(np.float32), nthreads=nt)
        og32 = {k: np.asarray(og32[k], np.float64) for k in ("means3D", "opacities", "shs", "scales", "rotations")}
        og32_sum = og32 if og32_sum is None else {k: og32_sum[k] + og32[k] for k in og32}
        del ost32, g32, g64, og32
        flags = (np.abs(alpha[v, 0] - oa[0]) > ALPHA_FLIP) | (np.abs(color[v] - oc).max(0) > COLOR_FLIP)
        # ... and the LOSS has kinks of its own: |clamp(C) - t| at C = t, and render()'s clamp at 0 and 1.  A pixel within rounding of one of them gets the
        # other one-sided derivative in float32 (with random targets ~1e-6 of the pixel-channels: a hundred per 8 views)
        tc64 = tcs[v].astype(np.float64)
        kink = ((np.abs(np.clip(oc, 0.0, 1.0) - tc64) < LOSS_KINK) | (np.abs(oc) < LOSS_KINK) | (np.abs(oc - 1.0) < LOSS_KINK)).any(0) & (oa[0] > 0)
        n_img, n_o32, n_kink = int(flags.sum()), int(flags32.sum()), int(kink.sum())
        flags |= flags32 | kink
        tainted = O.taint(ost, flags, tainted)
        tainted |= (radii[v] != orad) | frag_g
        out["views"].append(dict(l1_color=float(np.abs(color[v] - oc).mean()), l1_alpha=float(np.abs(alpha[v] - oa).mean()), l1_depth=float(np.abs(depth[v] - od).mean()),
                                 max_color=float(np.abs(color[v] - oc).max()), radii_diff=int((radii[v] != orad).sum()), flagged=n_img, flagged_o32=n_o32, flagged_kink=n_kink, frag_gauss=int(frag_g.sum()),
                                 n_vis=int((orad > 0).sum()), D=int(ost.num_rendered)))
        val, dC, dA = _pixel_loss(oc, oa[0], tcs[v].astype(np.float64), tas[v][0].astype(np.float64), 1.0 / V, 0.8, 3.0)
        loss_sum += val
        og = O.backward(ost, dC, None, dA, nthreads=nt)
        og = {k: og[k] for k in ("means3D", "opacities", "shs", "scales", "rotations")}
        og_sum = og if og_sum is None else {k: og_sum[k] + og[k] for k in og}
        if v < 2:      # (c): config 3's full loss on two views, MS-SSIM through the float64 torch restatement
            tC = torch.tensor(oc, dtype=torch.float64, requires_grad=True)
            tA = torch.tensor(oa, dtype=torch.float64, requires_grad=True)
            m = torch.tensor(tas[v], dtype=torch.float64)
            t = torch.tensor(tcs[v], dtype=torch.float64)
            imgs, refs = tC.clamp(0, 1) * m, t * m
            loss = 0.5 * (0.8 * (imgs - refs).abs().mean() + 3.0 * ((tA - m) ** 2).mean() + 0.2 * (1.0 - msssim(refs[None], imgs[None])))
            gC, gA = torch.autograd.grad(loss, [tC, tA])
            loss_full += float(loss)
            ogf = O.backward(ost, gC.numpy(), None, gA.numpy()[0], nthreads=nt)
            ogf = {k: ogf[k] for k in ("means3D", "opacities", "shs", "scales", "rotations")}
            og_full = ogf if og_full is None else {k: og_full[k] + ogf[k] for k in ogf}
        del ost
    out.update(ref=_chain_to_raw(og_sum, raw), ref_full=_chain_to_raw(og_full, raw), loss_ref=loss_sum, loss_full_ref=loss_full, tainted=tainted,
               ref32=_chain_to_raw(og32_sum, raw))
    # ---- (a) the benched call: c3d_gs_train_views_raw, loss "0.8 L1 + 3 MSE(alpha)" as bench.py's default line, 8 views
    tcd, tad = [dev(t) for t in tcs], [dev(t) for t in tas]
    for lanes in (4, 1):
        step = FusedViewStep(N, H, W, "cuda", lanes=lanes, views=V)
        grads = [torch.empty_like(p) for p in plist]
        lv = step.run(rs, plist, grads, tcd, tad, None, w_l1=0.8, w_alpha_mse=3.0, scale=1.0 / V, accumulate=False)
        lv = step.run(rs, plist, grads, tcd, tad, None, w_l1=0.8, w_alpha_mse=3.0, scale=1.0 / V, accumulate=False)      # second call: fitted capacity, the timed configuration
        out["step%d" % lanes] = (float(lv.item()), [g.cpu().numpy() for g in grads])
        if lanes == 4:   # (c) on the same object
            lv = step.run(rs[:2], plist, grads, tcd[:2], tad[:2], tad[:2], w_l1=0.8, w_alpha_mse=3.0, scale=0.5, accumulate=False, w_ssim=0.2)
            out["full"] = (float(lv.item()), [g.cpu().numpy() for g in grads])
        del step
    return out


def test_render_views_raw_matches_oracle_on_eight_cameras(full):
    for (el, az), r in zip(POSES, full["views"]):
        print("[1M view el %g az %g] L1 colour %.2e alpha %.2e depth %.2e, max colour %.2e, radii differ %d | fragile: %d pixels by image difference, %d by float32-vs-float64 oracle n_contrib, %d on a kink of the loss, %d Gaussians (radius / clamp / tiles) | N_vis %d, D(oracle) %d"
              % (el, az, r["l1_color"], r["l1_alpha"], r["l1_depth"], r["max_color"], r["radii_diff"], r["flagged"], r["flagged_o32"], r["flagged_kink"], r["frag_gauss"], r["n_vis"], r["D"]))
    for r in full["views"]:
        assert r["l1_color"] <= 1e-4 and r["l1_alpha"] <= 1e-4 and r["l1_depth"] <= 1e-4, r
        assert r["radii_diff"] <= 20, r
        assert r["max_color"] <= 2e-2, r
        assert r["flagged"] <= 2e-3 * W * H, r


F32_SLACK = 1.5      # beyond the north star's 1e-3, the kernel's max-norm error against float64 may exceed the float32 ORACLE's own by this factor at most


def _check(name, loss, loss_ref, grads, ref, tainted, ref32=None):
    print("[1M %s] loss %.8f vs float64 oracle %.8f; %d of %d Gaussians are blended into a pixel with a flipped float32/float64 decision" % (name, loss, loss_ref, int(tainted.sum()), N))
    assert abs(loss - loss_ref) <= 2e-5 * max(1.0, abs(loss_ref))
    for k, g in zip(RAW_NAMES, grads):
        want = np.asarray(ref[k], np.float64).reshape(N, -1)
        got = np.asarray(g, np.float64).reshape(N, -1)
        r = grad_report(got, want)
        mx = np.abs(want).max()
        tol = 1e-3 * np.abs(want) + 1e-6 * mx
        ratio = np.abs(got - want) / tol
        bad_rows = (ratio > TAIL).any(1)
        r_un = float(ratio[~tainted].max()) if (~tainted).any() else 0.0
        frac_un = float((ratio[~tainted] > 1.0).mean()) if (~tainted).any() else 0.0
        print("[1M %s] %-8s relL2 %.2e, outside tol %.2e of entries (worst %.0f x), max-norm %.2e | rows beyond %g x tol: %d, of them tainted: %d | untainted Gaussians: worst %.1f x tol, %.2e outside"
              % (name, k, r["rel_l2"], r["frac_viol"], r["worst"], r["max_norm"], TAIL, int(bad_rows.sum()), int((bad_rows & tainted).sum()), r_un, frac_un))
        if ref32 is not None:      # the float32 build of the oracle against its float64 build, same metric, side by side
            o32 = np.asarray(ref32[k], np.float64).reshape(N, -1)
            r32 = grad_report(o32, want)
            ratio32 = np.abs(o32 - want) / tol
            bad32 = (ratio32 > TAIL).any(1)
            both = int((bad_rows & bad32).sum())
            print("[1M %s] %-8s   float32 ORACLE vs float64: relL2 %.2e, outside tol %.2e (worst %.0f x), max-norm %.2e, rows beyond %g x tol: %d (%d of them also in the kernel's tail) | kernel / float32-oracle: relL2 %.2f x, max-norm %.2f x"
                  % (name, k, r32["rel_l2"], r32["frac_viol"], r32["worst"], r32["max_norm"], TAIL, int(bad32.sum()), both,
                     r["rel_l2"] / max(r32["rel_l2"], 1e-30), r["max_norm"] / max(r32["max_norm"], 1e-30)))
            # wherever the kernel's max-norm error is beyond the north star's 1e-3, the float32 oracle's own must be within a factor 1.5 of it (measured, MI355X: xyz
            # 0.67 x, f_dc / f_rest / opacity 1.00 x -- the same rows, scaling 0.71 x; rotation 1.75 x of 2.8e-4: both far below 1e-3, hence the max())
            assert r["max_norm"] <= max(F32_SLACK * r32["max_norm"], 1e-3), (k, r["max_norm"], r32["max_norm"])
            assert r["rel_l2"] <= F32_SLACK * r32["rel_l2"] + 1e-5, (k, r["rel_l2"], r32["rel_l2"])
        assert r["rel_l2"] <= 1e-3, (k, r)
        assert r["frac_viol"] <= 3e-3, (k, r)
        assert r["worst"] <= HARD, (k, r)
        assert r_un <= UNTAINTED_HARD, (k, r_un)
        assert int(bad_rows.sum()) <= 5e-4 * N, (k, int(bad_rows.sum()))
        # the tail IS the tainted Gaussians (a few rows of slack for a decision class the fragile sets do not model)
        assert int((bad_rows & ~tainted).sum()) <= max(3, int(0.03 * bad_rows.sum())), (k, int((bad_rows & tainted).sum()), int(bad_rows.sum()), float(tainted.mean()))


@pytest.mark.parametrize("lanes", [4, 1])
def test_train_views_raw_gradients_match_sum_of_float64_oracle_views(full, lanes):
    loss, grads = full["step%d" % lanes]
    _check("train_views lanes %d" % lanes, loss, full["loss_ref"], grads, full["ref"], full["tainted"], full["ref32"])


def test_train_views_raw_full_loss_with_msssim_matches_oracle_plus_torch(full):
    loss, grads = full["full"]
    _check("full loss (MS-SSIM inside)", loss, full["loss_full_ref"], grads, full["ref_full"], full["tainted"])


def test_drop_in_raw_entry_point_sync_free_matches_the_float64_oracle_on_eight_cameras(full):
    """the drop-in path at BASELINE size.  One autograd call per view through diff_gaussian_rasterization.rasterize_gaussians_raw -- after the shape's first
    call that is c3d_gs_forward_raw_nosync (hint-sized launches, the count word in pinned memory) + c3d_gs_backward_raw --, the loss of bench.py's default line spelled in torch,
    gradients accumulated over the eight views by autograd: same check, same bounds as the fused entry point."""
    import diff_gaussian_rasterization as dgr
    raw = full["raw"]
    dev = lambda a: torch.tensor(np.ascontiguousarray(a), dtype=torch.float32, device="cuda")
    params = [dev(raw["means3D"]), dev(raw["shs"][:, :1]), dev(raw["shs"][:, 1:]), dev(raw["opacities"]), dev(raw["scales"]), dev(raw["rotations"])]
    for p in params:
        p.requires_grad_(True)
    rs = [hip_settings(S.camera_settings(W, H, 49.1, el, az, 2.2, bg=(1, 1, 1)), "cuda") for el, az in POSES]
    V = len(rs)
    rng = np.random.default_rng(77)      # the fixture's targets
    tcs = [rng.uniform(size=(3, H, W)).astype(np.float32) for _ in range(V)]
    tas = [rng.uniform(size=(1, H, W)).astype(np.float32) for _ in range(V)]
    with torch.no_grad():                # the shape's first call takes the synchronous path and learns the count
        dgr.rasterize_gaussians_raw(params[0], None, params[1], params[2], params[3], params[4], params[5], rs[0])
    calls0 = dgr.sync_free_calls
    loss_total = 0.0
    for v in range(V):
        m2d = torch.zeros_like(params[0], requires_grad=True)
        color, radii, depth, alpha = dgr.rasterize_gaussians_raw(params[0], m2d, params[1], params[2], params[3], params[4], params[5], rs[v])
        loss = (0.8 * (color.clamp(0, 1) - dev(tcs[v])).abs().mean() + 3.0 * ((alpha - dev(tas[v])) ** 2).mean()) / V
        loss.backward()
        loss_total += float(loss.item())
    assert dgr.sync_free_calls - calls0 == V      # every differentiated call went through the sync-free entry point
    dgr.flush()
    _check("drop-in raw, sync-free", loss_total, full["loss_ref"], [p.grad.cpu().numpy() for p in params], full["ref"], full["tainted"], full["ref32"])


def test_render_views_raw_matches_oracle_forward_on_all_64_orbit_cameras(oracle_built):
    """BASELINE config 2 IS the 64 orbit cameras; the fixture above holds eight of them forward and backward.  Here all 64 go through
    c3d_gs_render_views_raw (groups of 16 on four HIP streams, as bench.py's forward target runs them) and every image, alpha, depth and radius is held to one float64 pass of the oracle
    (forward only: ~1.5 s per camera on the box's host cores)."""
    if not torch.cuda.is_available():
        pytest.fail("-m gpu tests need a HIP device (no CPU fallback exists)")
    from c3d_hip.gs_step import FusedViewRender
    nt = os.cpu_count() or 8
    raw = S.make_cloud(N, seed=1234, activated=False)
    act = S.make_cloud(N, seed=1234, activated=True)
    dev = lambda a: torch.tensor(np.ascontiguousarray(a), dtype=torch.float32, device="cuda")
    plist = [dev(raw["means3D"]), dev(raw["shs"][:, :1]), dev(raw["shs"][:, 1:]), dev(raw["opacities"]), dev(raw["scales"]), dev(raw["rotations"])]
    poses = S.orbit_poses_64()
    assert len(poses) == 64
    sts = [S.camera_settings(W, H, 49.1, e, az, r, bg=(1, 1, 1)) for (r, e, az) in poses]
    ren = FusedViewRender(N, H, W, "cuda", lanes=1, group=16, streams=4)      # as bench.py's forward target and render_views run them: four parts of 16 views on four HIP streams
    rs = [hip_settings(st, "cuda") for st in sts]
    color, depth, alpha, radii = ren.run(rs, plist, want_radii=True)
    assert ren._plan(len(rs)) == (4, 16)
    del ren
    one = FusedViewRender(N, H, W, "cuda", lanes=1, group=16)                 # ... and the same bits as the one-stream call
    for a_, b_ in zip(one.run(rs, plist, want_radii=True), (color, depth, alpha, radii)):
        assert torch.equal(a_, b_)
    del one, a_, b_
    color, depth, alpha, radii = color.cpu().numpy(), depth.cpu().numpy(), alpha.cpu().numpy(), radii.cpu().numpy()
    worst = dict(l1=0.0, a=0.0, d=0.0, mx=0.0, rad=0)
    for v, st in enumerate(sts):
        oc, orad, od, oa, ost = oracle_forward(act, st, dtype=np.float64, nthreads=nt)
        del ost
        l1, la, ld, mx, rd = float(np.abs(color[v] - oc).mean()), float(np.abs(alpha[v] - oa).mean()), float(np.abs(depth[v] - od).mean()), float(np.abs(color[v] - oc).max()), int((radii[v] != orad).sum())
        worst = dict(l1=max(worst["l1"], l1), a=max(worst["a"], la), d=max(worst["d"], ld), mx=max(worst["mx"], mx), rad=max(worst["rad"], rd))
        assert l1 <= 1e-4 and la <= 1e-4 and ld <= 1e-4, (poses[v], l1, la, ld)
        # single pixels where float32 and float64 take a per-splat decision differently (alpha >= 1/255, T < 1e-4) move by that splat's whole contribution: bounded in number, not in size
        flipped = int((np.abs(color[v] - oc).max(0) > COLOR_FLIP).sum())
        assert rd <= 20 and flipped <= 2e-3 * W * H, (poses[v], rd, flipped, mx)      # (measured over the 64 cameras: <= 3 radii, <= 406 of 2 M pixels, the largest single pixel 0.13)
    print("[1M forward, 64 cameras] worst L1 colour %.2e alpha %.2e depth %.2e, worst max colour %.2e, most radii differing in a view %d" % (worst["l1"], worst["a"], worst["d"], worst["mx"], worst["rad"]))
