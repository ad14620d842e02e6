#!/usr/bin/env python
"""bench.py -- Mpixels/s of the 3DGS forward+backward hot path on BASELINE.json's workload.

  python bench.py --gpus N --steps K --warmup W        (N > 1 without a launcher: starts its own N ranks under torch.distributed.run, 127.0.0.1, a free port)
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

Workload (config.workload): BASELINE.md section 3 configs 2-4 -- 1,000,000 synthetic Gaussians (seed 1234), SH degree 3,
1920x1080, the 64-camera orbit.  A "step" = one pass of the hot path over one batch of views on every rank:
`--views-per-gpu` (default 8) views rasterized forward and backward through the C-ABI -- by default in ONE fused library call
(`--render-path step`: c3d_gs_train_views_raw, the views of a step through every stage in ONE launch -- view = a grid dimension, groups of
`--group` views (default 16), `--lanes` groups in flight (default 1) -- pixel loss and its gradient inside, one per-Gaussian backward pass
for all views); `--render-path boundary | fused | accessor` time the plain drop-in API one autograd call
per view instead.  For N > 1 the step ends with the one gradient exchange of the shared-Gaussian training loop (`--exchange allreduce`,
default, or `allgather`: every rank's dense gradient + fixed-order local sum).  Per-GPU work is fixed as N grows ("weak"): at N = 8
this is config 4 (64 views/step, 8 per GPU).  Inputs are resident in HBM before the timed region.
`--mode fwd` (config 2: forward only, c3d_gs_render_views_raw) and `--mode train` (config 3: + fused Adam) change what a step contains
and say so in `metric`; `--workload mesh` runs BASELINE config 5 (DiffRastMesh).

value = views * W * H over all ranks / wall seconds / 1e6  (wall = max over ranks, barrier + synchronize on both sides).
roofline = the dominant kernel group (largest share of in-library GPU time, measured with HIP events on the launch streams),
algorithmic bytes per launch (DESIGN.md "Algorithmic bytes": per view x the views one launch covers) / its average duration.  Only with
`--lanes` > 1 (several groups in flight; not the default) do kernels of different groups share the CUs inside the timed region; the duration
used is then that of an extra single-group pass over the same step, the in-region figures go to `kernels_concurrent_avg_ms` /
`roofline.avg_ms_concurrent`.  `roofline.traffic` (PMC) and
`roofline.issue` (SQ counters) come from the newest committed rocprofv3 summaries under profiles/.
cpu_baseline = the CPU oracle (a port: the reference has no CPU path) on ONE view of the same workload.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for _p in (ROOT, os.path.join(ROOT, "comfyui-3d-pack_amd")):
    if _p not in sys.path:
        sys.path.insert(0, _p)
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
# kernel arguments in device memory: a ROCm runtime switch, read when the runtime initialises.  Every line of this bench is a chain of dependent launches; with it
# the dispatch of each is ~2 us shorter: 8-view step 5.99 -> 5.90 ms, mesh step 1.81 -> 1.77 ms, the node-default training run 1299 -> 1435 it/s (same box,
# profiles/r03/r03u_kernarg.txt).  c3d_hip sets the same default for every user of the package; an explicit HIP_FORCE_DEV_KERNARG=0 in the environment wins.
os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")

import numpy as np
import torch

HBM_PEAK_GBPS = 8000.0   # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy ceiling)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)      # (the first fused step fits the pair capacity and reallocates the workspace: keep it and its successor out of the timed region)
    ap.add_argument("--mode", choices=["fwdbwd", "fwd", "train"], default="fwdbwd")
    ap.add_argument("--views-per-gpu", type=int, default=8)
    ap.add_argument("--gaussians", type=int, default=1_000_000)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--sh-degree", type=int, default=3)
    ap.add_argument("--cpu-baseline", choices=["auto", "on", "off"], default="auto")
    ap.add_argument("--exchange", choices=["allgather", "allreduce", "zero1"], default="allreduce",
                    help="gradient exchange of the shared-Gaussian step: one RCCL all-reduce (default: 2*(n-1)/n * 236 MB per rank on the wire), the literal all-gather of every rank's gradient + rank-ordered local sum (7 * 236 MB received per rank at n = 8), or zero1 (--mode train only): "
                         "all-to-all reduce-scatter in rank order -> Adam on the owned 1/n -> all-gather of the parameters (c3d_hip.parallel.ZeroOneAdam)")
    ap.add_argument("--exchange-chunks", type=int, default=4,
                    help="N > 1, --exchange allreduce: Gaussian ranges of the per-Gaussian backward pass, each range's collective overlapping the next range's kernels (1 = one all-reduce after the step)")
    ap.add_argument("--timed-prof", choices=["on", "off"], default="on", help="HIP-event kernel timing inside the timed region")
    ap.add_argument("--lanes", type=int, default=0, help="view GROUPS a fused call splits its views into (ceil(views / lanes) views per launch of every stage, the groups one after the "
                                                       "other on the same stream; the library owns no streams); 0 = 1: all views of the step through every stage in ONE launch each, in every mode")
    ap.add_argument("--as-rank", type=int, default=-1, help="one-GPU runs: render the cameras rank R of an N-GPU run would (rank r = orbit cameras [r * views-per-gpu, (r + 1) * views-per-gpu): "
                                                            "ranks of an 8-GPU run are elevation bands); default: this process's own rank")
    ap.add_argument("--group", type=int, default=16, help="--mode fwd: views per launch of every stage (<= 16)")
    ap.add_argument("--streams", type=int, default=4, help="--mode fwd (render path step): HIP streams the views of a call are spread over, one library call per stream (FusedViewRender: one part's "
                                                          "binning chain runs underneath another part's compositing); 1 = one call on one stream")
    ap.add_argument("--sync-free", choices=["on", "verified", "unverified", "off"], default="on",
                    help="--render-path boundary | fused | accessor: how the drop-in rasterizer call learns its pair count (diff_gaussian_rasterization.sync_free).  on = verified (product "
                         "default): the whole forward enqueued at once, the host waits for the count word only and renders a view that did not fit again -- always exact; unverified: no "
                         "wait at all; off = the wheel's behaviour: project, read the count back, enqueue the second half")
    ap.add_argument("--forward-only", choices=["on", "off"], default="on", help="--render-path boundary | fused | accessor, calls that are not differentiated (--mode fwd): render with "
                                                                                "C3D_GS_FLAG_FORWARD_ONLY (no pair-activity record, no record-base scan, no final_T / n_contrib stores); off = the A/B partner")
    ap.add_argument("--inference-mode", choices=["on", "off"], default="off", help="--mode fwd on the drop-in API: run the steps under torch.inference_mode() (an inference caller)")
    ap.add_argument("--render-path", choices=["step", "fused", "accessor", "boundary"], default="step",
                    help="step: c3d_gs_train_views_raw, all views of the step forward+loss+backward in one sync-free library call (product default for training); fused: GaussianSplattingRenderer.render with activations folded into the kernels (product default); accessor: the same "
                         "API through the reference's op-by-op accessors; boundary: bare diff_gaussian_rasterization call on pre-activated leaves")
    ap.add_argument("--defer-status", choices=["on", "off"], default="on",
                    help="fused step: examine a step's overflow / fault words when the next step starts instead of waiting for them (what the trainer does; the last "
                         "step is examined before the timed region ends).  off: one host synchronisation per step")
    ap.add_argument("--targets", choices=["on", "off"], default="on",
                    help="N = 1, fwdbwd / train: after the timed region also run BASELINE config 2 (forward only, all 64 orbit cameras) and report the north star's forward-raster roofline figure in `targets`")
    ap.add_argument("--workload", choices=["gs", "mesh", "ref-default"], default="gs",
                    help="gs = BASELINE configs 2-4 (the metric); mesh = config 5 (DiffRastMesh); ref-default = the reference node's OWN default training run "
                         "(/root/reference/nodes.py:1175-1198: 10,000 initial Gaussians, batch 1, every default of GSParams incl. densification from step 500), --ref-res square images: it/s")
    ap.add_argument("--ref-res", type=int, default=512, help="--workload ref-default: reference image size (the node takes what it is given: 512 and 1024 are typical)")
    ap.add_argument("--loss", choices=["auto", "l1alpha", "full", "full-torch"], default="auto",
                    help="pixel loss of the step path: l1alpha = 0.8 L1 + 3 MSE(alpha) inside c3d_gs_train_views_raw; full = BASELINE config 3's loss, the reference's default "
                         "(main_3DGS.py:184-192): + 0.2 (1 - MS-SSIM), masked by the target alpha, through c3d_gs_forward_views_raw -> torch -> c3d_gs_backward_views_raw.  "
                         "auto: full for --mode train, l1alpha for --mode fwdbwd; full = the MS-SSIM term by the fused HIP kernels inside the same library call, full-torch = the step "
                         "split at the image with torch's op chain for the loss (what round 2 started from: 65 ms of MS-SSIM per step)")
    return ap.parse_args()


def algorithmic_bytes(N, K, P, n_vis, D):
    """SURVEY.md 8(d) per-unit figures, split per kernel group (DESIGN.md 'Algorithmic bytes')."""
    return {
        "gs_preprocess": N * (44 + 12 * K) + 48 * n_vis,
        "gs_depth_sort": 8 * n_vis,                       # one (key,id) read of the visible set; passes are overhead
        "gs_emit": 12 * D + 48 * n_vis * 0,               # key/value emit 12 B per pair
        "gs_tile_sort": 12 * D,                           # sorted read 12 B per pair; passes are overhead
        "gs_composite_fwd": 44 * D + 20 * P,              # per-tile splat gather 44 + rgb/depth/alpha stores 20
        "gs_composite_bwd": 48 * D + 32 * P + 48 * n_vis, # gather 44 + id 4; pixel grads 20 + aux 12; gradient record
        "gs_preprocess_bwd": 48 * n_vis + 2 * N * (44 + 12 * K),
        "adam": 28 * N * (11 + 3 * K),                    # p,g,m,v read + p,m,v write
    }


def code_digest():
    import c3d_hip
    return c3d_hip.code_digest()


def load_calibration():
    """newest committed profiles/*_pmc_calibration.json (profiles/microbench/pmc_calib.hip under rocprofv3): counter bytes / known bytes per access pattern"""
    try:
        cand = sorted(f for f in os.listdir(os.path.join(ROOT, "profiles")) if f.endswith("_pmc_calibration.json"))
        return (json.load(open(os.path.join(ROOT, "profiles", cand[-1]))), cand[-1]) if cand else ({}, None)
    except Exception:
        return {}, None


# which calibration pattern describes a kernel's reads / writes (profiles/microbench/pmc_calib.hip); None = the guide's x2 for reads, raw for writes
PMC_PATTERN = {"k_composite_bwd": ("k_cal_gather64", "k_cal_record_write48"), "k_composite_fwd_w": ("k_cal_gather64", "k_cal_stream_write"),
               "k_preprocess_views_r": ("k_cal_stream_read", "k_cal_stream_write"),
               "k_bwd_views_geom": ("k_cal_gather64", "k_cal_stream_write"), "k_bwd_views_sh": ("k_cal_stream_read", "k_cal_stream_write"),
               "k_emit": ("k_cal_gather64", "k_cal_stream_write"), "k_adam": ("k_cal_stream_read", "k_cal_stream_write")}


def calibrated_bytes(rec, kernel_name, cal):
    """PMC record of one kernel (fetch_MB_raw, write_MB) -> (bytes, note): raw counters divided by the measured counter/known ratio of the kernel's access pattern"""
    base = kernel_name.split("<")[0].strip()
    pat = PMC_PATTERN.get(base)
    fr, wr = rec["fetch_MB_raw"] * 1e6, rec["write_MB"] * 1e6
    if cal and pat and cal.get(pat[0], {}).get("fetch_over_known") and cal.get(pat[1], {}).get("write_over_known"):
        ff, wf = cal[pat[0]]["fetch_over_known"], cal[pat[1]]["write_over_known"]
        return fr / ff + wr / wf, "FETCH_SIZE / %.3f (%s) + WRITE_SIZE / %.3f (%s)" % (ff, pat[0], wf, pat[1])
    return 2.0 * fr + wr, "2 x FETCH_SIZE (MI355X_MICROARCH.md: wide streaming reads are tallied at half) + WRITE_SIZE raw; no calibration file for this pattern"


def load_profile_json(suffix, exclude=None):
    """newest committed profiles/*<suffix> whose `_meta.code_digest` is the digest of the code that is running -> (dict, file name, stale digest | None).
    Traffic measured on other kernel code is refused: the line then says traffic: null, stale: <digest>."""
    try:
        cand = sorted(f for f in os.listdir(os.path.join(ROOT, "profiles")) if f.endswith(suffix) and not (exclude and exclude in f))
        if not cand:
            return {}, None, None
        d = json.load(open(os.path.join(ROOT, "profiles", cand[-1])))
        dig = (d.get("_meta") or {}).get("code_digest")
        if dig != code_digest():
            return {}, cand[-1], dig or "unstamped"
        return d, cand[-1], None
    except Exception:
        return {}, None, None


def main_ref_default(a, world, rank, dev, dist):
    """The reference's default 3DGS training workload: Gaussian_Splatting_3D with every node default -- 10,000 random-ball
    Gaussians, SH degree 3, batch 1, loss 0.8 L1 + 3 MSE(alpha) + 0.2 (1 - MS-SSIM), white/black background per view, densify every 100 steps from step
    500, opacity reset at 3000 -- through this repo's mirror of GaussianSplatting3D.training (the fused one-call step).  Targets: renders of a denser
    synthetic cloud from 8 orbit poses.  A step is launch bound at this size (~150 launches, ~10-30 us kernels): the line reports it/s and what the host
    needs to enqueue a step next to the GPU time of the step's kernels."""
    import c3d_hip
    from c3d_hip import synthetic as S
    from c3d_hip.gs_step import FusedViewRender
    import diff_gaussian_rasterization as dgr
    from MVs_Algorithms.GaussianSplatting.main_3DGS import GaussianSplatting3D, GSParams
    R = a.ref_res
    t = lambda x: torch.as_tensor(np.asarray(x, dtype=np.float32)).to(dev)
    np.random.seed(0); torch.manual_seed(0)
    poses = [(1.75, float(e), float(az), 0.0, 0.0, 0.0) for e in (-15, 20) for az in (0, 90, 180, 270)]
    tgt = S.make_cloud(100_000, seed=4321, log_scale_mean=float(np.log(0.012)), radius=0.5, activated=False)
    rs = []
    for (r_, e_, az_, *_c) in poses:
        st = S.camera_settings(R, R, 49.1, e_, az_, r_, bg=(1.0, 1.0, 1.0))
        rs.append(dgr.GaussianRasterizationSettings(R, R, st["tanfovx"], st["tanfovy"], t(st["bg"]), 1.0, t(st["viewmatrix"]).reshape(4, 4), t(st["projmatrix"]).reshape(4, 4), 3, t(st["campos"]), False, False))
    tp = [t(tgt["means3D"]), t(tgt["shs"][:, :1]), t(tgt["shs"][:, 1:]), t(tgt["opacities"]), t(tgt["scales"]), t(tgt["rotations"])]
    with torch.no_grad():
        color, _, alpha, _ = FusedViewRender(100_000, R, R, dev, lanes=1, group=4).run(rs, tp)
    refs = [color[i].clamp(0, 1).permute(1, 2, 0).contiguous() for i in range(len(poses))]          # node layout: [H, W, 3]
    masks = [(alpha[i, 0] > 0.5).float() for i in range(len(poses))]
    gp = GSParams()                                                # every default of the node
    tr = GaussianSplatting3D(gp, None, device=dev)
    tr.fused_densify_stats = os.environ.get("C3D_BENCH_TORCH_DENSIFY_STATS") != "1"      # A/B hook of this bench only (profiles/r05o_*): the statistics as torch ops, as before round 5
    tr.prepare_training(refs, masks, poses, 49.1)
    import random
    rng = random.Random(0)
    n0 = tr.renderer.gaussians._xyz.shape[0]
    host = []

    def run(first, count):
        for step in range(first, first + count):
            tr.training_step(step, [rng.randint(0, len(poses) - 1) for _ in range(gp.batch_size)])
            if tr._step is not None and hasattr(tr._step, "last_host_ms"):
                host.append(tr._step.last_host_ms)
    run(0, a.warmup)
    torch.cuda.synchronize(dev)
    host.clear()
    # The timed region carries NO per-kernel event timing (round 6): an iteration is ~33 launches of 4-80 us, and the two event records around each of its ~17 kernel groups
    # cost ~10 us of GPU time per group -- a third of the iteration (profiles/r06/r06j_ref_default_iteration_timeline.txt: 10-11 us in front of every group's first kernel, 0-1 us
    # between the kernels of one group).  The per-kernel table comes from a short pass AFTER the timed region, as the mesh line does it.
    t0 = time.perf_counter()
    run(a.warmup, a.steps)
    if tr._step is not None:
        tr._step.finish()
    torch.cuda.synchronize(dev)
    dt = time.perf_counter() - t0
    host_timed = host[:]                      # (the pass below appends to the list run() closes over)
    prof, prof_steps = {}, min(a.steps, 100)
    if a.timed_prof == "on":
        c3d_hip.prof_enable(True)
        run(a.warmup + a.steps, prof_steps)
        if tr._step is not None:
            tr._step.finish()
        torch.cuda.synchronize(dev)
        prof = {k: (ms * a.steps / prof_steps, n * a.steps // prof_steps) for k, (ms, n) in c3d_hip.prof_read().items()}      # scaled to the timed region's step count
        c3d_hip.prof_enable(False)
    n1 = tr.renderer.gaussians._xyz.shape[0]
    kern_ms = sum(ms for ms, _ in prof.values()) / max(a.steps, 1)
    out = {"metric": "it/s, the reference node's default 3DGS training run (10k initial Gaussians, batch 1, %dx%d)" % (R, R), "value": round(a.steps / dt, 2), "unit": "it/s",
           "n_gpus": 1, "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(dt / a.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
           "dtype": "f32", "data": "synthetic",
           "config": {"workload": "GaussianSplatting3D.training with the node defaults (nodes.py:1175-1198): %d -> %d Gaussians over steps %d..%d, SH 3, batch 1, %dx%d, 8 reference views, "
                                  "loss 0.8 L1 + 3 MSE(alpha) + 0.2 (1 - MS-SSIM), densify from 500 every 100" % (n0, n1, a.warmup, a.warmup + a.steps, R, R),
                      "host_enqueue_ms_per_step": round(float(np.mean(host_timed)), 4) if host_timed else None, "kernel_ms_per_step": round(kern_ms, 4) if prof else None,
                      "step_over_kernel_time": round(dt / a.steps * 1e3 / kern_ms, 2) if prof and kern_ms > 0 else None, "points_start": n0, "points_end": n1},
           "roofline": None, "cpu_baseline": None, "kernels": {k: {"ms_per_step": round(ms / a.steps, 4), "launches_per_step": round(n / a.steps, 1)} for k, (ms, n) in prof.items()},
           "code_digest": code_digest()}
    print(json.dumps(out))


def main_mesh(a, world, rank, dev, dist, emit=True):
    """BASELINE config 5: 499,000-triangle displaced lat-long sphere, 1024^2 albedo, 1024x1024, 32 cameras (elev {-20,20} x 16
    azimuths, radius 2.0); one step = `--views-per-gpu` views of DiffRastRenderer.render forward + backward w.r.t. raw_albedo
    and v_offsets (rasterize + 2x antialias + 3x interpolate + texture + the torch elementwise ops around them)."""
    import c3d_hip
    from c3d_hip import synthetic as S
    from mesh_processer.mesh import Mesh
    from MVs_Algorithms.DiffRastMesh.diff_mesh_renderer import DiffRastRenderer
    from shared_utils.camera_utils import OrbitCamera, orbit_camera
    H = W = 1024
    v, f, vt, vn = S.make_uv_sphere(500, 500, radius=0.7, displacement=0.05)
    t = lambda x, dt=torch.float32: torch.tensor(x, dtype=dt, device=dev)
    mesh = Mesh(v=t(v), f=t(f, torch.int32), vt=t(vt), ft=t(f, torch.int32), device=dev)
    mesh.auto_normal()
    g = torch.Generator(device="cpu").manual_seed(7)
    mesh.albedo = torch.sigmoid(torch.randn((1024, 1024, 3), generator=g)).to(dev)
    r = DiffRastRenderer(mesh, True).to(dev)
    r.train_geo = True
    cam = OrbitCamera(W, H, fovy=49.1)
    poses = [orbit_camera(e, az, 2.0) for e in (-20.0, 20.0) for az in np.arange(16) * 22.5]
    mine = [poses[(rank * a.views_per_gpu + i) % len(poses)] for i in range(a.views_per_gpu)]
    targets = []
    with torch.no_grad():
        for p in mine:
            targets.append(r.render(p, cam.perspective, H, W)["image"].clone() * 0.9)

    import torch.nn.functional as F
    half = torch.full((H, W, 1), 0.5, device=dev)
    seed_grad = torch.tensor(1.0 / (a.views_per_gpu * world), device=dev)      # d(step loss) / d(view loss): the 1 / views factor without a division kernel per view
    # --render-path step (default): the whole step -- render, image loss (MSE against the target images), backward of every view, gradients summed -- as ONE
    # library call (c3d_mesh_train_views, every stage one launch over all views; what DiffMesh.training_step uses on a HIP device); fused: one autograd call per view, loss in torch
    use_step = a.render_path == "step"
    if use_step:
        from c3d_hip.mesh_step import FusedMeshStep
        mstep = FusedMeshStep(dev, lanes=1)      # round 4: the views of a step go through every stage in one launch; the library ignores `lanes`
        proj32 = cam.perspective.astype(np.float32)
        sviews = [((proj32 @ np.linalg.inv(p.astype(np.float32)).astype(np.float32)).astype(np.float32), (1.0, 1.0, 1.0)) for p in mine]
        tg_chw = [tg.permute(2, 0, 1).contiguous() for tg in targets]
        f32i, ft32i, vt32 = mesh.f.to(torch.int32).contiguous(), mesh.ft.to(torch.int32).contiguous(), mesh.vt.to(torch.float32).contiguous()
        d_ra, d_vo = torch.empty_like(r.raw_albedo), torch.empty_like(r.v_offsets)

    def step():
        if use_step:
            mstep.run(sviews, mesh.v, r.v_offsets, f32i, vt32, ft32i, r.raw_albedo, r.glctx, tg_chw, None, d_ra, d_vo, H, W, w_mse=1.0, w_ssim=0.0,
                      scale=1.0 / (a.views_per_gpu * world), accumulate=False)
            if world > 1:
                dist.all_reduce(d_ra); dist.all_reduce(d_vo)
            return
        # the step is host bound (~45 launches per view at 5-8 us each against 0.40 ms of kernels): the image loss is spelled with the library ops the
        # reference's trainer uses (F.mse_loss, diff_mesh.py:121) instead of sub / pow / mean chains -- 8 launches fewer per view, the same arithmetic
        for p, tg in zip(mine, targets):
            out = r.render(p, cam.perspective, H, W)
            loss = F.mse_loss(out["image"], tg) + 0.1 * F.mse_loss(out["alpha"], half)
            loss.backward(seed_grad)
        if world > 1:
            for q in (r.raw_albedo, r.v_offsets):
                dist.all_reduce(q.grad)
        r.raw_albedo.grad = None; r.v_offsets.grad = None

    def sync():
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier(); torch.cuda.synchronize(dev)
    for _ in range(a.warmup):
        step()
    sync()
    # inside the timed region only the dominant group is event-timed (every timed launch costs two event records on a host-bound step: timing all
    # nine groups cost 15 %); the per-group table comes from a separate pass right after it
    dom_groups = ["mesh_texture_bwd", "mesh_rasterize_bwd", "mesh_ras_tri"]       # the candidates for the dominant group + the dominant kernel itself (k_ras_tri, a slot of its own)
    c3d_hip.prof_enable(a.timed_prof == "on", only=dom_groups)
    t0 = time.perf_counter()
    for _ in range(a.steps):
        step()
    t_enq = time.perf_counter() - t0          # host time to enqueue the timed steps (nothing in a mesh step waits for the GPU)
    sync()
    dt = time.perf_counter() - t0
    prof_dom = c3d_hip.prof_read() if a.timed_prof == "on" else {}
    c3d_hip.prof_enable(False)
    prof = {}
    concurrent = False      # (rounds 2-3 ran the views on concurrent lanes and needed a separate single-lane pass for per-kernel durations)
    if a.timed_prof == "on":
        if concurrent:
            keep_step, mstep = mstep, FusedMeshStep(dev, lanes=1)
            step(); sync()
        c3d_hip.prof_enable(True)
        for _ in range(min(a.steps, 3)):
            step()
        sync()
        prof = {k: (ms * a.steps / min(a.steps, 3), n * a.steps // min(a.steps, 3)) for k, (ms, n) in c3d_hip.prof_read().items()}     # scaled to the timed region's step count
        c3d_hip.prof_enable(False)
        if concurrent:
            mstep = keep_step
        else:
            for g in dom_groups:
                if prof_dom.get(g, (0, 0))[1]:
                    prof[g] = prof_dom[g]              # the dominant groups: as measured inside the timed region
    if world > 1:
        tt = torch.tensor([dt], device=dev, dtype=torch.float64); dist.all_reduce(tt, op=dist.ReduceOp.MAX); dt = float(tt.item())
    P, V, T = H * W, v.shape[0], f.shape[0]
    # SURVEY 8(d) mesh formula, per view and per kernel group (launch counts differ: see 'launches')
    alg = {"mesh_rasterize": 16 * V + 12 * T + 32 * P, "mesh_interpolate": (32 + 24) * P, "mesh_texture": (8 + 12) * P,
           "mesh_antialias": (16 + 4 * 2 + 4 * 2) * P,
           "mesh_rasterize_bwd": 32 * P + 16 * V, "mesh_interpolate_bwd": (16 + 24 + 16) * P + 24 * V, "mesh_texture_bwd": (8 + 12 + 8) * P + 12 * 1024 * 1024,
           "mesh_antialias_bwd": (16 + 16 + 16) * P + 16 * V}
    vpl = min(a.views_per_gpu, 16) if use_step else 1      # views per launch
    ras_tri = prof.pop("mesh_ras_tri", None)      # k_ras_tri alone (a slot nested inside the rasterize group): the step's longest kernel
    kern = {k: {"avg_ms": round(ms / n, 4), "launches": n, "ms_per_view": round(ms / (a.steps * a.views_per_gpu), 4), "views_per_launch": vpl} for k, (ms, n) in prof.items()}
    dom = max(prof, key=lambda k: prof[k][0]) if prof else None
    roof = None
    roof_kernel = None
    if ras_tri and ras_tri[1]:
        # per LAUNCH of the dominant kernel: compulsory bytes of k_ras_tri = the vertices (16 V) and index triples (12 T) it reads and one 8-byte
        # depth | id word per pixel (the rest of SURVEY 8(d)'s 32 P of the rasterize op -- rast, rast_db -- belongs to the resolve / pixel pass), x the views of a launch
        avg_ms = ras_tri[0] / ras_tri[1]
        bpl = (16 * V + 12 * T + 8 * P) * vpl
        roof_kernel = {"bound": "hbm", "kernel": "k_ras_tri", "achieved": round(bpl / (avg_ms * 1e-3) / 1e9, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                       "frac": round(bpl / (avg_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4), "traffic": None, "avg_ms": round(avg_ms, 4), "alg_bytes_per_launch": int(bpl),
                       "views_per_launch": vpl, "measured": "HIP events around the kernel on the launch stream, %d launches" % ras_tri[1]}
        pmc_k, pmc_k_file, stale_k = load_profile_json("_mesh_pmc_traffic.json")
        rec = (pmc_k.get("k_ras_tri") or {}).get("long")
        if rec and vpl > 1:
            roof_kernel["traffic"] = int((rec["fetch_MB_raw"] + rec["write_MB"]) * 1e6)
            roof_kernel["traffic_note"] = "FETCH_SIZE (raw) + WRITE_SIZE per LONG launch of k_ras_tri (the step's launches over %d views; %d of them), profiles/%s" % (vpl, rec["launches"], pmc_k_file)
        elif stale_k:
            roof_kernel["stale"] = "profiles/%s was measured on code %s" % (pmc_k_file, stale_k)
    if dom:
        per_view_ms = prof[dom][0] / (a.steps * a.views_per_gpu)
        ach = alg.get(dom, 0) / (per_view_ms * 1e-3) / 1e9
        roof = {"bound": "hbm", "kernel": dom, "achieved": round(ach, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBPS, 4),
                "traffic": None, "avg_ms": round(per_view_ms, 4), "alg_bytes_per_launch": int(alg.get(dom, 0)), "note": "per view: the group's time in the timed region / views (a launch covers %d views; the group is several kernels)" % vpl}
        if concurrent:
            roof["measured"] = "single-lane pass after the timed region (kernels run alone); timed region used %d view lanes" % mstep.lanes
            if prof_dom.get(dom, (0, 0))[1]:
                roof["avg_ms_concurrent"] = round(prof_dom[dom][0] / (a.steps * a.views_per_gpu), 4)
        pmc, pmc_file, stale = load_profile_json("_mesh_pmc_traffic.json")
        grp = (pmc.get("_groups") or {}).get(dom)
        if grp and grp.get("long") and use_step:
            roof["traffic"] = int((grp["long"]["fetch_MB_raw"] + grp["long"]["write_MB"]) * 1e6 / vpl)
            roof["traffic_note"] = ("FETCH_SIZE (raw) + WRITE_SIZE of the group's kernels per LONG launch (the step's launches over %d views) / %d views, profiles/%s" % (vpl, vpl, pmc_file))
        elif stale:
            roof["stale"] = "profiles/%s was measured on code %s" % (pmc_file, stale)
    # whole-chain figure by SURVEY 8(d)'s op-level mesh formula: B_mesh_fwd = 16 V + 12 T + 220 P, backward = 2 x the image-space terms + 24 V + 12 Ht Wt
    Ht = Wt = 1024
    b_view = (16 * V + 12 * T + 220 * P) + (2 * 220 * P + 24 * V + 12 * Ht * Wt)
    per_view_s = dt / max(a.steps * a.views_per_gpu, 1)
    chain = {"what": "SURVEY 8(d) op-level compulsory bytes of one mesh view (forward + backward: what the UNFUSED op graph must move) / wall time per view in the timed region",
             "bytes_per_view": int(b_view), "ms_per_view": round(per_view_s * 1e3, 4), "achieved": round(b_view / per_view_s / 1e9, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
             "frac": round(b_view / per_view_s / 1e9 / HBM_PEAK_GBPS, 4)} if per_view_s > 0 else None
    cpu = None
    if rank == 0 and world == 1 and a.cpu_baseline != "off":
        cpu = mesh_cpu_baseline(v, f, vt, H, W)
    line = None
    if rank == 0:
        line = ({"metric": "Mpixels/s DiffRastMesh forward+backward @500k triangles 1024x1024", "value": round(a.views_per_gpu * world * a.steps * P / dt / 1e6, 2),
                          "unit": "Mpixels/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(dt / a.steps * 1e3, 3),
                          "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
                          "config": {"workload": "DiffRastMesh fwd+bwd, %d-triangle displaced sphere, 1024^2 albedo, 1024x1024, %d views/GPU/step" % (T, a.views_per_gpu),
                                     "render_path": ("step: c3d_mesh_train_views, %d views per launch" % vpl) if use_step else "fused: one autograd call per view",
                                     "parallelism": "view-parallel dp%d" % world, "host_enqueue_ms_per_step": round(t_enq / a.steps * 1e3, 3),
                                     "dist_backend": (dist.get_backend() if dist is not None else None), "rccl_ranks": (dist.get_world_size() if dist is not None else 1)},
                          "roofline": roof_kernel or roof, "roofline_group": (roof if roof_kernel else None), "roofline_chain": chain, "cpu_baseline": cpu, "kernels": kern,
                          "code_digest": code_digest()})
        if emit:
            print(json.dumps(line))
    if world > 1 and emit:
        dist.destroy_process_group()
    return line


def mesh_cpu_baseline(v, f, vt, H, W):
    """the mesh oracle (a port: nvdiffrast has no CPU path) on ONE view of config 5: rasterize -> interpolate -> texture -> 2x antialias, forward and backward"""
    try:
        from oracle import mesh_oracle as MO
        from c3d_hip import synthetic as S
        MO.build()
        pos, _, _ = S.mesh_clip_positions(v, -20.0, 0.0, 2.0, W, H)
        rng = np.random.default_rng(1)
        tex = rng.normal(size=(1, 1024, 1024, 3)).astype(np.float32)
        t1 = time.perf_counter()
        rast, db = MO.rasterize(pos, f, (H, W))
        texc, _ = MO.interpolate(vt[None], rast, f, db, "all")
        col = MO.texture(tex, texc)
        aa = MO.antialias(col, rast, pos, f)
        al = MO.antialias(np.clip(rast[..., 3:], 0, 1), rast, pos, f)
        dcol, dpos = MO.antialias_bwd(col, rast, pos, f, np.ones_like(aa))
        MO.antialias_bwd(np.clip(rast[..., 3:], 0, 1), rast, pos, f, np.ones_like(al))
        dtex, duv = MO.texture_bwd(tex, texc, dcol)
        dvt, drast = MO.interpolate_bwd(vt[None], rast, f, duv)
        MO.rasterize_bwd(pos, f, rast, drast)
        tc = time.perf_counter() - t1
        return {"value": round(H * W / tc / 1e6, 4), "unit": "Mpixels/s", "cores": os.cpu_count() or 1, "kind": "port",
                "sample": "1 view of config 5 (%d triangles, %dx%d) through the CPU mesh oracle, forward + backward, %.1f s (OpenMP loops over pixels / triangles)" % (f.shape[0], W, H, tc)}
    except Exception as ex:
        return {"value": None, "unit": "Mpixels/s", "cores": 0, "kind": "port", "sample": "failed: %r" % (ex,)}


def self_launch(n):
    """`python bench.py --gpus N` without a launcher: start N ranks of this same command line, one per GPU, under torch.distributed.run on a free
    loopback port (what the driver's torchrun form does), pass rank 0's JSON line through and return the launcher's exit code."""
    import socket
    import subprocess
    with socket.socket() as s_:
        s_.bind(("127.0.0.1", 0))
        port = s_.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("OMP_NUM_THREADS", "8")          # torchrun would set 1 (and say so on stderr); the CPU-baseline leg does not run at N > 1
    sys.exit(subprocess.call(cmd, env=env))


def main():
    a = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if "RANK" not in os.environ and a.gpus > 1:
        return self_launch(a.gpus)
    if world != a.gpus:
        raise SystemExit("bench.py --gpus %d runs under WORLD_SIZE=%d: the two must agree (plain `python bench.py --gpus N` starts its own N ranks)" % (a.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device: the MI355X path has no CPU fallback")
    # test hook: C3D_BENCH_SHARE_DEVICE=1 puts every rank on GPU 0 and talks gloo, so the N>1 control flow (sharding, barriers, exchange,
    # max-over-ranks timing) can be exercised on a 1-GPU box; numbers from such a run mean nothing
    share = os.environ.get("C3D_BENCH_SHARE_DEVICE") == "1"
    dev_index = 0 if share else local_rank
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    dist = None
    # test hook: C3D_BENCH_FORCE_DIST=1 under torchrun with ONE rank creates the nccl group and issues every collective of the N > 1 path anyway (RCCL on HIP memory
    # on a one-GPU box: tests/test_zz_rccl_world1.py); timings of such a run carry the (pointless) collectives
    force_dist = os.environ.get("C3D_BENCH_FORCE_DIST") == "1" and world == 1 and "RANK" in os.environ
    if force_dist:
        from c3d_hip import parallel as _par
        _par.SKIP_SINGLE_RANK = False
    dist_on = world > 1 or force_dist        # the step ends with a gradient exchange
    if dist_on:
        import torch.distributed as dist
        if share:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)

    if a.workload == "mesh":
        return main_mesh(a, world, rank, dev, dist)
    if a.workload == "ref-default":
        return main_ref_default(a, world, rank, dev, dist)
    import c3d_hip
    from c3d_hip import synthetic as S
    import diff_gaussian_rasterization as dgr

    if a.lanes <= 0:
        a.lanes = 1
    dgr.sync_free({"on": "verified", "off": False}.get(a.sync_free, a.sync_free))
    dgr.forward_only(a.forward_only == "on")
    N, W, H, deg = a.gaussians, a.width, a.height, a.sh_degree
    K, P = (deg + 1) ** 2, a.width * a.height
    use_renderer = a.render_path != "boundary"
    cloud = S.make_cloud(N, seed=1234, sh_degree=deg, activated=True)          # what the rasterizer consumes (cpu_baseline leg, boundary path)
    poses = S.orbit_poses_64()
    pose_rank = a.as_rank if (a.as_rank >= 0 and world == 1) else rank
    my_poses = [poses[(pose_rank * a.views_per_gpu + i) % len(poses)] for i in range(a.views_per_gpu)]
    t = lambda x: torch.as_tensor(np.asarray(x, dtype=np.float32)).to(dev)
    settings, cams = [], []
    for (r, e, az) in my_poses:
        st = S.camera_settings(W, H, 49.1, e, az, r, bg=(1.0, 1.0, 1.0), sh_degree=deg)
        settings.append(dgr.GaussianRasterizationSettings(H, W, st["tanfovx"], st["tanfovy"], t(st["bg"]), 1.0,
                                                          t(st["viewmatrix"]).reshape(4, 4), t(st["projmatrix"]).reshape(4, 4),
                                                          deg, t(st["campos"]), False, False))
        cams.append(type("Cam", (), dict(image_height=H, image_width=W, FoVx=2 * np.arctan(st["tanfovx"]), FoVy=2 * np.arctan(st["tanfovy"]),
                                         world_view_transform=settings[-1].viewmatrix, full_proj_transform=settings[-1].projmatrix,
                                         camera_center=settings[-1].campos))())
    white = torch.ones(3, device=dev)
    if use_renderer:
        # the reference's call stack: GaussianSplattingRenderer.render over GaussianModel's raw parameters
        from MVs_Algorithms.GaussianSplatting.main_3DGS_renderer import GaussianSplattingRenderer
        raw = S.make_cloud(N, seed=1234, sh_degree=deg, activated=False)
        renderer = GaussianSplattingRenderer(sh_degree=deg, device=dev)
        renderer.initialize({"xyz": raw["means3D"], "features": raw["shs"], "scaling_raw": raw["scales"], "rotation_raw": raw["rotations"],
                             "opacity_raw": raw["opacities"]})
        renderer.force_unfused = a.render_path == "accessor"
        gm = renderer.gaussians
        plist = [gm._xyz, gm._features_dc, gm._features_rest, gm._opacity, gm._scaling, gm._rotation]
        if a.mode == "fwd":
            for q in plist:
                q.requires_grad_(False)
        lr_list = [1.6e-4, 2.5e-3, 1.25e-4, 0.05, 5e-3, 1e-3]
    else:
        params = {k: torch.tensor(v, device=dev, requires_grad=(a.mode != "fwd")) for k, v in cloud.items()}
        names = ["means3D", "shs", "opacities", "scales", "rotations"]
        plist = [params[k] for k in names]
        lr_list = [1.6e-4, 2.5e-3, 0.05, 5e-3, 1e-3]

    def render(i):
        if use_renderer:
            out = renderer.render(cams[i], bg_color=white)
            return out["image"], out["radii"], out["depth"], out["alpha"]
        m2d = torch.zeros_like(params["means3D"], requires_grad=True) if a.mode != "fwd" else None
        return dgr.GaussianRasterizer(settings[i])(means3D=params["means3D"], means2D=m2d, opacities=params["opacities"], shs=params["shs"],
                                                   scales=params["scales"], rotations=params["rotations"])

    # training targets: renders of the xyz-jittered cloud, made once, untimed
    targets = []
    with torch.no_grad():
        g = torch.Generator(device="cpu").manual_seed(4321)
        jit = 0.002 * torch.randn(N, 3, generator=g).to(dev)
        plist[0].data.add_(jit)
        for i in range(len(settings)):
            c, _, _, al = render(i)
            targets.append((c.clone(), al.clone()))
        plist[0].data.sub_(jit)
    tgt_alpha = torch.stack([tg[1] for tg in targets])                 # [V,1,H,W]: the masks of config 3 ("masks = its alpha")
    tgt_masked = torch.stack([tg[0] for tg in targets]) * tgt_alpha
    opt = None
    zero = None
    if a.exchange == "zero1" and (a.mode != "train" or a.render_path != "step"):
        raise SystemExit("--exchange zero1 contains the optimizer step: use it with --mode train (fused step path)")
    if a.mode == "train":
        from c3d_hip.optim import FusedAdam
        opt = FusedAdam([{"params": [q], "lr": lr} for q, lr in zip(plist, lr_list)], lr=0.0, eps=1e-15)
        if a.exchange == "zero1":
            from c3d_hip.parallel import ZeroOneAdam
            zero = ZeroOneAdam(opt, plist, None, average=False)     # parameters re-pointed into one flat buffer, gradients in another, moments for the owned 1/n only
    stats = {"n_vis": [], "D": []}
    fused_step = None
    loss_kind = a.loss if a.loss != "auto" else ("full" if a.mode == "train" else "l1alpha")
    ms_ssim = None
    if loss_kind == "full-torch" and a.render_path == "step" and a.mode != "fwd":
        from shared_utils.msssim import MS_SSIM
        ms_ssim = MS_SSIM(data_range=1, size_average=True, channel=3)
        ms_ssim.use_hip = False       # the comparison point: torch's op chain all the way (grouped convolutions + elementwise ops)
    if a.render_path == "step" and a.mode != "fwd":
        from c3d_hip.gs_step import FusedViewStep
        fused_step = FusedViewStep(N, H, W, dev, lanes=a.lanes, views=len(settings))
        fused_step.time_events = True
        fused_step.defer_status = a.defer_status == "on"
        from c3d_hip.parallel import FlatGrads, status_max
        fused_step.status_sync = status_max(None) if dist_on else None      # N > 1: every rank decides about a step (fit, regrow, redo) from the same status words
        flat_grads = FlatGrads(plist) if zero is None else None        # one buffer: the kernels write into what the collective sends
        step_grads = flat_grads.views if zero is None else zero.grads
        for q, gq in zip(plist, step_grads):
            q.grad = gq                      # the optimizer reads .grad

    view_render = None
    if a.render_path == "step" and a.mode == "fwd":
        from c3d_hip.gs_step import FusedViewRender
        view_render = FusedViewRender(N, H, W, dev, lanes=a.lanes, group=a.group, streams=a.streams)     # all views of the step in one call (one library call per stream)

    def step(collect=False):
        nonlocal fused_step, view_render
        exchanged = False
        if view_render is not None and not collect:
            with torch.no_grad():
                view_render.run(settings, plist)
        elif fused_step is not None and not collect and ms_ssim is None:
            full = loss_kind == "full"      # BASELINE config 3's loss: masked by the target alpha, + 0.2 (1 - MS-SSIM), all inside the library call
            # N > 1, all-reduce mode: the per-Gaussian backward pass runs in `--exchange-chunks` Gaussian ranges and each range's f_rest rows (76 % of the
            # gradient bytes) start their all-reduce as soon as they are enqueued, underneath the next range's kernels (FlatGrads.exchange_rows)
            overlap = dist_on and a.exchange == "allreduce" and a.exchange_chunks > 1
            fused_step.run(settings, [q.detach() for q in plist], step_grads, [tg[0] for tg in targets], [tg[1] for tg in targets], ([tg[1] for tg in targets] if full else None),
                           w_l1=0.8, w_l2=0.0, w_alpha_mse=3.0, scale=1.0 / (a.views_per_gpu * world), accumulate=False, w_ssim=(0.2 if full else 0.0),
                           param_chunks=(a.exchange_chunks if overlap else 1), after_chunk=(flat_grads.exchange_rows if overlap else None))
            if overlap:
                flat_grads.exchange_finish()
                exchanged = True
            for q, gq in zip(plist, step_grads):
                q.grad = gq
        elif fused_step is not None and not collect:
            # BASELINE config 3's loss, exactly the reference's step (main_3DGS.py:169-192): images and references masked by the target alpha,
            # 0.8 L1 + 3 MSE(alpha) + 0.2 (1 - MS-SSIM) over the batch; the rasterizer runs as two sync-free library calls around torch's loss
            colors, _, alphas, _ = fused_step.forward(settings, [q.detach() for q in plist])
            colors.requires_grad_(True); alphas.requires_grad_(True)
            with torch.enable_grad():
                imgs, refs = colors.clamp(0, 1) * tgt_alpha, tgt_masked
                loss = 0.8 * (imgs - refs).abs().mean() + 3.0 * ((alphas - tgt_alpha) ** 2).mean() + 0.2 * (1.0 - ms_ssim(refs, imgs))
                dcolor, dalpha = torch.autograd.grad(loss / world, [colors, alphas])
            fused_step.backward(step_grads, dcolor, dalpha, accumulate=False)
            for q, gq in zip(plist, step_grads):
                q.grad = gq
        else:
            for i in range(len(settings)):
                color, radii, depth, alpha = render(i)
                if collect:
                    stats["n_vis"].append(int((radii > 0).sum().item()))
                    dgr.flush()                      # the pair count of a sync-free forward call reaches the host asynchronously
                    stats["D"].append(int(dgr.last_num_rendered))
                if a.mode != "fwd":
                    tc, ta = targets[i]
                    loss = (color - tc).abs().mean() * 0.8 + 3.0 * ((alpha - ta) ** 2).mean()
                    (loss / (a.views_per_gpu * world)).backward()
        if zero is not None and not collect:
            zero.step()                      # reduce-scatter (all-to-all + rank-ordered sum) -> Adam on the owned slice -> all-gather(parameters)
            return
        if exchanged:
            pass
        elif a.mode != "fwd" and dist_on and fused_step is not None and not collect:
            flat_grads.exchange(None, a.exchange, average=False)
        elif a.mode != "fwd" and dist_on:
            flat = torch.cat([q.grad.reshape(N, -1) for q in plist], dim=1)   # [N, 59] dense gradient
            if a.exchange == "allgather":
                buf = torch.empty((world * N, flat.shape[1]), device=dev)
                dist.all_gather_into_tensor(buf, flat)
                buf = buf.view(world, N, -1)
                flat = buf[0].clone()
                for r in range(1, world):   # fixed rank order -> bit-identical replicas
                    flat += buf[r]
            else:
                dist.all_reduce(flat)
            off = 0
            for q in plist:
                w = q.grad[0].numel()
                q.grad.copy_(flat[:, off:off + w].reshape(q.grad.shape))
                off += w
        if a.mode == "train" and zero is None:
            opt.step()
        if a.mode != "fwd" and fused_step is None:
            for q in plist:
                q.grad = None

    if a.inference_mode == "on":
        if a.mode != "fwd":
            raise SystemExit("--inference-mode on: forward only (--mode fwd)")
        _step_body = step

        def step(collect=False):             # an inference caller of the drop-in API: nothing is differentiated
            with torch.inference_mode():
                return _step_body(collect)

    def sync():
        if fused_step is not None:
            fused_step.finish()              # a deferred step's status words: examined INSIDE the timed region
        torch.cuda.synchronize(dev)
        if dist_on:
            dist.barrier()
            torch.cuda.synchronize(dev)

    for w in range(a.warmup):
        step(collect=(w == 0))
    if a.warmup == 0:
        with torch.no_grad():
            pass
    sync()
    # Inside the timed region only the dominant kernel is timed (two event records per timed launch; timing all ~20 launches of a view costs
    # the multi-stream schedule ~2.5 %, profiles/README.md); the full per-kernel table comes from the separate pass below.
    multi = fused_step if fused_step is not None else view_render
    concurrent_groups = multi is not None and (multi.lanes > 1 or (getattr(multi, "streams", 1) > 1 and multi._plan(a.views_per_gpu)[0] > 1))
    only_dom = ["gs_composite_fwd" if a.mode == "fwd" else "gs_composite_bwd"] if concurrent_groups else None
    c3d_hip.prof_enable(a.timed_prof == "on", only=only_dom)
    ms0 = torch.cuda.memory_stats(dev)
    t0 = time.perf_counter()
    for _ in range(a.steps):
        step()
    t_enqueue = time.perf_counter() - t0      # the host's share: everything enqueued, nothing waited for (per-view paths: is the loop host-bound?)
    sync()
    dt = time.perf_counter() - t0
    ms1 = torch.cuda.memory_stats(dev)
    # hipMalloc / hipFree calls of torch's caching allocator inside the timed region (each synchronises the device): a per-view loop whose buffer sizes change keeps paying them
    allocator = {"device_mallocs": int(ms1.get("num_device_alloc", 0) - ms0.get("num_device_alloc", 0)), "device_frees": int(ms1.get("num_device_free", 0) - ms0.get("num_device_free", 0)),
                 "reserved_GB": round(ms1.get("reserved_bytes.all.current", 0) / 1e9, 2)}
    prof = c3d_hip.prof_read() if a.timed_prof == "on" else {}
    c3d_hip.prof_enable(False)
    # With view lanes > 1 the kernels of different views share the CUs, so a kernel's wall duration inside the timed region is no longer
    # its own cost.  The roofline figure therefore comes from an extra, untimed single-lane pass over the same step (same inputs, same
    # kernels); the concurrent durations of the timed region are reported next to it as "kernels_concurrent".
    prof_conc = None
    prof_views = a.steps * a.views_per_gpu          # views the `prof` table covers
    if concurrent_groups:
        prof_conc = {k: v for k, v in prof.items() if v[1]}
        if fused_step is not None:
            keep_obj, fused_step = fused_step, FusedViewStep(N, H, W, dev, lanes=1, pair_capacity=fused_step.capacity, views=len(settings))
            fused_step._fitted = True
        else:
            keep_obj, view_render = view_render, FusedViewRender(N, H, W, dev, lanes=1, group=a.group, pair_capacity=view_render.capacity)
            view_render._fitted = True
        step()
        sync()
        c3d_hip.prof_enable(True)
        for _ in range(min(a.steps, 4)):
            step()
        sync()
        prof = c3d_hip.prof_read()
        prof_views = min(a.steps, 4) * a.views_per_gpu
        c3d_hip.prof_enable(False)
        if fused_step is not None:
            fused_step = keep_obj
        else:
            view_render = keep_obj
    per_rank = None
    if dist_on:
        # every rank's own step time and pair counts into the one line: rank r renders one elevation band of the orbit, the step is as slow as the slowest band
        mine_ = torch.tensor([dt / a.steps * 1e3, float(np.mean(stats["D"])) if stats["D"] else 0.0, float(np.mean(stats["n_vis"])) if stats["n_vis"] else 0.0], device=dev, dtype=torch.float64)
        allr = [torch.zeros_like(mine_) for _ in range(dist.get_world_size())]
        dist.all_gather(allr, mine_)
        rows = [x.tolist() for x in allr]
        per_rank = {"ms_per_step": [round(r_[0], 3) for r_ in rows], "ms_per_step_min": round(min(r_[0] for r_ in rows), 3), "ms_per_step_max": round(max(r_[0] for r_ in rows), 3),
                    "tile_splat_pairs_per_view": [int(r_[1]) for r_ in rows], "n_visible_per_view": [int(r_[2]) for r_ in rows],
                    "views": ["rank %d: orbit cameras %s" % (r_, [(rank_pose[1], rank_pose[2]) for rank_pose in [poses[(r_ * a.views_per_gpu + i) % len(poses)] for i in range(a.views_per_gpu)]][:2] + ["..."]) for r_ in range(dist.get_world_size())]}
        tt = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())

    views_total = a.views_per_gpu * world * a.steps
    value = views_total * P / dt / 1e6
    if not stats["D"]:
        stats = {"n_vis": [0], "D": [0]}
    n_vis, D = float(np.mean(stats["n_vis"])), float(np.mean(stats["D"]))
    alg = algorithmic_bytes(N, K, P, n_vis, D)
    kern = {}
    for name, (ms, n) in prof.items():
        avg = ms / n
        # every stage of the chain is ONE launch for all views of a group (round 4): the units a launch processes = views per timed scope
        vpl = prof_views / n if name in ("gs_preprocess", "gs_preprocess_bwd", "gs_depth_sort", "gs_emit", "gs_tile_sort", "gs_ranges", "gs_composite_fwd", "gs_composite_bwd") else 1.0
        kern[name] = {"avg_ms": round(avg, 4), "launches": n, "views_per_launch": round(vpl, 2), "share": 0.0, "ms_per_view": round(ms / max(prof_views, 1), 4),
                      "alg_GBps": round(alg.get(name, 0) * vpl / (avg * 1e-3) / 1e9, 1) if avg > 0 else None}
    tot = sum(ms for ms, _ in prof.values()) or 1.0
    for name, (ms, n) in prof.items():
        kern[name]["share"] = round(ms / tot, 3)
    # next to every algorithmic rate the rate the memory system really saw: PMC bytes (calibrated per access pattern) / the kernel's duration in the PMC pass.
    # The batched kernels stream the parameters once per LAUNCH while SURVEY 8(d) credits them per VIEW: their alg_GBps is a contract figure, not an HBM rate.
    cal, cal_file = load_calibration()
    pmc_all, pmc_all_file, _ = load_profile_json("_pmc_traffic.json", exclude="_mesh_")
    group_kernels = {"gs_composite_bwd": ["k_composite_bwd"], "gs_composite_fwd": ["k_composite_fwd_w"], "gs_preprocess": ["k_preprocess_views_r", "k_preprocess"],
                     "gs_preprocess_bwd": ["k_bwd_views_geom", "k_bwd_views_sh"], "gs_emit": ["k_emit"], "adam": ["k_adam"]}
    for name, bases in group_kernels.items():
        if name not in kern:
            continue
        byts, us = 0.0, 0.0
        for kname, rec in pmc_all.items():
            if kname.startswith("_") or kname.split("<")[0].strip() not in bases:
                continue
            bb, _ = calibrated_bytes(rec, kname, cal)
            byts += bb * rec["launches"]; us += rec["avg_us"] * rec["launches"]
        kern[name]["hbm_GBps"] = round(byts / (us * 1e-6) / 1e9, 1) if us > 0 else None
    roof = None
    # HBM traffic of the dominant kernel: rocprofv3 PMC counters cannot be read from inside the process, so the per-launch figure comes
    # from the committed summary of two separate --pmc passes over this same command (profiles/summarize_pmc.py; FETCH_SIZE raw + WRITE_SIZE,
    # MI355X_MICROARCH.md: FETCH_SIZE under-reports wide streaming reads by 2x on gfx950 -- both figures are in the file).
    pmc, pmc_file, pmc_stale = load_profile_json("_pmc_traffic.json", exclude="_mesh_")
    # kernel behind a profiling group; template instances ("k_composite_bwd<true>": with the fused pixel loss) are matched by base name, most launches first
    pmc_base = {"gs_composite_bwd": "k_composite_bwd", "gs_composite_fwd": "k_composite_fwd_w", "gs_preprocess": "k_preprocess_views_r",
                "gs_preprocess_bwd": "k_bwd_views_geom", "gs_emit": "k_emit"}

    def kernel_row(table, group, launches=None):
        # the template instance of the group's kernel that took the most time in the profiled command (the multi-view launches of the timed path, not the per-view
        # launches of the warm-up's statistics pass)
        base = pmc_base.get(group)
        cand = [k for k in table if not k.startswith("_") and k.split("<")[0].strip() == base]
        return max(cand, key=lambda k: float(table[k].get("launches", 0)) * float(table[k].get("avg_us", 0) or 0)) if cand else None
    if prof:
        dom = max(prof, key=lambda k: prof[k][0])
        avg_s = prof[dom][0] / prof[dom][1] * 1e-3
        dom_vpl = kern[dom]["views_per_launch"]
        dom_bytes = alg.get(dom, 0) * dom_vpl                 # SURVEY 8(d)'s per-view figure x the views one launch processes
        ach = dom_bytes / avg_s / 1e9
        roof = {"bound": "hbm", "kernel": dom, "achieved": round(ach, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                "frac": round(ach / HBM_PEAK_GBPS, 4), "traffic": None, "avg_ms": round(avg_s * 1e3, 4),
                "alg_bytes_per_launch": int(dom_bytes), "alg_bytes_per_view": int(alg.get(dom, 0)), "views_per_launch": dom_vpl,
                "measured": "HIP events on the launch stream inside the timed region (one group of views per launch: the kernel runs alone)"}
        rec_name = kernel_row(pmc, dom)
        rec = pmc.get(rec_name) if rec_name else None
        if rec and a.workload == "gs" and N == 1_000_000 and (W, H) == (1920, 1080):
            tb, tnote = calibrated_bytes(rec, rec_name, cal)
            roof["traffic"] = int(tb)
            roof["traffic_note"] = "per launch of %s from profiles/%s (same code digest): %s%s; raw FETCH_SIZE + WRITE_SIZE = %d" % (
                rec_name, pmc_file, tnote, (" [profiles/%s]" % cal_file) if cal_file else "", int((rec["fetch_MB_raw"] + rec["write_MB"]) * 1e6))
            roof["hbm_GBps"] = round(tb / (rec["avg_us"] * 1e-6) / 1e9, 1)
        elif pmc_stale:
            roof["stale"] = "profiles/%s was measured on code %s, this is %s" % (pmc_file, pmc_stale, code_digest())

        # instruction-issue view of the same kernel (it is what actually bounds the compositing kernels): SQ counters of a
        # single-lane rocprofv3 pass, committed like the PMC traffic (per SIMD: counters are per shader engine = 32 SIMDs)
        try:
            import csv
            sq = sorted(f for f in os.listdir(os.path.join(ROOT, "profiles")) if f.endswith("_sq_instruction_mix_lanes1.csv"))
            sq_meta = os.path.join(ROOT, "profiles", sq[-1] + ".meta.json") if sq else None
            if sq and not (os.path.exists(sq_meta) and json.load(open(sq_meta)).get("code_digest") == code_digest()):
                roof["issue"] = {"stale": "profiles/%s describes other kernel code" % sq[-1]}
                sq = []
            if sq and pmc_base.get(dom):
                rows = {row["kernel"]: row for row in csv.DictReader(open(os.path.join(ROOT, "profiles", sq[-1])))}
                hit = kernel_row(rows, dom, launches=lambda r: r.get("launches", 0))
                for row in ([rows[hit]] if hit else []):
                    if True:
                        n_ins = sum(float(row[c]) for c in ("SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_BRANCH", "SQ_INSTS_SMEM", "SQ_INSTS_VMEM")) / 32.0
                        roof["issue"] = {"instructions_per_simd": int(n_ins), "valu": int(float(row["SQ_INSTS_VALU"]) / 32), "salu": int(float(row["SQ_INSTS_SALU"]) / 32),
                                         "lds": int(float(row["SQ_INSTS_LDS"]) / 32),
                                         "busy_cycles": int(float(row["SQ_BUSY_CYCLES"])), "cycles_per_instruction": round(float(row["SQ_BUSY_CYCLES"]) / n_ins, 2),
                                         "kernel": hit, "source": "profiles/" + sq[-1]}
                        # the secondary roofline of SURVEY 8(d): FP32 vector issue.  One VALU instruction = 64 lane operations; the chip's peak is 157.3 TFLOP/s with
                        # every instruction an FMA = 78.6 T lane-operations per second.  valu_frac = cycles the VALU pipes were busy / kernel cycles (pipe-activity pass).
                        us = float(row.get("avg_us", 0) or 0)
                        if us > 0:
                            ops = float(row["SQ_INSTS_VALU"]) * 32.0 * 64.0 / (us * 1e-6)     # counters are per shader engine: x 32 engines, x 64 lanes
                            roof["issue"]["valu_lane_ops_per_s"] = round(ops, -9)
                            roof["issue"]["valu_lane_ops_frac_of_78.6T"] = round(ops / 78.6e12, 4)
                        pa_file = sq[-1].replace("_sq_instruction_mix_", "_sq_pipe_activity_")
                        if os.path.exists(os.path.join(ROOT, "profiles", pa_file)):
                            pa = {r_["kernel"]: r_ for r_ in csv.DictReader(open(os.path.join(ROOT, "profiles", pa_file)))}
                            if hit in pa:
                                # SQ_ACTIVE_INST_VALU counts one unit per issued wave instruction, summed over the 32 SIMDs the counter row covers; a wave64 instruction
                                # occupies a CDNA4 SIMD (32 lanes) for 2 cycles -- the x4 of the SIMD-16 parts' "VALUBusy" formula gives fractions above 1 here (1.58 for the
                                # forward compositing kernel).  SQ_BUSY_CYCLES are the cycles of the launch.
                                roof["issue"]["valu_frac"] = round(float(pa[hit]["SQ_ACTIVE_INST_VALU"]) * 2.0 / 32.0 / float(row["SQ_BUSY_CYCLES"]), 4)
                                roof["issue"]["cycles_per_valu_instruction"] = round(float(row["SQ_BUSY_CYCLES"]) / (float(row["SQ_INSTS_VALU"]) / 32.0), 2)
                                roof["issue"]["lds_frac"] = round(float(pa[hit]["SQ_LDS_IDX_ACTIVE"]) / 8.0 / float(row["SQ_BUSY_CYCLES"]), 4)
                                roof["issue"]["pipe_source"] = "profiles/" + pa_file
        except Exception:
            pass
        if prof_conc is not None:
            roof["measured"] = "single-stream pass after the timed region (kernels run alone); the timed region had %d groups of views in flight" % max(a.lanes, multi._plan(a.views_per_gpu)[0] if hasattr(multi, "_plan") else 1)
            if prof_conc and dom in prof_conc:
                roof["avg_ms_concurrent"] = round(prof_conc[dom][0] / prof_conc[dom][1], 4)
    # whole-chain figure: SURVEY 8(d)'s algorithmic bytes of a VIEW over the wall time a view takes in the timed
    # region -- the number the north star's ">= 60 % of HBM roofline on forward raster" is about
    chain = None
    per_view_s = dt / max(a.steps * a.views_per_gpu, 1)
    b_fwd = N * (44 + 12 * K) + 96 * n_vis + 68 * D + 20 * P
    b_bwd = 32 * P + 48 * D + 96 * n_vis + 2 * N * (44 + 12 * K)
    b_view = b_fwd if a.mode == "fwd" else b_fwd + b_bwd
    if per_view_s > 0:
        chain = {"what": "SURVEY 8(d) algorithmic bytes of one view (%s) / wall time per view in the timed region" % ("B_fwd" if a.mode == "fwd" else "B_fwd + B_bwd"),
                 "bytes_per_view": int(b_view), "ms_per_view": round(per_view_s * 1e3, 4), "achieved": round(b_view / per_view_s / 1e9, 1), "peak": HBM_PEAK_GBPS,
                 "unit": "GB/s", "frac": round(b_view / per_view_s / 1e9 / HBM_PEAK_GBPS, 4), "target_frac": 0.6 if a.mode == "fwd" else None}
        # the same with the parameter set counted once per LAUNCH that streams it (k_preprocess_views / k_bwd_views_*: once per step, or once per group of
        # `lanes` views forward-only) instead of once per view as the contract formula does: what the memory system is really asked for
        share = a.views_per_gpu if a.mode != "fwd" else (max(1, min(a.group, view_render._plan(a.views_per_gpu)[1])) if view_render is not None else 1)      # forward-only: once per group of views
        par = N * (44 + 12 * K)
        b_once = b_view - par * (1 if a.mode == "fwd" else 3) * (1.0 - 1.0 / share)
        chain["bytes_per_view_params_once_per_launch"] = int(b_once)
        chain["frac_params_once_per_launch"] = round(b_once / per_view_s / 1e9 / HBM_PEAK_GBPS, 4)
    kern_conc = None
    if prof_conc:
        kern_conc = {name: round(ms / n, 4) for name, (ms, n) in prof_conc.items()}

    cpu = None
    if rank == 0 and world == 1 and a.cpu_baseline != "off":
        try:
            from oracle import gs_oracle as O
            ncore = os.cpu_count() or 1
            r, e, az = my_poses[0]
            st = S.camera_settings(W, H, 49.1, e, az, r, bg=(1.0, 1.0, 1.0), sh_degree=deg)
            t1 = time.perf_counter()
            oc, orad, od, oa, ost = O.forward(cloud["means3D"], cloud["opacities"], st, shs=cloud["shs"], scales=cloud["scales"],
                                              rotations=cloud["rotations"], nthreads=ncore)
            if a.mode != "fwd":
                O.backward(ost, np.ones((3, H, W), np.float32) / P, nthreads=ncore)
            tc = time.perf_counter() - t1
            t_sort = O.last_sort_seconds(np.float32)
            cpu = {"value": round(P / tc / 1e6, 4), "unit": "Mpixels/s", "cores": ncore, "kind": "port", "seconds": round(tc, 2), "serial_sort_seconds": round(t_sort, 2),
                   "value_without_serial_sort": round(P / max(tc - t_sort, 1e-9) / 1e6, 4),
                   "sample": "1 view of the same workload (%d Gaussians, %dx%d, %s) on the CPU oracle, %.1f s of which %.1f s are ONE serial qsort of the (tile, depth) pairs; the rest "
                             "(projection, compositing, backward) runs OpenMP over Gaussians / tiles on %d threads.  `value_without_serial_sort` is what those threads do; a stated "
                             "baseline, not a target" % (N, W, H, a.mode, tc, t_sort, ncore)}
        except Exception as ex:   # the baseline leg must never take the bench down
            cpu = {"value": None, "unit": "Mpixels/s", "cores": 0, "kind": "port", "sample": "failed: %r" % (ex,)}

    # north-star target carried into THIS line: ">= 60 % of MI355X HBM roofline on 1M-Gaussian 1080p forward raster" is a statement about
    # BASELINE config 2 -- forward only, the 64 orbit cameras.  Run it here, after the timed region, so that the driver's record holds the number.
    targets_out = None
    host_ms_timed = (round(fused_step.last_host_ms, 3) if fused_step is not None else round(t_enqueue / a.steps * 1e3, 3))      # (the extra passes below run more steps)
    gpu_span_timed = (round(getattr(fused_step, "last_gpu_ms", 0.0), 3) if fused_step is not None else None)
    if rank == 0 and world == 1 and a.mode != "fwd" and a.targets == "on" and use_renderer and a.render_path == "step":
        try:
            from c3d_hip.gs_step import FusedViewRender
            all_settings = []
            for (r_, e_, az_) in poses:
                st = S.camera_settings(W, H, 49.1, e_, az_, r_, bg=(1.0, 1.0, 1.0), sh_degree=deg)
                all_settings.append(dgr.GaussianRasterizationSettings(H, W, st["tanfovx"], st["tanfovy"], t(st["bg"]), 1.0, t(st["viewmatrix"]).reshape(4, 4),
                                                                      t(st["projmatrix"]).reshape(4, 4), deg, t(st["campos"]), False, False))
            vr = FusedViewRender(N, H, W, dev, lanes=1, group=16, streams=a.streams)
            pl_ = [q.detach() for q in plist]
            with torch.no_grad():
                vr.run(all_settings, pl_); vr.run(all_settings, pl_)          # capacity fit + warm-up
                torch.cuda.synchronize(dev)
                t1 = time.perf_counter()
                for _ in range(3):
                    vr.run(all_settings, pl_)
                torch.cuda.synchronize(dev)
                tf = (time.perf_counter() - t1) / 3
            pv = tf / len(all_settings)
            targets_out = {"what": "BASELINE config 2 after the timed region: forward only, the 64 orbit cameras in one FusedViewRender.run (%d HIP streams, one c3d_gs_render_views_raw call of 16-view launches each), 3 passes" % vr._plan(len(all_settings))[0],
                       "fwd_Mpx": round(len(all_settings) * P / tf / 1e6, 1), "fwd_ms_per_view": round(pv * 1e3, 4),
                       "fwd_chain_frac": round(b_fwd / pv / 1e9 / HBM_PEAK_GBPS, 4), "fwd_chain_GBps": round(b_fwd / pv / 1e9, 1), "fwd_bytes_per_view": int(b_fwd),
                       "target_frac": 0.6, "north_star": ">= 60 % of MI355X HBM roofline on 1M-Gaussian 1080p forward raster (SURVEY 8d algorithmic bytes B_fwd / wall time per view / 8 TB/s)"}
            del vr
        except Exception as ex:      # never take the headline down
            targets_out = {"error": repr(ex)}
        # BASELINE configs 3 and 5 in the driver's line as well: the same commands as `--mode train` and `--workload mesh`, a few steps each, after the timed region
        if a.mode == "fwdbwd" and fused_step is not None and a.loss == "auto" and not dist_on:
            try:
                from c3d_hip.optim import FusedAdam
                opt = FusedAdam([{"params": [q], "lr": lr} for q, lr in zip(plist, lr_list)], lr=0.0, eps=1e-15)
                a.mode, loss_kind = "train", "full"       # (step() reads both: config 3's loss inside the library call + the fused Adam step)
                for _ in range(3):
                    step()
                sync()
                t1 = time.perf_counter()
                for _ in range(5):
                    step()
                sync()
                tt = (time.perf_counter() - t1) / 5
                targets_out.update({"train_what": "BASELINE config 3 after the timed region: 5 steps of `--mode train` (8 views: forward, 0.8 L1 + 3 MSE(alpha) + 0.2 (1 - MS-SSIM) masked, backward, fused Adam)",
                                "train_Mpx": round(a.views_per_gpu * P / tt / 1e6, 1), "train_ms_per_step": round(tt * 1e3, 3)})
            except Exception as ex:
                targets_out["train_error"] = repr(ex)
            finally:
                a.mode, loss_kind, opt = "fwdbwd", "l1alpha", None
                for q in plist:
                    q.grad = None
            try:
                import copy
                am = copy.copy(a)
                am.workload, am.steps, am.warmup, am.cpu_baseline, am.views_per_gpu, am.render_path = "mesh", 10, 3, "off", 8, "step"
                ml = main_mesh(am, 1, 0, dev, None, emit=False)
                targets_out.update({"mesh_what": "BASELINE config 5 after the timed region: 10 steps of `--workload mesh` (499k triangles, 1024x1024, 8 views per step, forward + backward)",
                                "mesh_Mpx": ml["value"], "mesh_ms_per_step": ml["ms_per_step"], "mesh_chain_frac": (ml["roofline_chain"] or {}).get("frac"),
                                "mesh_dominant_kernel": {k: (ml["roofline"] or {}).get(k) for k in ("kernel", "avg_ms", "frac", "traffic")}})
            except Exception as ex:
                targets_out["mesh_error"] = repr(ex)

    if rank == 0:
        out = {
            "metric": "Mpixels/s 3DGS forward+backward @1M Gaussians 1080p" if a.mode == "fwdbwd" else
                      ("Mpixels/s 3DGS forward @1M Gaussians 1080p" if a.mode == "fwd" else "Mpixels/s 3DGS training step (forward+loss+backward+Adam) @1M Gaussians 1080p"),
            "value": round(value, 2), "unit": "Mpixels/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": round(dt / a.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": "3DGS %s, %d synthetic Gaussians (seed 1234) SH deg %d, %dx%d, %d orbit views/GPU/step of the 64-camera orbit"
                                   % (a.mode, N, deg, W, H, a.views_per_gpu),
                       "global_views_per_step": a.views_per_gpu * world, "parallelism": "view-parallel dp%d" % world,
                       "exchange": (a.exchange if dist_on and a.mode != "fwd" else "none"), "dist_backend": (dist.get_backend() if dist_on else None),
                       "rccl_ranks": (dist.get_world_size() if dist_on else 1), "per_rank": per_rank,
                       "exchange_chunks": (a.exchange_chunks if dist_on and a.mode != "fwd" and a.exchange == "allreduce" and fused_step is not None else 1), "render_path": a.render_path,
                       "loss": (None if a.mode == "fwd" else ("0.8 L1 + 3 MSE(alpha) + 0.2 (1 - MS-SSIM), masked (BASELINE config 3)%s" % (", MS-SSIM by torch ops" if ms_ssim is not None else ", fused HIP")
                                                                if loss_kind != "l1alpha" and a.render_path == "step" else "0.8 L1 + 3 MSE(alpha)")), "view_lanes": (a.lanes if a.render_path == "step" else 1),
                       "hip_streams": (view_render._plan(a.views_per_gpu)[0] if view_render is not None else 1),
                       "host_enqueue_ms_per_step": host_ms_timed,
                       "sync_free_drop_in": ({"on": "verified", "off": False}.get(a.sync_free, a.sync_free)) if a.render_path != "step" else None,
                       "forward_only_flag": (a.forward_only == "on") if (a.render_path != "step" and a.mode == "fwd") else None,
                       "inference_mode": (a.inference_mode == "on") if a.render_path != "step" else None,
                       "allocator_in_timed_region": allocator, "drop_in_calls_rendered_twice": int(dgr.redone_calls) if a.render_path != "step" else None,
                       "drop_in_calls_beyond_launch_hint": int(dgr.beyond_hint_calls) if a.render_path != "step" else None,
                       "defer_status": (fused_step.defer_status if fused_step is not None else None),
                       "gpu_span_ms_last_step": gpu_span_timed,
                       "n_visible": n_vis, "tile_splat_pairs": D},
            "roofline": roof, "roofline_chain": chain, "targets": targets_out, "cpu_baseline": cpu, "kernels": kern, "kernels_concurrent_avg_ms": kern_conc,
            "code_digest": code_digest(),
        }
        print(json.dumps(out))
    if dist_on:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
