"""Regenerates tests/golden/ref_gs_train.npz by RUNNING THE REFERENCE'S OWN TRAINING LOOP -- GaussianSplatting3D.prepare_training and
.training (/root/reference/MVs_Algorithms/GaussianSplatting/main_3DGS.py:106-232) with its camera controller (:76-82,
shared_utils/camera_utils.py:216-251), renderer glue, Adam set-up, learning-rate schedule, densify / prune and opacity reset
(main_3DGS_renderer.py:435-781) -- on the CPU in this container, for a few dozen steps on a small cloud, and recording the model after
every step.

Replaced, because they are third-party packages this image lacks (all named in the reference's requirement files):
  * diff_gaussian_rasterization -> tests/fake_dgr.py: the CPU oracle's forward and backward wired as one autograd function;
  * pytorch_msssim.MS_SSIM      -> a constant 0: the run uses lambda_ssim = 0, where the reference's loss is (1-0)*L1 + lambda_alpha*MSE
                                   + 0*(1 - ms_ssim) -- the MS-SSIM term cannot be evaluated on images this small anyway;
  * kiui.cam.orbit_camera, kiui.op.inverse_sigmoid -> restated (kiui 0.2.14): the orbit pose x = r cos(e) sin(a), y = -r sin(e),
    z = r cos(e) cos(a) with an OpenGL look-at at the target; log(x / (1 - x));
  * comfy.utils.ProgressBar     -> a recorder whose update_absolute() snapshots the model: that is how the per-step trajectory is read.
GaussianSplatting3D.__init__ (:86-104) is reproduced line by line except `renderer.initialize`, whose point-cloud initialisation needs
simple_knn's CUDA kernel: the model's tensors are installed directly.  Everything else that runs is the reference's code.

  python tests/golden/make_golden_ref_gs_train.py [--check]
"""
import os
import random
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), HERE]
import make_golden_ref_py as G  # noqa: E402

OUT = os.path.join(HERE, "ref_gs_train.npz")

PARAMS = dict(training_iterations=26, batch_size=2, lambda_ssim=0.0, lambda_alpha=3, invert_bg_prob=0.5, num_pts=0, density_start_iter=3,
              density_end_iter=22, densification_interval=6, opacity_reset_interval=9, densify_grad_threshold=0.0002, sh_degree=3,
              position_lr_init=0.00016 * 50, scaling_lr=0.02, opacity_lr=0.1)         # livelier than the defaults so that 26 steps move things
SEEDS = dict(python=11, numpy=12, torch=13)
H, W, FOVY = 32, 40, 49.1


def scene():
    rng = np.random.default_rng(5)
    N = 160
    d = {}
    d["xyz"] = (rng.normal(size=(N, 3)) * 0.3).astype(np.float32)
    d["f_dc"] = (rng.normal(size=(N, 1, 3)) * 0.8).astype(np.float32)
    d["f_rest"] = (rng.normal(size=(N, 15, 3)) * 0.05).astype(np.float32)
    d["scaling"] = rng.normal(np.log(0.06), 0.5, (N, 3)).astype(np.float32)
    d["rotation"] = rng.normal(size=(N, 4)).astype(np.float32)
    d["opacity"] = rng.normal(0.0, 1.2, (N, 1)).astype(np.float32)
    V = 4
    yy, xx = np.meshgrid(np.linspace(-1, 1, H), np.linspace(-1, 1, W), indexing="ij")
    imgs, masks = [], []
    for v in range(V):
        cx, cy, rad = 0.2 * np.cos(v), 0.2 * np.sin(v), 0.55 + 0.05 * v
        m = ((xx - cx) ** 2 + (yy - cy) ** 2 < rad ** 2).astype(np.float32)
        img = np.stack([0.5 + 0.5 * np.sin(3 * xx + v), 0.5 + 0.5 * np.cos(2 * yy - v), 0.5 + 0.4 * np.sin(xx * yy * 4 + v)], -1).astype(np.float32)
        imgs.append(img); masks.append(m)
    d["ref_images"], d["ref_masks"] = np.stack(imgs), np.stack(masks)              # [V,H,W,3], [V,H,W]
    d["poses"] = np.asarray([[2.0, -20.0, 0.0, 0, 0, 0], [2.0, 10.0, 90.0, 0, 0, 0], [2.2, 30.0, 180.0, 0.05, 0, 0], [1.9, 0.0, -90.0, 0, 0.05, 0]], np.float64)
    return d


def orbit_camera(elevation, azimuth, radius=1, is_degree=True, target=None, opengl=True):
    """kiui.cam.orbit_camera (0.2.14) as that package documents it; the look-at is the reference's own camera_utils.look_at"""
    cam = sys.modules["shared_utils.camera_utils"]
    if is_degree:
        elevation, azimuth = np.deg2rad(elevation), np.deg2rad(azimuth)
    x = radius * np.cos(elevation) * np.sin(azimuth)
    y = -radius * np.sin(elevation)
    z = radius * np.cos(elevation) * np.cos(azimuth)
    if target is None:
        target = np.zeros([3], dtype=np.float32)
    campos = np.array([x, y, z]) + target
    T = np.eye(4, dtype=np.float32)
    T[:3, :3] = cam.look_at(campos, target, opengl)
    T[:3, 3] = campos
    return T


class Recorder:
    """stands in for comfy.utils.ProgressBar; update_absolute(step + 1) is the last statement of every training step"""
    model = None
    rows = []

    def __init__(self, total):
        pass

    def update_absolute(self, value, *a, **k):
        g = Recorder.model
        opt = g.optimizer
        lr = [grp["lr"] for grp in opt.param_groups if grp["name"] == "xyz"][0]
        Recorder.rows.append([value, g._xyz.shape[0], float(g._xyz.detach().double().sum()), float(g._xyz.detach().double().abs().sum()),
                              float(g.get_opacity.detach().double().mean()), float(g.get_scaling.detach().double().mean()),
                              float(g._features_dc.detach().double().sum()), float(g._rotation.detach().double().abs().sum()), lr,
                              float(g.max_radii2D.double().sum()), float(g.denom.double().sum()), float(g.xyz_gradient_accum.double().sum())])


def reference_trainer_module():
    G._install_stubs()
    G._cpu_redirect()
    for name in ("torchvision", "torchvision.transforms", "torchvision.transforms.functional", "PIL", "PIL.Image"):
        if name not in sys.modules:
            try:
                __import__(name)
            except Exception:
                m = G._Stub(name); m.__path__ = []
                sys.modules[name] = m

    class _Event:
        def __init__(self, *a, **k):
            pass

        def record(self, *a, **k):
            pass
    torch.cuda.Event, torch.cuda.synchronize, torch.cuda.empty_cache = _Event, (lambda *a, **k: None), (lambda *a, **k: None)
    sys.modules["kiui.op"].inverse_sigmoid = lambda x: torch.log(x / (1 - x))
    sys.modules["kiui.cam"].orbit_camera = orbit_camera
    sys.modules["comfy.utils"].ProgressBar = Recorder

    class _ZeroSSIM(torch.nn.Module):
        def __init__(self, *a, **k):
            super().__init__()

        def forward(self, x, y):
            return torch.zeros((), dtype=x.dtype)
    sys.modules["pytorch_msssim"].MS_SSIM = _ZeroSSIM
    sys.modules["pytorch_msssim"].SSIM = _ZeroSSIM
    import fake_dgr
    fake_dgr.RECORD = False
    sys.modules["diff_gaussian_rasterization"] = fake_dgr
    _, cam, ren = G.reference_modules()
    sys.modules["shared_utils"].__path__ = [os.path.join(G.REF, "shared_utils")]
    G._load("shared_utils.image_utils", "shared_utils/image_utils.py")
    # main_3DGS.py imports its renderer relatively: give it the package context it has in the reference tree
    pkg = types.ModuleType("ref_gs_pkg"); pkg.__path__ = [os.path.join(G.REF, "MVs_Algorithms", "GaussianSplatting")]
    sys.modules["ref_gs_pkg"] = pkg
    sys.modules["ref_gs_pkg.main_3DGS_renderer"] = ren
    tr = G._load("ref_gs_pkg.main_3DGS", "MVs_Algorithms/GaussianSplatting/main_3DGS.py")
    return tr, ren, cam


def generate():
    tr, ren, cam = reference_trainer_module()
    sc = scene()
    out = {"scene_" + k: v for k, v in sc.items()}
    gp = tr.GSParams(**PARAMS)
    t = object.__new__(tr.GaussianSplatting3D)                       # __init__ :86-104 without renderer.initialize (see the header)
    t.device = torch.device("cpu")
    t.renderer = ren.GaussianSplattingRenderer(sh_degree=gp.sh_degree)
    g = t.renderer.gaussians
    P = lambda a: torch.nn.Parameter(torch.from_numpy(a.copy()).requires_grad_(True))
    g._xyz, g._features_dc, g._features_rest = P(sc["xyz"]), P(sc["f_dc"]), P(sc["f_rest"])
    g._scaling, g._rotation, g._opacity = P(sc["scaling"]), P(sc["rotation"]), P(sc["opacity"])
    g.init_xyz = torch.from_numpy(sc["xyz"].copy())
    g.max_radii2D = torch.zeros((sc["xyz"].shape[0],))
    g.spatial_lr_scale = 1.0
    g.training_setup(gp)
    g.active_sh_degree = g.max_sh_degree
    t.optimizer = g.optimizer
    t.ms_ssim_loss = sys.modules["pytorch_msssim"].MS_SSIM()
    t.gs_params = gp
    imgs = [torch.from_numpy(a.copy()) for a in sc["ref_images"]]
    masks = [torch.from_numpy(a.copy()) for a in sc["ref_masks"]]
    t.prepare_training(imgs, masks, [tuple(p) for p in sc["poses"]], FOVY)
    random.seed(SEEDS["python"]); np.random.seed(SEEDS["numpy"]); torch.manual_seed(SEEDS["torch"])
    Recorder.model, Recorder.rows = g, []
    t.training()
    out["trajectory"] = np.asarray(Recorder.rows, np.float64)
    out["trajectory_columns"] = np.asarray(["step", "points", "sum_xyz", "sum_abs_xyz", "mean_opacity", "mean_scaling", "sum_f_dc", "sum_abs_rotation",
                                            "lr_xyz", "sum_max_radii2D", "sum_denom", "sum_xyz_gradient_accum"])
    for k in ("xyz", "features_dc", "features_rest", "scaling", "rotation", "opacity"):
        out["final_" + k] = getattr(g, "_" + k).detach().numpy()
    out["ref_imgs_torch"], out["ref_masks_torch"] = t.ref_imgs_torch.numpy(), t.ref_masks_torch.numpy()
    return {k: np.asarray(v) for k, v in out.items()}


def main():
    out = generate()
    if "--check" in sys.argv:
        ref = np.load(OUT)
        bad = [k for k in out if k not in ref.files or out[k].shape != ref[k].shape or not np.array_equal(out[k], ref[k])]
        bad += [k for k in ref.files if k not in out]
        if bad:
            print("MISMATCH:", bad)
            sys.exit(1)
        print("ok: %d arrays identical to the committed fixture" % len(out))
        return
    np.savez_compressed(OUT, **out)
    print("wrote", OUT, "with", len(out), "arrays")
    print(np.array2string(out["trajectory"][:, :6], precision=5, suppress_small=True))


if __name__ == "__main__":
    main()
