"""Regenerates tests/golden/ref_mesh_train.npz by RUNNING THE REFERENCE'S OWN MESH TRAINING LOOP -- DiffMesh.prepare_training and
.training (/root/reference/MVs_Algorithms/DiffRastMesh/diff_mesh.py:58-159) with its camera controller (:18-22), DiffRastRenderer.render
(diff_mesh_renderer.py:72-159) and Adam over raw_albedo + v_offsets (:47) -- on the CPU in this container, recording the parameters
after every step.  Geometry trains (train_mesh_geometry=True), so gradients flow back through antialias, interpolate and rasterize.

Replaced, because they are third-party packages this image lacks:
  * nvdiffrast.torch             -> tests/fake_dr.py: the CPU mesh oracle's forward and backward per op as autograd functions;
  * pytorch_msssim.MS_SSIM       -> a constant 0; the run uses ms_ssim_loss_weight = 0 (loss = MSE + regularisers);
  * kiui.mesh_utils.laplacian_smooth_loss / normal_consistency -> this repo's restatements of the published kiui source (the same functions the
    mirror's trainer calls: they are identical on both sides and therefore not pinned by this fixture -- the weights 0.01 / 0.001 / 0.1 and what
    they are applied to are; tests/test_host_logic.py holds the restatements to a literal sparse-matrix / edge-table form in float64);
  * kiui.cam.orbit_camera, kiui.op.inverse_sigmoid / safe_normalize -> restated as in make_golden_ref_gs_train.py / _render.py;
  * comfy.utils.ProgressBar      -> the per-step recorder.
DiffMesh.__init__ (:26-56) is reproduced line by line with the device set to the CPU; remeshing (pymeshlab) is kept out of reach with
remesh_after_n_iteration above the iteration count.

  python tests/golden/make_golden_ref_mesh_train.py [--check]
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
import make_golden_ref_gs_train as GT  # noqa: E402
import make_golden_ref_render as GR  # noqa: E402

OUT = os.path.join(HERE, "ref_mesh_train.npz")
ARGS = dict(training_iterations=10, batch_size=2, texture_learning_rate=0.05, train_mesh_geometry=True, geometry_learning_rate=0.002,
            ms_ssim_loss_weight=0.0, remesh_after_n_iteration=10_000, invert_bg_prob=0.5, force_cuda_rasterize=True)
SEEDS = dict(python=21, numpy=22, torch=23)
H, W, FOVY = 40, 48, 49.1


def regularisers():
    """the mirror's restatements of the two kiui.mesh_utils losses (from the published kiui source; the wheel is absent), loaded from the file alone (no package import)"""
    src = open(os.path.join(ROOT, "comfyui-3d-pack_amd", "MVs_Algorithms", "DiffRastMesh", "diff_mesh.py")).read()
    ns = {"torch": torch, "F": torch.nn.functional}
    start = src.index("def laplacian_smooth_loss"); end = src.index("class DiffMesh:")
    exec(compile(src[start:end], "diff_mesh_regularisers", "exec"), ns)
    return ns["laplacian_smooth_loss"], ns["normal_consistency"]


def scene():
    sc = GR.scene()
    rng = np.random.default_rng(9)
    V = 3
    yy, xx = np.meshgrid(np.linspace(-1, 1, H), np.linspace(-1, 1, W), indexing="ij")
    imgs, masks = [], []
    for v in range(V):
        m = ((xx * 0.9) ** 2 + (yy * 1.1) ** 2 < (0.62 + 0.04 * v) ** 2).astype(np.float32)
        img = np.stack([0.5 + 0.45 * np.sin(4 * xx + v), 0.5 + 0.45 * np.cos(3 * yy + 2 * v), 0.3 + 0.1 * v + 0 * xx], -1).astype(np.float32)
        imgs.append(img); masks.append(m)
    sc["ref_images"], sc["ref_masks"] = np.stack(imgs), np.stack(masks)
    sc["poses"] = np.asarray([[2.0, -15.0, 20.0, 0, 0, 0], [2.1, 25.0, 140.0, 0, 0, 0], [1.9, 0.0, -100.0, 0, 0.03, 0]], np.float64)
    sc["raw_albedo_noise"] = (rng.normal(size=sc["albedo"].shape) * 0.2).astype(np.float32)
    return sc


class Recorder:
    trainer = None
    rows = []

    def __init__(self, total):
        pass

    def update_absolute(self, value, *a, **k):
        r = Recorder.trainer.renderer
        Recorder.rows.append([value, float(r.raw_albedo.detach().double().sum()), float(r.raw_albedo.detach().double().abs().sum()),
                              float(r.v_offsets.detach().double().sum()), float(r.v_offsets.detach().double().abs().sum())])


def reference_modules():
    GT.reference_trainer_module()                                    # stubs, CPU redirect, kiui restatements, image_utils, MS_SSIM = 0
    sys.modules["mesh_processer.mesh"].safe_normalize = lambda x, eps=1e-20: x / torch.sqrt(torch.clamp(torch.sum(x * x, -1, keepdim=True), min=eps))
    lap, ncons = regularisers()
    for name in ("kiui.mesh_utils",):
        m = G._Stub(name); m.__path__ = []
        sys.modules[name] = m
        setattr(sys.modules["kiui"], "mesh_utils", m)
    sys.modules["kiui.mesh_utils"].laplacian_smooth_loss, sys.modules["kiui.mesh_utils"].normal_consistency = lap, ncons
    sys.modules["comfy.utils"].ProgressBar = Recorder
    import fake_dr
    nv = types.ModuleType("nvdiffrast"); nv.torch = fake_dr; nv.__path__ = []
    sys.modules["nvdiffrast"], sys.modules["nvdiffrast.torch"] = nv, fake_dr
    pkg = types.ModuleType("ref_mesh_pkg"); pkg.__path__ = [os.path.join(G.REF, "MVs_Algorithms", "DiffRastMesh")]
    sys.modules["ref_mesh_pkg"] = pkg
    G._load("ref_mesh_pkg.diff_mesh_renderer", "MVs_Algorithms/DiffRastMesh/diff_mesh_renderer.py")
    return G._load("ref_mesh_pkg.diff_mesh", "MVs_Algorithms/DiffRastMesh/diff_mesh.py")


def generate():
    dm = reference_modules()
    sc = scene()
    out = {"scene_" + k: v for k, v in sc.items()}
    T = torch.from_numpy
    mesh = types.SimpleNamespace(v=T(sc["v"].copy()), f=T(sc["f"].copy()), vt=T(sc["vt"].copy()), ft=T(sc["f"].copy()), vn=T(sc["vn"].copy()),
                                 fn=T(sc["f"].copy()), albedo=T(sc["albedo"].copy()))
    a = ARGS
    t = object.__new__(dm.DiffMesh)                                   # __init__ :39-56 with the device on the CPU
    t.device = torch.device("cpu")
    t.train_mesh_geometry, t.remesh_after_n_iteration = a["train_mesh_geometry"], a["remesh_after_n_iteration"]
    t.renderer = dm.DiffRastRenderer(mesh, a["force_cuda_rasterize"])
    with torch.no_grad():
        t.renderer.raw_albedo.add_(T(sc["raw_albedo_noise"]))
    t.optimizer = torch.optim.Adam(t.renderer.get_params(a["texture_learning_rate"], a["train_mesh_geometry"], a["geometry_learning_rate"]))
    t.ms_ssim_loss = sys.modules["pytorch_msssim"].MS_SSIM()
    t.lambda_ssim, t.training_iterations, t.batch_size, t.invert_bg_prob = a["ms_ssim_loss_weight"], a["training_iterations"], a["batch_size"], a["invert_bg_prob"]
    t.prepare_training([T(x.copy()) for x in sc["ref_images"]], [T(x.copy()) for x in sc["ref_masks"]], [tuple(p) for p in sc["poses"]], FOVY)
    random.seed(SEEDS["python"]); np.random.seed(SEEDS["numpy"]); torch.manual_seed(SEEDS["torch"])
    Recorder.trainer, Recorder.rows = t, []
    t.training()
    out["trajectory"] = np.asarray(Recorder.rows, np.float64)
    out["final_raw_albedo"], out["final_v_offsets"] = t.renderer.raw_albedo.detach().numpy(), t.renderer.v_offsets.detach().numpy()
    out["final_mesh_v"], out["final_mesh_albedo"] = t.renderer.mesh.v.numpy(), t.renderer.mesh.albedo.numpy()     # after update_mesh()
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
    print(np.array2string(out["trajectory"], precision=5, suppress_small=True))


if __name__ == "__main__":
    main()
