"""The 3DGS trainer against a run of the REFERENCE'S OWN TRAINING LOOP.

tests/golden/ref_gs_train.npz holds the model after every one of 26 steps of the reference's GaussianSplatting3D.training
(main_3DGS.py:132-232: view sampling, camera controller, background choice, render glue, loss, Adam, learning-rate schedule, densify /
prune at steps 6, 12, 18, opacity reset at steps 9, 18), executed on the CPU in the build container over tests/fake_dgr.py (the CPU oracle's
forward and backward as one autograd function); tests/golden/make_golden_ref_gs_train.py says exactly what was replaced and why.  The
mirror's op-by-op trainer path runs here over the same stand-in, from the same seeds, and must follow the same trajectory step by step:
same point counts, same parameters.  (The fused HIP step is held to this op-by-op path by the GPU tests.)"""
import os
import random
import subprocess
import sys

import numpy as np
import pytest
import torch

GOLD_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
sys.path.insert(0, GOLD_DIR)


def test_fixture_is_what_the_reference_produces_now():
    if not os.path.isdir("/root/reference/MVs_Algorithms"):
        pytest.skip("/root/reference is not mounted here")
    r = subprocess.run([sys.executable, os.path.join(GOLD_DIR, "make_golden_ref_gs_train.py"), "--check"], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]


def test_trainer_follows_the_reference_training_loop_step_by_step(monkeypatch):
    import fake_dgr
    from make_golden_ref_gs_train import FOVY, PARAMS, SEEDS
    monkeypatch.setattr(fake_dgr, "RECORD", False)
    monkeypatch.setitem(sys.modules, "diff_gaussian_rasterization", fake_dgr)
    from MVs_Algorithms.GaussianSplatting.main_3DGS import GaussianSplatting3D, GSParams
    z = np.load(os.path.join(GOLD_DIR, "ref_gs_train.npz"))
    T = lambda k: torch.from_numpy(z["scene_" + k].copy())
    gp = GSParams()
    for k, v in PARAMS.items():
        assert hasattr(gp, k), k
        setattr(gp, k, v)
    init = dict(xyz=T("xyz"), features=torch.cat((T("f_dc"), T("f_rest")), dim=1), scaling_raw=T("scaling"), rotation_raw=T("rotation"),
                opacity_raw=T("opacity"), spatial_lr_scale=1.0)
    t = GaussianSplatting3D(gp, init, device="cpu")
    t.prepare_training([T("ref_images")[i] for i in range(4)], [T("ref_masks")[i] for i in range(4)], [tuple(p) for p in z["scene_poses"]], FOVY)
    np.testing.assert_allclose(t.ref_imgs_torch.numpy(), z["ref_imgs_torch"], atol=1e-7)
    np.testing.assert_allclose(t.ref_masks_torch.numpy(), z["ref_masks_torch"], atol=1e-7)
    g = t.renderer.gaussians
    rows = []

    def snapshot(value):
        lr = [grp["lr"] for grp in g.optimizer.param_groups if grp["name"] == "xyz"][0]
        rows.append([value, g._xyz.shape[0], float(g._xyz.detach().double().sum()), float(g._xyz.detach().double().abs().sum()),
                     float(g.get_opacity.detach().double().mean()), float(g.get_scaling.detach().double().mean()),
                     float(g._features_dc.detach().double().sum()), float(g._rotation.detach().double().abs().sum()), lr,
                     float(g.max_radii2D.double().sum()), float(g.denom.double().sum()), float(g.xyz_gradient_accum.double().sum())])
    random.seed(SEEDS["python"]); np.random.seed(SEEDS["numpy"]); torch.manual_seed(SEEDS["torch"])
    t.training(progress=snapshot)
    got, want = np.asarray(rows), z["trajectory"]
    cols = list(z["trajectory_columns"])
    assert got.shape == want.shape
    assert list(want[:, 1].astype(int)[[5, 6, 12, 18]]) == [160, 307, 579, 933]          # the densify steps of the reference run
    for s in range(want.shape[0]):
        assert got[s, 1] == want[s, 1], ("point count", s + 1, got[s, 1], want[s, 1])
        np.testing.assert_allclose(got[s], want[s], rtol=2e-5, atol=2e-5, err_msg="step %d, columns %s" % (s + 1, cols))
    long = {"xyz": "xyz", "features_dc": "f_dc", "features_rest": "f_rest", "scaling": "scaling", "rotation": "rotation", "opacity": "opacity"}
    P = g._param_dict()
    for k, name in long.items():
        np.testing.assert_allclose(P[name].detach().numpy(), z["final_" + k], rtol=1e-4, atol=2e-6, err_msg=k)


# ------------------------------------------------------------------------------------------------ the mesh trainer
def test_mesh_fixture_is_what_the_reference_produces_now():
    if not os.path.isdir("/root/reference/MVs_Algorithms"):
        pytest.skip("/root/reference is not mounted here")
    r = subprocess.run([sys.executable, os.path.join(GOLD_DIR, "make_golden_ref_mesh_train.py"), "--check"], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]


def test_mesh_trainer_follows_the_reference_training_loop_step_by_step(monkeypatch):
    """DiffMesh.training of the reference (diff_mesh.py:81-159: view sampling, controller, render, masked MSE + geometry regularisers,
    Adam over raw_albedo and v_offsets, update_mesh) ran 10 steps on the CPU over tests/fake_dr.py with the geometry training, i.e. with
    gradients through antialias / interpolate / rasterize; the mirror over the same stand-in must retrace it."""
    from types import SimpleNamespace
    import fake_dr
    from make_golden_ref_mesh_train import ARGS, FOVY, SEEDS
    from MVs_Algorithms.DiffRastMesh import diff_mesh as DM, diff_mesh_renderer as MR
    monkeypatch.setattr(MR, "dr", fake_dr)
    z = np.load(os.path.join(GOLD_DIR, "ref_mesh_train.npz"))
    T = lambda k: torch.from_numpy(z["scene_" + k].copy())
    mesh = SimpleNamespace(v=T("v"), f=T("f"), vt=T("vt"), ft=T("f"), vn=T("vn"), fn=T("f"), albedo=T("albedo"))
    t = DM.DiffMesh(mesh, device="cpu", **ARGS)
    with torch.no_grad():
        t.renderer.raw_albedo.add_(T("raw_albedo_noise"))
    t.prepare_training([T("ref_images")[i] for i in range(3)], [T("ref_masks")[i] for i in range(3)], [tuple(p) for p in z["scene_poses"]], FOVY)
    rows = []

    def snapshot(value):
        r = t.renderer
        rows.append([value, float(r.raw_albedo.detach().double().sum()), float(r.raw_albedo.detach().double().abs().sum()),
                     float(r.v_offsets.detach().double().sum()), float(r.v_offsets.detach().double().abs().sum())])
    random.seed(SEEDS["python"]); np.random.seed(SEEDS["numpy"]); torch.manual_seed(SEEDS["torch"])
    t.training(progress=snapshot)
    got, want = np.asarray(rows), z["trajectory"]
    assert got.shape == want.shape
    assert want[-1, 4] > 4.0                                           # the geometry really moved in the reference run
    np.testing.assert_allclose(got, want, rtol=2e-5, atol=2e-5)
    np.testing.assert_allclose(t.renderer.raw_albedo.detach().numpy(), z["final_raw_albedo"], rtol=5e-4, atol=2e-6)    # Adam on near-zero texel gradients amplifies float32 association differences: 3 of 768 texels at 2e-4
    np.testing.assert_allclose(t.renderer.v_offsets.detach().numpy(), z["final_v_offsets"], rtol=1e-4, atol=2e-5)     # d||Lv||/dv = Lv / ||Lv|| (the published loss is a norm, not its square) is ill-conditioned where the mesh is flat
    np.testing.assert_allclose(t.renderer.mesh.v.numpy(), z["final_mesh_v"], rtol=1e-5, atol=2e-5)            # update_mesh() at the end of training: v + v_offsets (tolerance of the offsets)
    np.testing.assert_allclose(t.renderer.mesh.albedo.numpy(), z["final_mesh_albedo"], rtol=5e-4, atol=2e-6)


def test_capture_restore_resumes_the_trajectory(monkeypatch):
    """GaussianModel.capture / restore (reference main_3DGS_renderer.py:255-288): a run interrupted after 11 steps -- past a densification and an
    opacity reset -- and resumed in a fresh trainer from the captured tuple ends exactly where the uninterrupted run ends."""
    import copy
    import fake_dgr
    from make_golden_ref_gs_train import FOVY, PARAMS, SEEDS
    monkeypatch.setattr(fake_dgr, "RECORD", False)
    monkeypatch.setitem(sys.modules, "diff_gaussian_rasterization", fake_dgr)
    from MVs_Algorithms.GaussianSplatting.main_3DGS import GaussianSplatting3D, GSParams
    z = np.load(os.path.join(GOLD_DIR, "ref_gs_train.npz"))
    T = lambda k: torch.from_numpy(z["scene_" + k].copy())
    gp = GSParams()
    for k, v in PARAMS.items():
        setattr(gp, k, v)

    def trainer():
        init = dict(xyz=T("xyz"), features=torch.cat((T("f_dc"), T("f_rest")), dim=1), scaling_raw=T("scaling"), rotation_raw=T("rotation"),
                    opacity_raw=T("opacity"), spatial_lr_scale=1.0)
        t = GaussianSplatting3D(gp, init, device="cpu")
        t.prepare_training([T("ref_images")[i] for i in range(4)], [T("ref_masks")[i] for i in range(4)], [tuple(p) for p in z["scene_poses"]], FOVY)
        return t
    views = [[s % 4, (s + 1) % 4] for s in range(16)]
    np.random.seed(1); torch.manual_seed(1)
    a = trainer()
    for s in range(16):
        a.training_step(s, views[s])
        if s == 10:
            state = copy.deepcopy(a.renderer.gaussians.capture())
            rng_np, rng_t = np.random.get_state(), torch.get_rng_state()
    assert len(state) == 12 and state[0] == 3 and state[1].shape[0] > 160 and "param_groups" in state[10]
    b = trainer()
    b.renderer.gaussians.restore(state, gp)
    b.optimizer = b.renderer.gaussians.optimizer
    g = b.renderer.gaussians
    b.params = [g._xyz, g._features_dc, g._features_rest, g._opacity, g._scaling, g._rotation]
    np.random.set_state(rng_np); torch.set_rng_state(rng_t)
    for s in range(11, 16):
        b.training_step(s, views[s])
    ga = a.renderer.gaussians
    assert g._xyz.shape == ga._xyz.shape
    for name in ("_xyz", "_features_dc", "_features_rest", "_opacity", "_scaling", "_rotation"):
        assert torch.equal(getattr(g, name).detach(), getattr(ga, name).detach()), name


# ------------------------------------------------------------------------------------------------ view-parallel training over gloo (world 2)
def _free_port():
    import socket
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _build_cpu_trainer(group, exchange):
    import fake_dgr
    from make_golden_ref_gs_train import FOVY, PARAMS
    fake_dgr.RECORD = False
    sys.modules["diff_gaussian_rasterization"] = fake_dgr
    from MVs_Algorithms.GaussianSplatting.main_3DGS import GaussianSplatting3D, GSParams
    z = np.load(os.path.join(GOLD_DIR, "ref_gs_train.npz"))
    T = lambda k: torch.from_numpy(z["scene_" + k].copy())
    gp = GSParams()
    for k, v in PARAMS.items():
        setattr(gp, k, v)
    gp.invert_bg_prob = 1.0                                           # a fixed background: ranks render different numbers of views per step
    init = dict(xyz=T("xyz"), features=torch.cat((T("f_dc"), T("f_rest")), dim=1), scaling_raw=T("scaling"), rotation_raw=T("rotation"),
                opacity_raw=T("opacity"), spatial_lr_scale=1.0)
    t = GaussianSplatting3D(gp, init, device="cpu", process_group=group, exchange=exchange)
    t.prepare_training([T("ref_images")[i] for i in range(4)], [T("ref_masks")[i] for i in range(4)], [tuple(p) for p in z["scene_poses"]], FOVY)
    return t


VIEWS = [[s % 4, (s + 2) % 4] for s in range(10)]
VIEWS4 = [[(s + i) % 4 for i in range(4)] for s in range(10)]      # world 4: a global batch of four views, one per rank


def _snap(t):
    g = t.renderer.gaussians
    return [q.detach().clone() for q in (g._xyz, g._features_dc, g._features_rest, g._opacity, g._scaling, g._rotation)]


def _rank_worker(rank, world, port, exchange, out):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    here = os.path.dirname(os.path.abspath(__file__))
    root = os.path.dirname(here)
    sys.path[:0] = [root, os.path.join(root, "comfyui-3d-pack_amd"), here, os.path.join(here, "golden")]
    np.random.seed(7); torch.manual_seed(7)
    t = _build_cpu_trainer(dist.group.WORLD, exchange)
    early = None
    for s, v in enumerate(VIEWS4 if world == 4 else VIEWS):
        t.training_step(s, v)
        if s == 5:
            early = _snap(t)
    out[rank] = (early, _snap(t), t.renderer.gaussians._xyz.shape[0])
    dist.destroy_process_group()


@pytest.mark.parametrize("exchange", ["allreduce", "allgather", "zero1"])
def test_two_gloo_ranks_train_like_one_process_and_stay_identical(exchange):
    """N > 1 on the CPU: two ranks, one view each per step, one gradient exchange per step.  Until the first densification the replicas
    follow the single-process run of the same global batch (mean-of-means = mean over equal shards); through densification -- whose
    statistics are combined over ranks and whose split samples come from a per-step seeded generator -- they stay bit-identical."""
    import torch.multiprocessing as mp
    world, port = 2, _free_port()
    out = mp.Manager().dict()
    mp.spawn(_rank_worker, args=(world, port, exchange, out), nprocs=world, join=True)
    (e0, f0, n0), (e1, f1, n1) = out[0], out[1]
    assert n0 == n1 and n0 > 160                                      # densified (step 6), same count on both ranks
    for a, b in zip(f0, f1):
        assert torch.equal(a, b)                                      # replicas bit-identical at the end
    for a, b in zip(e0, e1):
        assert torch.equal(a, b)
    np.random.seed(7); torch.manual_seed(7)
    t = _build_cpu_trainer(None, exchange)
    for s in range(6):
        t.training_step(s, VIEWS[s])
    for a, b in zip(_snap(t), e0):
        assert a.shape == b.shape
        torch.testing.assert_close(a, b, rtol=2e-5, atol=2e-6)


@pytest.mark.parametrize("exchange", ["allreduce", "zero1"])
def test_four_gloo_ranks_train_like_one_process_and_stay_identical(exchange):
    """The same at world 4 (nothing beyond two ranks was exercised at trainer level): a global batch of four views, one per rank; all-reduce (ring over
    four ranks) and ZeRO-1 (four parameter segments).  Replicas bit-identical through densification, and equal to the single-process run of the same global batch up
    to the first densification."""
    import torch.multiprocessing as mp
    world, port = 4, _free_port()
    out = mp.Manager().dict()
    mp.spawn(_rank_worker, args=(world, port, exchange, out), nprocs=world, join=True)
    e0, f0, n0 = out[0]
    assert n0 > 160
    for r in range(1, world):
        er, fr, nr = out[r]
        assert nr == n0
        for a, b in zip(f0, fr):
            assert torch.equal(a, b)
        for a, b in zip(e0, er):
            assert torch.equal(a, b)
    np.random.seed(7); torch.manual_seed(7)
    t = _build_cpu_trainer(None, exchange)
    for s in range(6):
        t.training_step(s, VIEWS4[s])
    for a, b in zip(_snap(t), e0):
        assert a.shape == b.shape
        torch.testing.assert_close(a, b, rtol=2e-5, atol=2e-6)


def _build_cpu_mesh_trainer(group, exchange):
    from types import SimpleNamespace
    import fake_dr
    from make_golden_ref_mesh_train import ARGS, FOVY
    from MVs_Algorithms.DiffRastMesh import diff_mesh as DM, diff_mesh_renderer as MR
    MR.dr = fake_dr
    z = np.load(os.path.join(GOLD_DIR, "ref_mesh_train.npz"))
    T = lambda k: torch.from_numpy(z["scene_" + k].copy())
    mesh = SimpleNamespace(v=T("v"), f=T("f"), vt=T("vt"), ft=T("f"), vn=T("vn"), fn=T("f"), albedo=T("albedo"))
    a = dict(ARGS); a["invert_bg_prob"] = 1.0
    t = DM.DiffMesh(mesh, device="cpu", process_group=group, exchange=exchange, **a)
    t.prepare_training([T("ref_images")[i] for i in range(3)], [T("ref_masks")[i] for i in range(3)], [tuple(p) for p in z["scene_poses"]], FOVY)
    return t


MESH_VIEWS = [[s % 3, (s + 1) % 3] for s in range(6)]


def _mesh_rank_worker(rank, world, port, exchange, out):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    here = os.path.dirname(os.path.abspath(__file__))
    root = os.path.dirname(here)
    sys.path[:0] = [root, os.path.join(root, "comfyui-3d-pack_amd"), here, os.path.join(here, "golden")]
    np.random.seed(3); torch.manual_seed(3)
    t = _build_cpu_mesh_trainer(dist.group.WORLD, exchange)
    for s, v in enumerate(MESH_VIEWS):
        t.training_step(s, v)
    out[rank] = (t.renderer.raw_albedo.detach().clone(), t.renderer.v_offsets.detach().clone())
    dist.destroy_process_group()


def test_two_gloo_ranks_fit_the_mesh_like_one_process():
    """the mesh trainer's view-parallel path: texture and vertex-offset gradients exchanged once per step, replicas identical and equal to
    the single-process run of the same global batch"""
    import torch.multiprocessing as mp
    world, port = 2, _free_port()
    out = mp.Manager().dict()
    mp.spawn(_mesh_rank_worker, args=(world, port, "allreduce", out), nprocs=world, join=True)
    (a0, o0), (a1, o1) = out[0], out[1]
    assert torch.equal(a0, a1) and torch.equal(o0, o1)
    np.random.seed(3); torch.manual_seed(3)
    t = _build_cpu_mesh_trainer(None, "allreduce")
    for s, v in enumerate(MESH_VIEWS):
        t.training_step(s, v)
    torch.testing.assert_close(t.renderer.raw_albedo.detach(), a0, rtol=2e-5, atol=2e-6)
    torch.testing.assert_close(t.renderer.v_offsets.detach(), o0, rtol=2e-4, atol=2e-6)
    assert float(o0.abs().max()) > 0
