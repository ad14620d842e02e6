"""Replay of the reference-loop fixture through the HIP kernels.

tests/golden/ref_gs_train.npz is a 26-step run of the reference's own training loop over the CPU oracle; tests/test_ref_train_loop.py
replays it on the CPU.  This file replays it on the GPU -- the mirror's trainer with the HIP rasterizer, op-by-op and fused -- and compares
the trajectories.  The two rasterizers agree to ~1e-5 per image, so the trajectories agree closely until a densification decision sits
on a threshold; the tolerances below allow for that.

NOT YET RUN ON HARDWARE: written after round 1's GPU budget was spent.  It is therefore opt-in (C3D_RUN_REPLAY=1) and skipped in the
default `pytest -m gpu` run; enable it once, adjust the tolerances to what the hardware shows, then drop the switch."""
import os
import random
import sys

import numpy as np
import pytest
import torch

pytestmark = [pytest.mark.gpu, pytest.mark.skipif(os.environ.get("C3D_RUN_REPLAY") != "1", reason="opt-in until first validated on hardware")]
GOLD_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
sys.path.insert(0, GOLD_DIR)


@pytest.mark.parametrize("fused", [False, True])
def test_hip_trainer_replays_the_reference_loop(fused):
    from make_golden_ref_gs_train import FOVY, PARAMS, SEEDS
    from MVs_Algorithms.GaussianSplatting.main_3DGS import GaussianSplatting3D, GSParams
    z = np.load(os.path.join(GOLD_DIR, "ref_gs_train.npz"))
    T = lambda k: torch.from_numpy(z["scene_" + k].copy())
    gp = GSParams()
    for k, v in PARAMS.items():
        setattr(gp, k, v)
    if fused:
        gp.invert_bg_prob = 1.0                                       # the fused step needs a fixed background; compare only its own invariants
    init = dict(xyz=T("xyz"), features=torch.cat((T("f_dc"), T("f_rest")), dim=1), scaling_raw=T("scaling"), rotation_raw=T("rotation"),
                opacity_raw=T("opacity"), spatial_lr_scale=1.0)
    t = GaussianSplatting3D(gp, init, device="cuda")
    t.use_fused_step = fused
    t.prepare_training([T("ref_images")[i] for i in range(4)], [T("ref_masks")[i] for i in range(4)], [tuple(p) for p in z["scene_poses"]], FOVY)
    g = t.renderer.gaussians
    rows = []

    def snapshot(value):
        rows.append([value, g._xyz.shape[0], float(g._xyz.detach().double().sum()), float(g._xyz.detach().double().abs().sum()),
                     float(g.get_opacity.detach().double().mean()), float(g.get_scaling.detach().double().mean())])
    random.seed(SEEDS["python"]); np.random.seed(SEEDS["numpy"]); torch.manual_seed(SEEDS["torch"])
    t.training(progress=snapshot)
    got, want = np.asarray(rows), z["trajectory"][:, :6]
    assert got.shape == want.shape and np.isfinite(got).all()
    if fused:
        assert got[5, 1] == 160 and got[-1, 1] > 300                  # same schedule: nothing before step 6, growth afterwards
        return
    # op-by-op path, same seeds: identical until the first densification (step 6) up to float32 rasterizer differences ...
    np.testing.assert_allclose(got[:6], want[:6], rtol=1e-3, atol=1e-3)
    # ... then the same schedule; a handful of threshold decisions may differ (the split samples come from the device generator)
    assert np.all(np.abs(got[:, 1] - want[:, 1]) <= 0.05 * want[:, 1])
    np.testing.assert_allclose(got[:, 4:6], want[:, 4:6], rtol=0.05, atol=5e-3)      # mean opacity / scaling follow the resets and the splits
