"""3DGS trainer: GSParams, GaussianSplattingCameraController, GaussianSplatting3D.

Host-side mirror of /root/reference/MVs_Algorithms/GaussianSplatting/main_3DGS.py: GSParams defaults :15-42,
controller :76-82, training loop :129-232 (per step: LR update :154, `batch_size` random views :158-174, loss =
(1-l_ssim) L1 + l_alpha MSE(alpha, mask) + l_ssim (1 - MS-SSIM) :184-192, backward + Adam :205-207).
Differences, all additive: device is a parameter; the optimizer is the fused HIP Adam; `process_group` shards the
step's views over ranks with one gradient exchange (c3d_hip/parallel.py); on a HIP device the per-view render loop of a step is
replaced by the fused multi-view step (c3d_hip/gs_step.py) for every loss configuration, with the same per-view background draws.
"""
import random
from dataclasses import dataclass

import numpy as np
import torch
import torch.nn.functional as F

from c3d_hip import parallel
from shared_utils.camera_utils import BaseCameraController, MiniCam, get_projection_matrix, orbit_camera
from shared_utils.msssim import MS_SSIM
from .main_3DGS_renderer import GaussianSplattingRenderer


@dataclass
class GSParams:
    # training
    training_iterations: int = 30_000
    batch_size: int = 1
    lambda_ssim: float = 0.2
    lambda_alpha: float = 3
    lambda_offset: float = 0
    lambda_offset_opacity: float = 0
    invert_bg_prob: float = 0.5
    # learning rates
    feature_lr: float = 0.0025
    opacity_lr: float = 0.05
    scaling_lr: float = 0.005
    rotation_lr: float = 0.001
    position_lr_init: float = 0.00016
    position_lr_final: float = 0.0000016
    position_lr_delay_mult: float = 0.01
    position_lr_max_steps: int = 30_000
    # densify / prune
    num_pts: int = 10_000
    K: int = 3
    percent_dense: float = 0.01
    density_start_iter: int = 500
    density_end_iter: int = 15_000
    densification_interval: int = 100
    opacity_reset_interval: int = 3000
    densify_grad_threshold: float = 0.0002
    # model
    sh_degree: int = 3


class GaussianSplattingCameraController(BaseCameraController):
    def post_init(self):
        self.projection_matrix = get_projection_matrix(self.cam.near, self.cam.far, self.cam.fovx, self.cam.fovy).transpose(0, 1).to(self.device)

    def get_render_result(self, render_pose, bg_color, **kwargs):
        cam = MiniCam(render_pose, self.cam.W, self.cam.H, self.cam.fovy, self.cam.fovx, self.cam.near, self.cam.far,
                      self.projection_matrix, device=self.device)
        return self.renderer.render(cam, bg_color=bg_color, **kwargs)

    def render_all_pose(self, all_cam_poses, **kwargs):
        """orbit rendering (reference camera_utils.py:160-175 via the renderer nodes): without autograd and without per-view options the whole
        orbit goes through ONE batched library call; otherwise the reference's per-view loop."""
        g = self.renderer.gaussians
        if kwargs or torch.is_grad_enabled() or not g._xyz.is_cuda or len(all_cam_poses) == 0 or not self.renderer.raw_storage_ok():
            # (a storage the raw-parameter kernels do not take -- SH degree > 3, odd f_rest shapes -- goes view by view through render(); without autograd every such call
            #  takes the rasterizer's synchronous, exact path: the images that leave here are complete)
            return super().render_all_pose(all_cam_poses, **kwargs)
        cams, bgs = [], []
        for radius, elevation, azimuth, cx, cy, cz in all_cam_poses:
            pose = orbit_camera(elevation, azimuth, radius, target=np.array([cx, cy, cz], dtype=np.float32))
            cams.append(MiniCam(pose, self.cam.W, self.cam.H, self.cam.fovy, self.cam.fovx, self.cam.near, self.cam.far, self.projection_matrix, device=self.device))
            bgs.append(self.static_bg if self.static_bg is not None else (self.white_bg if np.random.rand() > self.invert_bg_prob else self.black_bg))
        out = self.renderer.render_views(cams, bgs)
        return out["image"], out["alpha"], out


def _fit(img, H, W, device):
    """[1,H0,W0,C] in [0,1] -> [1,C,H,W] on device (bilinear when the size differs; image_utils.py:8-14)"""
    t = img.permute(0, 3, 1, 2).contiguous().float().to(device)
    return t if t.shape[-2:] == (H, W) else F.interpolate(t, (H, W), mode="bilinear", align_corners=False)


class GaussianSplatting3D:
    def __init__(self, gs_params=None, init_input=None, device='cuda', process_group=None, exchange="allreduce"):
        self.device = torch.device(device)
        self.gs_params = gs_params = gs_params or GSParams()
        self.group, self.exchange = process_group, exchange
        self.exchange_chunks = 1      # > 1 (fused step, world > 1, "allreduce"): overlap the gradient exchange with the per-Gaussian backward pass, range by range
        self.renderer = GaussianSplattingRenderer(sh_degree=gs_params.sh_degree, device=device)
        self.renderer.initialize(init_input, num_pts=gs_params.num_pts)
        g = self.renderer.gaussians
        g.training_setup(gs_params)
        g.active_sh_degree = g.max_sh_degree          # no progressive SH level, as the reference
        self.optimizer = g.optimizer
        self.ms_ssim_loss = MS_SSIM(data_range=1, size_average=True, channel=3)
        self.params = [g._xyz, g._features_dc, g._features_rest, g._opacity, g._scaling, g._rotation]
        parallel.broadcast_parameters(self.params, src=0, group=process_group)
        self.use_fused_step, self._step = True, None       # the fused rasterizer step serves every loss configuration (see _can_fuse)
        self._zero = None                                    # exchange == "zero1": parallel.ZeroOneAdam (sharded moments; rebuilt after every densification)
        self.defer_step_status = True                        # fused step: no host synchronisation per step once the pair capacity is fitted (c3d_hip/gs_step.py)
        self.image_loss_in_torch = False                     # True: fused forward / backward halves with the image loss (incl. MS-SSIM) as torch ops in between
        self.fused_densify_stats = True                      # densification statistics of the step's last view by one kernel from the step's workspace (False: read_view + torch ops)

    def prepare_training(self, reference_images, reference_masks, reference_orbit_camera_poses, reference_orbit_camera_fovy):
        self.ref_imgs_num = len(reference_images)
        self.ref_size_H, self.ref_size_W = reference_images[0].shape[0], reference_images[0].shape[1]
        self.cam_controller = GaussianSplattingCameraController(self.renderer, self.ref_size_W, self.ref_size_H,
                                                                reference_orbit_camera_fovy, self.gs_params.invert_bg_prob, None, self.device)
        self.all_ref_cam_poses = reference_orbit_camera_poses
        self._ref_cams = {}          # view index -> MiniCam of that reference pose (fused step): the poses are fixed for the run
        H, W = self.ref_size_H, self.ref_size_W
        self.ref_imgs_torch = torch.cat([_fit(im.unsqueeze(0), H, W, self.device) for im in reference_images], dim=0)            # [V,3,H,W]
        self.ref_masks_torch = torch.cat([_fit(m.unsqueeze(2).unsqueeze(0), H, W, self.device) for m in reference_masks], dim=0)  # [V,1,H,W]

    def training_step(self, step, view_indices):
        """one optimisation step over `view_indices` (the GLOBAL batch; this rank renders its shard). -> loss value tensor"""
        p = self.gs_params
        world = torch.distributed.get_world_size(self.group) if (torch.distributed.is_available() and torch.distributed.is_initialized()) else 1
        rank = torch.distributed.get_rank(self.group) if world > 1 else 0
        self.renderer.gaussians.update_learning_rate(step)
        mine = parallel.shard_views(view_indices, rank, world)
        if self._can_fuse():
            return self._fused_step(mine, len(view_indices), world, step)
        imgs, refs, alphas, masks = [], [], [], []
        out = None
        for i in mine:
            out = self.cam_controller.render_at_pose(self.all_ref_cam_poses[i])
            m = self.ref_masks_torch[i]
            imgs.append((out["image"] * m).unsqueeze(0)); refs.append((self.ref_imgs_torch[i] * m).unsqueeze(0))
            alphas.append(out["alpha"].unsqueeze(0)); masks.append(m.unsqueeze(0))
        imgs, refs, alphas, masks = torch.cat(imgs), torch.cat(refs), torch.cat(alphas), torch.cat(masks)
        loss = (1 - p.lambda_ssim) * F.l1_loss(imgs, refs) + p.lambda_alpha * F.mse_loss(alphas, masks)
        if p.lambda_ssim > 0:
            loss = loss + p.lambda_ssim * (1 - self.ms_ssim_loss(refs, imgs))
        g = self.renderer.gaussians
        if p.lambda_offset > 0 or p.lambda_offset_opacity > 0:
            off = (g.init_xyz - g._xyz).norm(dim=-1, keepdim=True)
            if p.lambda_offset > 0:
                loss = loss + p.lambda_offset * off.mean()
            if p.lambda_offset_opacity > 0:
                loss = loss + p.lambda_offset_opacity * (off.detach() * g.get_opacity).mean()
        loss.backward()
        # batch losses are means over views: equal shards => the global gradient is the mean of the ranks' gradients
        if self.exchange == "zero1":
            z = self._zero_adam()
            for dst, q in zip(z.grads, self.params):
                dst.copy_(q.grad) if q.grad is not None else dst.zero_()
            z.step()                                           # reduce-scatter -> Adam on the owned slice -> all-gather(parameters)
        else:
            parallel.exchange_gradients(self.params, self.group, self.exchange, average=True)
            self.optimizer.step()
        stats = None
        if out is not None and self._in_density_window(step):
            vg = out["viewspace_points"].grad
            stats = (out["radii"], vg if vg is not None else torch.zeros_like(out["viewspace_points"]))
        self.optimizer.zero_grad()
        self._densify(step, stats)
        return loss.detach()

    def _zero_adam(self):
        if self._zero is None:
            self._zero = parallel.ZeroOneAdam(self.optimizer, self.params, self.group, average=True)
        return self._zero

    # ---- densify / prune schedule (reference :209-224) ----
    def _in_density_window(self, step):
        p = self.gs_params
        return p.density_start_iter <= step <= p.density_end_iter

    def _densify(self, step, stats, stats_done=False):
        """stats = (radii [N], dL/dmeans2D [N,3]) of this rank's LAST view of the step (the reference looks at the last view only, :210-213),
        or None when the rank rendered nothing.  With several ranks the statistics are combined (sum / max) so that replicas stay identical.
        stats_done: the fused step has already accumulated them (FusedViewStep.accumulate_densify_stats)."""
        p, g = self.gs_params, self.renderer.gaussians
        if not self._in_density_window(step):
            return
        n = g._xyz.shape[0]
        if not stats_done:
            if stats is None:
                radii, vg = torch.zeros((n,), dtype=torch.int32, device=self.device), torch.zeros((n, 3), device=self.device)
            else:
                radii, vg = stats
            vis = radii > 0
            if self.group is not None and torch.distributed.get_world_size(self.group) > 1:
                add = torch.cat((torch.norm(vg[:, :2], dim=-1, keepdim=True) * vis.unsqueeze(-1), vis.unsqueeze(-1).float()), dim=1)
                torch.distributed.all_reduce(add, group=self.group)
                rmax = torch.where(vis, radii, torch.zeros_like(radii)).float()
                torch.distributed.all_reduce(rmax, op=torch.distributed.ReduceOp.MAX, group=self.group)
                g.xyz_gradient_accum += add[:, :1]
                g.denom += add[:, 1:]
                g.max_radii2D = torch.maximum(g.max_radii2D, rmax)
            else:
                g.add_densification_stats(vg, vis, radii)
        changed = False
        if self._zero is not None and (step % p.densification_interval == 0 or step % p.opacity_reset_interval == 0):
            self._zero.unshard()                             # the surgery below edits whole moments (torch.optim layout); a new ZeroOneAdam adopts them afterwards
            self._zero = None
        if step % p.densification_interval == 0:
            gen = None                                     # one process: torch's global generator, exactly as the reference draws
            if self.group is not None and torch.distributed.get_world_size(self.group) > 1:
                gen = torch.Generator(device=self.device)
                gen.manual_seed(0x3D65 + step)             # identical draws on every rank
            self.last_densify = g.densify_and_prune(p.densify_grad_threshold, min_opacity=0.005, extent=4, max_screen_size=1, generator=gen)
            changed = True
        if step % p.opacity_reset_interval == 0:
            g.reset_opacity()
            changed = True
        if changed:
            self.params = [g._xyz, g._features_dc, g._features_rest, g._opacity, g._scaling, g._rotation]
            if self._step is not None:
                self._step.finish()                          # a deferred step's status is examined BEFORE the object goes away: an overflow on the last step is still reported
            self._step = None                                # pair buffers / gradient buffers are sized by N

    # ---- fused multi-view step: all of this rank's views in sync-free library calls ----
    def _can_fuse(self):
        """Every loss the reference's trainer can be configured with goes through the fused rasterizer step (round 2): L1, alpha MSE and
        MS-SSIM inside c3d_gs_train_views_raw (`image_loss_in_torch` switches to the forward / backward halves with torch's loss in between).
        Backgrounds are per view, drawn exactly as BaseCameraController.render_at_pose draws them."""
        g = self.renderer.gaussians
        return self.use_fused_step and self.device.type == "cuda" and 0 <= g.max_sh_degree <= 3

    def _image_loss(self, colors, alphas, mine):
        """the reference's batch loss on stacked tensors (main_3DGS.py:169-192): masked images, L1 + alpha MSE + (1 - MS-SSIM)"""
        p = self.gs_params
        m = self.ref_masks_torch[mine]
        imgs, refs = colors.clamp(0, 1) * m, self.ref_imgs_torch[mine] * m         # render() returns image.clamp(0, 1)
        loss = (1 - p.lambda_ssim) * F.l1_loss(imgs, refs) + p.lambda_alpha * F.mse_loss(alphas, m)
        if p.lambda_ssim > 0:
            loss = loss + p.lambda_ssim * (1 - self.ms_ssim_loss(refs, imgs))
        return loss

    def _fused_step(self, mine, global_batch, world, step=-1):
        import math
        from c3d_hip.gs_step import FusedViewStep
        from diff_gaussian_rasterization import GaussianRasterizationSettings
        p, ctl, H, W = self.gs_params, self.cam_controller, self.ref_size_H, self.ref_size_W
        g = self.renderer.gaussians
        if self._step is None:
            self._step = FusedViewStep(self.params[0].shape[0], H, W, self.device)
            self._step.defer_status = self.defer_step_status
            self._step.status_sync = parallel.status_max(self.group) if world > 1 else None      # every rank decides about a step (fit, regrow, redo) from the same status words
            if self.exchange == "zero1":
                self._flat_grads, self._step_grads = None, self._zero_adam().grads      # the kernels write into what the all-to-all sends
            else:
                self._flat_grads = parallel.FlatGrads(self.params)       # the kernels write into what the collective sends
                self._step_grads = self._flat_grads.views
        views = []
        for i in mine:
            # The reference builds a MiniCam per render (camera_utils.py:253-262): three matrices uploaded and one 4x4 product on the device, every iteration, for
            # a pose that never changes.  At the node's default scene size those six launches are 8 % of an iteration: built once per reference view here.
            if getattr(self, "_ref_cams", None) is None:
                self._ref_cams = {}
            cam = self._ref_cams.get(i)
            if cam is None:
                radius, elev, azim, cx, cy, cz = self.all_ref_cam_poses[i]
                cam = MiniCam(orbit_camera(elev, azim, radius, target=np.array([cx, cy, cz], dtype=np.float32)), W, H, ctl.cam.fovy, ctl.cam.fovx,
                              ctl.cam.near, ctl.cam.far, ctl.projection_matrix, device=self.device)
                # MiniCam keeps the reference's transposed VIEW of w2c: handing that to the rasterizer costs a strided-copy launch per view and iteration -- one of the ~25
                # launches of an iteration at the node's default scene size.  Same values, contiguous float32, made once per reference pose.
                cam.world_view_transform = cam.world_view_transform.float().contiguous()
                cam.full_proj_transform = cam.full_proj_transform.float().contiguous()
                cam.camera_center = cam.camera_center.float().contiguous()
                self._ref_cams[i] = cam
            # the background of this view: one np.random draw per view, in view order, as render_at_pose (camera_utils.py:246-249)
            bg = ctl.static_bg if ctl.static_bg is not None else (ctl.white_bg if np.random.rand() > ctl.invert_bg_prob else ctl.black_bg)
            views.append(GaussianRasterizationSettings(H, W, math.tan(cam.FoVx * 0.5), math.tan(cam.FoVy * 0.5), bg, 1.0,
                                                       cam.world_view_transform, cam.full_proj_transform, g.active_sh_degree, cam.camera_center, False, False))
        n_mine = max(len(mine), 1)
        overlapped = False
        plist = [q.detach() for q in self.params]
        if len(mine) == 0 or not self.image_loss_in_torch:
            # The whole image loss of main_3DGS.py:184-192 inside ONE library call: loss value and dL/dimage of the L1 / alpha-MSE terms come out of
            # a fused pixel pass, the MS-SSIM term (weight lambda_ssim, include/c3d_loss.h) from the HIP multi-scale SSIM kernels on the view's lane.
            # (c - ref) * mask: the unmasked reference is the target and the mask the pixel weight, as the reference's (c * m - ref * m).
            # world > 1, all-reduce mode, exchange_chunks > 1: the per-Gaussian backward pass goes by Gaussian ranges and each range's f_rest rows
            # start their all-reduce underneath the next range's kernels (FlatGrads.exchange_rows); the rest follows in exchange_finish below
            overlapped = world > 1 and self.exchange == "allreduce" and self.exchange_chunks > 1
            if self.exchange == "zero1":
                self._step_grads = self._zero_adam().grads            # (re-pointed after a densification rebuilt the object)
            loss = self._step.run(views, plist, self._step_grads, [self.ref_imgs_torch[i].contiguous() for i in mine],
                                  [self.ref_masks_torch[i].contiguous() for i in mine], [self.ref_masks_torch[i].contiguous() for i in mine],
                                  w_l1=1.0 - p.lambda_ssim, w_l2=0.0, w_alpha_mse=p.lambda_alpha, scale=1.0 / n_mine, w_ssim=p.lambda_ssim,
                                  accumulate=False,      # every gradient is written exactly once: no zero-fill
                                  param_chunks=(self.exchange_chunks if overlapped else 1),
                                  after_chunk=((lambda g0, g1: self._flat_grads.exchange_rows(g0, g1, self.group)) if overlapped else None))
        else:
            # the same step split at the image: forward all views -> torch differentiates the image loss -> backward all views (any loss torch can express)
            colors, _, alphas, _ = self._step.forward(views, plist)
            colors.requires_grad_(True); alphas.requires_grad_(True)
            with torch.enable_grad():
                loss = self._image_loss(colors, alphas, list(mine))
                dcolor, dalpha = torch.autograd.grad(loss, [colors, alphas])
            self._step.backward(self._step_grads, dcolor, dalpha, accumulate=False)
            loss = loss.detach()
        if p.lambda_offset > 0 or p.lambda_offset_opacity > 0:
            # parameter-space regularisers (main_3DGS.py:194-203): plain autograd, added on top of what the kernels wrote
            with torch.enable_grad():
                off = (g.init_xyz - g._xyz).norm(dim=-1, keepdim=True)
                reg = p.lambda_offset * off.mean() if p.lambda_offset > 0 else 0.0
                if p.lambda_offset_opacity > 0:
                    reg = reg + p.lambda_offset_opacity * (off.detach() * g.get_opacity).mean()
                gx, go = torch.autograd.grad(reg, [g._xyz, g._opacity], allow_unused=True)
            if gx is not None:
                self._step_grads[0].add_(gx)
            if go is not None:
                self._step_grads[3].add_(go)
            loss = loss + reg.detach()
        if self.exchange == "zero1":
            self._zero_adam().step()
        else:
            if overlapped:
                self._flat_grads.exchange_finish(self.group, average=True)
            else:
                self._flat_grads.exchange(self.group, self.exchange, average=True)
            for q, gq in zip(self.params, self._step_grads):
                q.grad = gq
            self.optimizer.step()
        step_obj = self._step
        for q in self.params:
            q.grad = None
        if self._in_density_window(step):
            g = self.renderer.gaussians
            single = self.group is None or torch.distributed.get_world_size(self.group) <= 1
            if self.fused_densify_stats and single and len(mine) and g.max_radii2D.dtype == torch.float32 and g.max_radii2D.numel() == g._xyz.shape[0]:
                # one process: the statistics of the step's last view are accumulated by ONE kernel straight from the step's workspace (the node's default run is
                # bound by its launch count: read_view + add_densification_stats were two copies and nine torch launches per iteration)
                step_obj.accumulate_densify_stats(len(mine) - 1, g.xyz_gradient_accum, g.denom, g.max_radii2D)
                self._densify(step, None, stats_done=True)
            else:
                self._densify(step, step_obj.read_view(len(mine) - 1) if len(mine) else None)
        return loss.detach()

    def training(self, progress=None):
        rng = random.Random(0) if self.group is not None else random   # ranks must draw the same views
        for step in range(self.gs_params.training_iterations):
            idx = [rng.randint(0, self.ref_imgs_num - 1) for _ in range(self.gs_params.batch_size)]
            self.training_step(step, idx)
            if progress is not None:
                progress(step + 1)
        if self._step is not None:
            self._step.finish()
        if self._zero is not None:      # ZeRO-1: hand the sharded Adam moments back to the optimizer (capture() / state_dict() after training must see them)
            self._zero.unshard()
            self._zero = None
        if self.device.type == "cuda":
            torch.cuda.synchronize(self.device)
        self.need_update = True
