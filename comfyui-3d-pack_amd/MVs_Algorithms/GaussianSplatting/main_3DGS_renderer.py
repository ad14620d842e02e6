"""GaussianModel (the parts the render / training step reads) and GaussianSplattingRenderer.

Host-side mirror of /root/reference/MVs_Algorithms/GaussianSplatting/main_3DGS_renderer.py for the hot path:
  get_expon_lr_func :21-43          GaussianModel activations :226-234, accessors :294-321
  training_setup (6 Adam groups, eps 1e-15) :435-453, update_learning_rate :455-461
  GaussianSplattingRenderer.__init__/initialize/render :783-949 (render = SURVEY 8a-a1)
Same public names, argument meaning and return dict; written from scratch, device-agnostic in construction (the
reference hard-codes "cuda"), and the optimizer is the fused HIP Adam.  Densify / prune / PLY I/O are SURVEY 8f rows,
not in this file yet.
"""
import math

import numpy as np
import torch
from torch import nn

from c3d_hip.ply import PlyData
from mesh_processer.mesh_utils import construct_list_of_gs_attributes, read_gs_ply, write_gs_ply
from shared_utils.sh_utils import RGB2SH, SH2RGB


def inverse_sigmoid(x):
    return torch.log(x / (1 - x))


def get_expon_lr_func(lr_init, lr_final, lr_delay_steps=0, lr_delay_mult=1.0, max_steps=1000000):
    """log-linear interpolation lr_init -> lr_final over max_steps, optional sine warm-up (reference :21-43)"""
    def helper(step):
        if lr_init == lr_final:
            return lr_init
        if step < 0 or (lr_init == 0.0 and lr_final == 0.0):
            return 0.0
        delay = 1.0
        if lr_delay_steps > 0:
            delay = lr_delay_mult + (1 - lr_delay_mult) * np.sin(0.5 * np.pi * np.clip(step / lr_delay_steps, 0, 1))
        t = np.clip(step / max_steps, 0, 1)
        return delay * np.exp(np.log(lr_init) * (1 - t) + np.log(lr_final) * t)
    return helper


class GaussianModel:
    """Parameter store: raw (pre-activation) tensors; the rasterizer consumes the activated accessors."""

    def __init__(self, sh_degree: int, device="cuda"):
        self.device = torch.device(device)
        self.active_sh_degree = 0
        self.max_sh_degree = sh_degree
        e = torch.empty(0)
        self.init_xyz = e
        self._xyz = self._features_dc = self._features_rest = self._scaling = self._rotation = self._opacity = e
        self.max_radii2D = self.xyz_gradient_accum = self.denom = e
        self.optimizer = None
        self.percent_dense = 0
        self.spatial_lr_scale = 0
        self.scaling_activation, self.scaling_inverse_activation = torch.exp, torch.log
        self.opacity_activation, self.inverse_opacity_activation = torch.sigmoid, inverse_sigmoid
        self.rotation_activation = torch.nn.functional.normalize

    # ---- accessors (reference :294-321) ----
    @property
    def get_gaussians_num(self):
        return self._xyz.shape[0]

    @property
    def get_scaling(self):
        return self.scaling_activation(self._scaling)

    @property
    def get_rotation(self):
        return self.rotation_activation(self._rotation)

    @property
    def get_xyz(self):
        return self._xyz

    @property
    def get_features(self):
        return torch.cat((self._features_dc, self._features_rest), dim=1)   # [N, K, 3]

    @property
    def get_opacity(self):
        return self.opacity_activation(self._opacity)

    def oneupSHdegree(self):
        if self.active_sh_degree < self.max_sh_degree:
            self.active_sh_degree += 1

    # ---- construction ----
    def _set(self, xyz, f_dc, f_rest, scaling_raw, rotation_raw, opacity_raw):
        mk = lambda a: nn.Parameter(torch.as_tensor(a, dtype=torch.float32).to(self.device).contiguous().requires_grad_(True))
        self._xyz, self._features_dc, self._features_rest = mk(xyz), mk(f_dc), mk(f_rest)
        self._scaling, self._rotation, self._opacity = mk(scaling_raw), mk(rotation_raw), mk(opacity_raw)
        self.init_xyz = self._xyz.detach().clone()
        self.max_radii2D = torch.zeros(self._xyz.shape[0], device=self.device)

    def create_from_arrays(self, xyz, colors_rgb, scales, spatial_lr_scale=1.0, opacity=0.1):
        """points + RGB colours + per-point isotropic scale (stand-in for create_from_pcd :407-433, whose kNN scale comes
        from simple_knn.distCUDA2 -- an init-only dependency that is out of scope)."""
        self.spatial_lr_scale = spatial_lr_scale
        xyz = torch.as_tensor(np.asarray(xyz), dtype=torch.float32)
        n, K = xyz.shape[0], (self.max_sh_degree + 1) ** 2
        feats = torch.zeros((n, K, 3))
        feats[:, 0] = RGB2SH(torch.as_tensor(np.asarray(colors_rgb), dtype=torch.float32))
        sc = torch.log(torch.as_tensor(np.asarray(scales), dtype=torch.float32).clamp_min(1e-7))
        if sc.ndim == 1:
            sc = sc[:, None].repeat(1, 3)
        rot = torch.zeros((n, 4)); rot[:, 0] = 1
        op = inverse_sigmoid(opacity * torch.ones((n, 1)))
        self._set(xyz, feats[:, :1], feats[:, 1:], sc, rot, op)

    def create_from_tensors(self, xyz, features, scaling_raw, rotation_raw, opacity_raw, spatial_lr_scale=1.0):
        """raw tensors in the layout create_from_ply produces (:486-498): features [N,K,3]"""
        self.spatial_lr_scale = spatial_lr_scale
        features = torch.as_tensor(features, dtype=torch.float32)
        self._set(xyz, features[:, :1], features[:, 1:], scaling_raw, rotation_raw, opacity_raw)
        self.active_sh_degree = self.max_sh_degree

    # ---- PLY wire format (reference to_ply :475-484, create_from_ply :486-498) ----
    def to_ply(self):
        n = lambda t: t.detach().cpu().numpy()
        xyz = n(self._xyz)
        f_dc = n(self._features_dc.detach().transpose(1, 2).flatten(start_dim=1).contiguous())      # channel-major
        f_rest = n(self._features_rest.detach().transpose(1, 2).flatten(start_dim=1).contiguous())
        return write_gs_ply(xyz, np.zeros_like(xyz), f_dc, f_rest, n(self._opacity), n(self._scaling), n(self._rotation),
                            construct_list_of_gs_attributes(self._features_dc, self._features_rest, self._scaling, self._rotation))

    def create_from_ply(self, plydata):
        xyz, f_dc, f_extra, opac, scales, rots = read_gs_ply(plydata)
        t = lambda a: torch.tensor(a, dtype=torch.float)
        self._set(t(xyz), t(f_dc).transpose(1, 2).contiguous(), t(f_extra).transpose(1, 2).contiguous(), t(scales), t(rots), t(opac))
        self.active_sh_degree = self.max_sh_degree

    # ---- optimisation (reference :435-461) ----
    def training_setup(self, training_args):
        self.percent_dense = training_args.percent_dense
        n = self.get_xyz.shape[0]
        self.xyz_gradient_accum = torch.zeros((n, 1), device=self.device)
        self.denom = torch.zeros((n, 1), device=self.device)
        groups = [
            {'params': [self._xyz], 'lr': training_args.position_lr_init * self.spatial_lr_scale, "name": "xyz"},
            {'params': [self._features_dc], 'lr': training_args.feature_lr, "name": "f_dc"},
            {'params': [self._features_rest], 'lr': training_args.feature_lr / 20.0, "name": "f_rest"},
            {'params': [self._opacity], 'lr': training_args.opacity_lr, "name": "opacity"},
            {'params': [self._scaling], 'lr': training_args.scaling_lr, "name": "scaling"},
            {'params': [self._rotation], 'lr': training_args.rotation_lr, "name": "rotation"},
        ]
        if self.device.type == "cuda":
            from c3d_hip.optim import FusedAdam
            self.optimizer = FusedAdam(groups, lr=0.0, eps=1e-15)
        else:   # host-logic tests only; rendering on CPU is impossible anyway
            self.optimizer = torch.optim.Adam(groups, lr=0.0, eps=1e-15)
        self.xyz_scheduler_args = get_expon_lr_func(lr_init=training_args.position_lr_init * self.spatial_lr_scale,
                                                    lr_final=training_args.position_lr_final * self.spatial_lr_scale,
                                                    lr_delay_mult=training_args.position_lr_delay_mult,
                                                    max_steps=training_args.position_lr_max_steps)

    def update_learning_rate(self, iteration):
        for group in self.optimizer.param_groups:
            if group["name"] == "xyz":
                group['lr'] = lr = self.xyz_scheduler_args(iteration)
                return lr


class GaussianSplattingRenderer:
    def __init__(self, sh_degree=3, white_background=True, radius=1, device="cuda"):
        self.sh_degree, self.white_background, self.radius = sh_degree, white_background, radius
        self.device = torch.device(device)
        self.gaussians = GaussianModel(sh_degree, device=device)
        self.bg_color = torch.tensor([1, 1, 1] if white_background else [0, 0, 0], dtype=torch.float32, device=self.device)
        self.force_unfused = False   # True: take the reference's op-by-op accessor path (tests compare both)

    def initialize(self, input=None, num_pts=5000, radius=0.5):
        """input None -> the reference's random ball (:811-826): r = radius * cbrt(U), colours U/255, lr scale 10."""
        if input is None:
            phis = np.random.random((num_pts,)) * 2 * np.pi
            thetas = np.arccos(np.random.random((num_pts,)) * 2 - 1)
            r = radius * np.cbrt(np.random.random((num_pts,)))
            xyz = np.stack((r * np.sin(thetas) * np.cos(phis), r * np.sin(thetas) * np.sin(phis), r * np.cos(thetas)), axis=1)
            shs = np.random.random((num_pts, 3)) / 255.0
            # mean nearest-neighbour spacing of a uniform ball as the isotropic scale (distCUDA2 stand-in)
            spacing = radius * (4.0 / 3.0 * np.pi / max(num_pts, 1)) ** (1.0 / 3.0)
            self.gaussians.create_from_arrays(xyz, SH2RGB(shs), np.full((num_pts,), spacing), spatial_lr_scale=10)
        elif isinstance(input, PlyData):
            self.gaussians.create_from_ply(input)
        elif isinstance(input, dict):
            self.gaussians.create_from_tensors(**input)
        else:
            raise TypeError("initialize(): pass None, a GS PlyData or a dict of raw tensors (mesh / point-cloud initialisers need simple_knn: out of scope)")

    def render(self, viewpoint_camera, scaling_modifier=1.0, gaussain_idx=None, bg_color=None, override_color=None,
               compute_cov3D_python=False, convert_SHs_python=False):
        """-> dict(image[3,H,W] clamped, depth[1,H,W], alpha[1,H,W], viewspace_points[N,3], visibility_filter[N], radii[N])"""
        from diff_gaussian_rasterization import GaussianRasterizationSettings, GaussianRasterizer
        if compute_cov3D_python or convert_SHs_python:
            raise NotImplementedError("python SH / covariance paths are dead code in the reference (SURVEY 3.1 note a)")
        g = self.gaussians
        settings = GaussianRasterizationSettings(
            image_height=int(viewpoint_camera.image_height), image_width=int(viewpoint_camera.image_width),
            tanfovx=math.tan(viewpoint_camera.FoVx * 0.5), tanfovy=math.tan(viewpoint_camera.FoVy * 0.5),
            bg=self.bg_color if bg_color is None else bg_color, scale_modifier=scaling_modifier,
            viewmatrix=viewpoint_camera.world_view_transform, projmatrix=viewpoint_camera.full_proj_transform,
            sh_degree=g.active_sh_degree, campos=viewpoint_camera.camera_center, prefiltered=False, debug=False)
        fused = (gaussain_idx is None and override_color is None and g.max_sh_degree == 3 and g._xyz.is_cuda and not self.force_unfused)
        xyz = g.get_xyz if gaussain_idx is None else g.get_xyz[gaussain_idx]
        # zero tensor whose gradient is the screen-space positional gradient (densification statistic)
        screenspace_points = torch.zeros_like(xyz, dtype=xyz.dtype, requires_grad=True, device=xyz.device) + 0
        try:
            screenspace_points.retain_grad()
        except Exception:
            pass
        if fused:
            # same result as the accessor path below; exp / sigmoid / normalize / cat are folded into the projection kernels
            from diff_gaussian_rasterization import rasterize_gaussians_raw
            image, radii, depth, alpha = rasterize_gaussians_raw(xyz, screenspace_points, g._features_dc, g._features_rest, g._opacity,
                                                                 g._scaling, g._rotation, settings)
        else:
            feats, opac, scales, rots = g.get_features, g.get_opacity, g.get_scaling, g.get_rotation
            if gaussain_idx is not None:
                feats, opac, scales, rots = feats[gaussain_idx], opac[gaussain_idx], scales[gaussain_idx], rots[gaussain_idx]
            shs, colors = (feats, None) if override_color is None else (None, override_color)
            image, radii, depth, alpha = GaussianRasterizer(raster_settings=settings)(
                means3D=xyz, means2D=screenspace_points, shs=shs, colors_precomp=colors, opacities=opac,
                scales=scales, rotations=rots, cov3D_precomp=None)
        return {"image": image.clamp(0, 1), "depth": depth, "alpha": alpha, "viewspace_points": screenspace_points,
                "visibility_filter": radii > 0, "radii": radii}
