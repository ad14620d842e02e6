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
from typing import NamedTuple

import numpy as np
import torch
from torch import nn

from c3d_hip.ply import PlyData
from mesh_processer.mesh_utils import construct_list_of_gs_attributes, read_gs_ply, write_gs_ply
from shared_utils.sh_utils import RGB2SH, SH2RGB


class PointCloud(NamedTuple):
    """points / colours / normals, all [N,3] arrays (reference :51-54)"""
    points: np.ndarray
    colors: np.ndarray
    normals: np.ndarray


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
        self.covariance_activation = lambda scaling, modifier, rotation: build_covariance_from_scaling_rotation(scaling, modifier, rotation)

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

    def get_covariance(self, scaling_modifier=1, gaussain_idx=None):
        """[N,6] world covariances (reference :397-401; the keyword keeps the reference's spelling)"""
        if gaussain_idx is None:
            return build_covariance_from_scaling_rotation(self.get_scaling, scaling_modifier, self._rotation)
        return build_covariance_from_scaling_rotation(self.get_scaling[gaussain_idx], scaling_modifier, self._rotation[gaussain_idx])

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

    def create_from_pcd(self, pcd, spatial_lr_scale=1.0):
        """points + colours -> model, as the reference (:407-433): SH DC from the colours, isotropic scale = sqrt of the mean squared distance to
        the 3 nearest neighbours (simple_knn.distCUDA2 -> the HIP kernel behind include/c3d_knn.h), identity rotation, opacity 0.1.
        `pcd` has .points [N,3] and .colors [N,3] (the reference's PointCloud NamedTuple or anything shaped like it)."""
        from simple_knn._C import distCUDA2
        self.spatial_lr_scale = spatial_lr_scale
        pts = torch.tensor(np.asarray(pcd.points)).float().to(self.device)
        n, K = pts.shape[0], (self.max_sh_degree + 1) ** 2
        feats = torch.zeros((n, K, 3), dtype=torch.float32, device=self.device)
        feats[:, 0] = RGB2SH(torch.tensor(np.asarray(pcd.colors)).float().to(self.device))
        dist2 = torch.clamp_min(distCUDA2(pts), 0.0000001)
        scales = torch.log(torch.sqrt(dist2))[..., None].repeat(1, 3)
        rots = torch.zeros((n, 4), device=self.device)
        rots[:, 0] = 1
        opac = inverse_sigmoid(0.1 * torch.ones((n, 1), dtype=torch.float32, device=self.device))
        self._set(pts, feats[:, :1], feats[:, 1:], scales, rots, opac)

    def create_from_mesh(self, mesh, num_pts):
        """ceil(num_pts / F) uniformly sampled points per face (v0 (1 - sqrt r1) + v1 (1 - r2) sqrt r1 + v2 r2 sqrt r1), random near-black
        colours, then create_from_pcd (reference :500-524, vectorised).  Deviation: the reference prepends a dummy vertex "because .obj indices
        start at 1" and then indexes it with the loader's 0-based faces, which shifts every face by one vertex; faces are used as they are here."""
        v = mesh.v.detach().cpu().numpy().astype(np.float64)
        f = mesh.f.detach().cpu().numpy().astype(np.int64)
        per = int(math.ceil(num_pts / max(f.shape[0], 1)))
        r1, r2 = np.random.random((f.shape[0], per, 1)), np.random.random((f.shape[0], per, 1))
        s1 = np.sqrt(r1)
        v0, v1, v2 = (v[f[:, k]][:, None, :] for k in range(3))
        xyz = (v0 * (1.0 - s1) + v1 * (1.0 - r2) * s1 + v2 * r2 * s1).reshape(-1, 3)
        shs = np.random.random((xyz.shape[0], 3)) / 255.0
        self.create_from_pcd(PointCloud(points=xyz, colors=SH2RGB(shs), normals=np.zeros_like(xyz)), 10)

    def create_from_uv_data(self, uv_grids_data):
        """one Gaussian per texel of a mesh's UV grid (reference :526-539): uv_grids_data = (uv_coords, uv_coords_3d, uv_normals_3d); only the 3D positions
        are used, colours are the same random near-black as the other initialisers, learning-rate scale 10"""
        if uv_grids_data is None or len(uv_grids_data) != 3:
            raise TypeError("create_from_uv_data(): pass (uv_coords, uv_coords_3d, uv_normals_3d)")
        xyz = np.asarray([np.asarray(p, dtype=np.float64) for p in uv_grids_data[1]], dtype=np.float64).reshape(-1, 3)
        shs = np.random.random((xyz.shape[0], 3)) / 255.0
        self.create_from_pcd(PointCloud(points=xyz, colors=SH2RGB(shs), normals=np.zeros_like(xyz)), 10)

    def create_from_tensors(self, xyz, features, scaling_raw, rotation_raw, opacity_raw, spatial_lr_scale=1.0):
        """raw tensors in the layout create_from_ply produces (:486-498): features [N,K,3]"""
        self.spatial_lr_scale = spatial_lr_scale
        features = torch.as_tensor(features, dtype=torch.float32)
        self._set(xyz, features[:, :1], features[:, 1:], scaling_raw, rotation_raw, opacity_raw)
        self.active_sh_degree = self.max_sh_degree

    def get_points_cloud(self):
        """positions + SH2RGB of the full [N,K,3] feature block, zero normals (reference :468-473)"""
        xyz = self._xyz.detach().cpu().numpy()
        return PointCloud(points=xyz, colors=SH2RGB(self.get_features.detach().cpu().numpy()), normals=np.zeros_like(xyz))

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

    # ---- checkpoint / resume (reference :255-288: the same 12-tuple in the same order) ----
    def capture(self):
        return (self.active_sh_degree, self._xyz, self._features_dc, self._features_rest, self._scaling, self._rotation, self._opacity,
                self.max_radii2D, self.xyz_gradient_accum, self.denom, self.optimizer.state_dict(), self.spatial_lr_scale)

    def restore(self, model_args, training_args):
        (self.active_sh_degree, self._xyz, self._features_dc, self._features_rest, self._scaling, self._rotation, self._opacity,
         self.max_radii2D, xyz_gradient_accum, denom, opt_dict, self.spatial_lr_scale) = model_args
        self.init_xyz = self._xyz.detach().clone() if self.init_xyz.shape != self._xyz.shape else self.init_xyz
        self.training_setup(training_args)
        self.xyz_gradient_accum, self.denom = xyz_gradient_accum, denom
        self.optimizer.load_state_dict(opt_dict)

    def update_learning_rate(self, iteration):
        for group in self.optimizer.param_groups:
            if group["name"] == "xyz":
                group['lr'] = lr = self.xyz_scheduler_args(iteration)
                return lr

    # ---- densify / prune (reference :543-781).  The reference runs clone -> split -> prune as three rounds of boolean-mask indexing and
    # torch.cat over 6 parameters + 12 Adam moments + 4 side arrays.  Here the three rounds are folded into ONE source-index list: every
    # surviving point of the result names the point it is copied from and what it is (kept / clone / split child), and each array is
    # rebuilt with a single gather.  The outcome (set and order of points, parameter values, optimizer state, statistics) is the reference's.
    _NAMES = ("xyz", "f_dc", "f_rest", "opacity", "scaling", "rotation")
    use_device_densify = True        # HIP device: include/c3d_densify.h; False keeps the torch form (tests compare the two)

    def _param_dict(self):
        return dict(zip(self._NAMES, (self._xyz, self._features_dc, self._features_rest, self._opacity, self._scaling, self._rotation)))

    def _install(self, new_tensors, moments_fn):
        """swap every parameter for new_tensors[name]; moments_fn(old_moment) -> new moment (or None to zero-fill)"""
        out = {}
        for group in self.optimizer.param_groups:
            old = group["params"][0]
            newp = nn.Parameter(new_tensors[group["name"]].contiguous().requires_grad_(True))
            st = self.optimizer.state.pop(old, None)
            if st is not None and len(st):
                for k in ("exp_avg", "exp_avg_sq"):
                    m = moments_fn(st[k])
                    st[k] = torch.zeros_like(newp) if m is None else m.contiguous()
                self.optimizer.state[newp] = st
            group["params"][0] = newp
            out[group["name"]] = newp
        self._xyz, self._features_dc, self._features_rest = out["xyz"], out["f_dc"], out["f_rest"]
        self._opacity, self._scaling, self._rotation = out["opacity"], out["scaling"], out["rotation"]

    def add_densification_stats(self, viewspace_point_grad, update_filter, radii=None):
        """viewspace_point_grad [N,>=2]: gradient of the loss w.r.t. the screen-space means of ONE view (reference :767-769)"""
        norm = torch.norm(viewspace_point_grad[:, :2], dim=-1, keepdim=True)
        f = update_filter.to(norm.dtype).unsqueeze(-1)
        self.xyz_gradient_accum += norm * f
        self.denom += f
        if radii is not None:
            self.max_radii2D = torch.where(update_filter, torch.maximum(self.max_radii2D, radii.to(self.max_radii2D.dtype)), self.max_radii2D)

    def reset_opacity(self):
        new = inverse_sigmoid(torch.clamp_max(self.get_opacity.detach(), 0.01))
        tensors = {k: v.detach() for k, v in self._param_dict().items()}
        keep_names = {"opacity"}
        tensors["opacity"] = new
        # only the opacity group loses its moments (reference replace_tensor_to_optimizer :543-556)
        for group in self.optimizer.param_groups:
            if group["name"] not in keep_names:
                continue
            old = group["params"][0]
            newp = nn.Parameter(new.contiguous().requires_grad_(True))
            st = self.optimizer.state.pop(old, None)
            if st is not None and len(st):
                st["exp_avg"], st["exp_avg_sq"] = torch.zeros_like(newp), torch.zeros_like(newp)
                self.optimizer.state[newp] = st
            group["params"][0] = newp
            self._opacity = newp

    def prune_points(self, mask):
        """drop the points where mask is True (reference :575-590)"""
        src = torch.nonzero(~mask, as_tuple=False).squeeze(-1)
        self._gather(src, None)
        self.xyz_gradient_accum, self.denom = self.xyz_gradient_accum[src], self.denom[src]
        self.max_radii2D = self.max_radii2D[src]

    def _gather(self, src, fresh):
        """rebuild every per-point array as old[src]; fresh [len(src)] bool marks points that start with zero Adam moments"""
        tensors = {k: v.detach().index_select(0, src) for k, v in self._param_dict().items()}
        if fresh is None:
            mom = lambda m: m.index_select(0, src)
        else:
            def mom(m):
                g = m.index_select(0, src)
                return g * (~fresh).to(g.dtype).view(-1, *([1] * (g.ndim - 1)))
        self.init_xyz = self.init_xyz.index_select(0, src)
        self._install(tensors, mom)
        return tensors

    def _densify_and_prune_device(self, max_grad, min_opacity, extent, max_screen_size, generator, noise):
        """The same result on a HIP device through include/c3d_densify.h: classification and the four prefix sums on the device, ONE host
        read (six counts: the new point count has to be known to allocate), the source-index list written by one kernel, and ONE gather launch for the six
        parameters, their twelve Adam moments and init_xyz -- where the reference needs a device -> host synchronisation per boolean mask and a
        torch.cat per array.  Children are positioned with a few torch ops on the gathered rows (no synchronisation)."""
        import ctypes as C
        import c3d_hip as _h
        lib, dev = _h.lib(), self._xyz.device
        N = self._xyz.shape[0]
        u8 = dict(dtype=torch.uint8, device=dev)
        with torch.no_grad(), torch.cuda.device(dev):
            st = _h.stream(dev)
            plan = torch.empty((lib.c3d_densify_plan_bytes(N),), **u8)
            counts = torch.empty((8,), dtype=torch.int32, device=dev)
            old = {k: v.detach().contiguous() for k, v in self._param_dict().items()}
            _h.check(lib.c3d_densify_plan(N, _h.ptr(self.xyz_gradient_accum.contiguous()), _h.ptr(self.denom.contiguous()), _h.ptr(old["scaling"]), _h.ptr(old["opacity"]),
                                          float(max_grad), float(self.percent_dense * extent), float(min_opacity), float(0.1 * extent if max_screen_size else 0.0),
                                          _h.ptr(plan), _h.ptr(counts), st), "c3d_densify_plan")
            nK, nC, nS, nS_all, nC_all = counts.tolist()[:5]                  # the one host synchronisation of the step
            M = nK + nC + 2 * nS
            src = torch.empty((max(M, 1),), dtype=torch.int32, device=dev)
            fresh = torch.empty((max(M, 1),), **u8)
            crank = torch.empty((max(2 * nS, 1),), dtype=torch.int32, device=dev)
            _h.check(lib.c3d_densify_fill(N, _h.ptr(plan), (C.c_uint32 * 8)(nK, nC, nS, nS_all, nC_all, 0, 0, 0), _h.ptr(src), _h.ptr(fresh), _h.ptr(crank), st), "c3d_densify_fill")
            # every per-point array through ONE gather launch
            names = list(self._NAMES)
            srcs, dsts, rows, zero = [], [], [], []
            new = {k: torch.empty((M,) + tuple(old[k].shape[1:]), dtype=torch.float32, device=dev) for k in names}
            moms = {}
            for group in self.optimizer.param_groups:
                stt = self.optimizer.state.get(group["params"][0])
                if stt is not None and len(stt):
                    moms[group["name"]] = {k: (stt[k].contiguous(), torch.empty_like(new[group["name"]])) for k in ("exp_avg", "exp_avg_sq")}
            new_init = torch.empty((M, 3), dtype=torch.float32, device=dev)
            for k in names:
                srcs.append(old[k]); dsts.append(new[k]); rows.append(old[k][0].numel() if N else 1); zero.append(0)
                for kk in moms.get(k, {}).values():
                    srcs.append(kk[0]); dsts.append(kk[1]); rows.append(old[k][0].numel() if N else 1); zero.append(1)     # clones / children start with zero moments
            srcs.append(self.init_xyz.contiguous()); dsts.append(new_init); rows.append(3); zero.append(0)
            if M:
                na = len(srcs)
                _h.check(lib.c3d_gather_rows(na, (C.c_void_p * na)(*[t_.data_ptr() for t_ in srcs]), (C.c_void_p * na)(*[t_.data_ptr() for t_ in dsts]), (C.c_int32 * na)(*rows),
                                             (C.c_int32 * na)(*zero), _h.ptr(src), _h.ptr(fresh), M, st), "c3d_gather_rows")
            if nS:
                # split children (reference :641-670, N = 2): position sampled from the parent's Gaussian, scale / (0.8 * 2).  The noise tensor has the
                # reference's shape (2 x ALL split parents, 3) and row (child, parent rank), so that an equally seeded generator gives the same samples
                if noise is None:
                    noise = torch.randn((2 * nS_all, 3), device=dev, generator=generator)
                lo = nK + nC
                par = src[lo:lo + 2 * nS].long()
                row = (torch.arange(2 * nS, device=dev) >= nS).long() * nS_all + crank[:2 * nS].long()
                std = torch.exp(old["scaling"][par])
                rot = build_rotation(old["rotation"][par])
                new["xyz"][lo:lo + 2 * nS] = torch.bmm(rot, (noise[row] * std).unsqueeze(-1)).squeeze(-1) + old["xyz"][par]
                new["scaling"][lo:lo + 2 * nS] = self.scaling_inverse_activation(std / 1.6)
            self.init_xyz = new_init
            mom_new = {k: {kk: vv[1] for kk, vv in m.items()} for k, m in moms.items()}
            for group in self.optimizer.param_groups:          # install: parameters and moments are already gathered
                oldp = group["params"][0]
                newp = nn.Parameter(new[group["name"]].requires_grad_(True))
                stt = self.optimizer.state.pop(oldp, None)
                if stt is not None and len(stt):
                    stt["exp_avg"], stt["exp_avg_sq"] = mom_new[group["name"]]["exp_avg"], mom_new[group["name"]]["exp_avg_sq"]
                    self.optimizer.state[newp] = stt
                group["params"][0] = newp
                new[group["name"]] = newp
            self._xyz, self._features_dc, self._features_rest = new["xyz"], new["f_dc"], new["f_rest"]
            self._opacity, self._scaling, self._rotation = new["opacity"], new["scaling"], new["rotation"]
            self.xyz_gradient_accum = torch.zeros((M, 1), device=dev)
            self.denom = torch.zeros((M, 1), device=dev)
            self.max_radii2D = torch.zeros((M,), device=dev)
        n_keep_all = N - nS_all
        return {"cloned": int(nC_all), "split": int(nS_all), "pruned": int((n_keep_all - nK) + (nC_all - nC) + 2 * (nS_all - nS)), "points": int(M)}

    def densify_and_prune(self, max_grad, min_opacity, extent, max_screen_size, generator=None, noise=None):
        """clone small / split large high-gradient points, then prune (reference densify_and_prune :748-750 = clone :672-690, split
        :641-670 with N=2, prune :771-781), in one pass.  Statistics are zeroed as densification_postfix does (:634-637) -- which is
        also why the reference's screen-size criterion never fires: max_radii2D is all zero by the time prune() looks at it.
        On a HIP device the index construction and the gather run in the library (_densify_and_prune_device); this torch form is the CPU path of the
        host-logic tests and the statement both are held to.  `noise` (tests): the [2 * split, 3] normal samples instead of a draw from `generator`."""
        if self._xyz.is_cuda and self._xyz.shape[0] > 0 and self.use_device_densify:
            return self._densify_and_prune_device(max_grad, min_opacity, extent, max_screen_size, generator, noise)
        N = self._xyz.shape[0]
        dev = self._xyz.device
        with torch.no_grad():
            g = self.xyz_gradient_accum / self.denom
            g[g.isnan()] = 0.0
            g = g.norm(dim=-1)                                              # [N]
            scal = self.get_scaling.detach()
            smax = scal.max(dim=1).values
            opac = self.get_opacity.detach().squeeze(-1)
            hot = g >= max_grad
            small = smax <= self.percent_dense * extent
            clone, split = hot & small, hot & ~small
            ar = torch.arange(N, device=dev)
            i_keep, i_clone, i_split = ar[~split], ar[clone], ar[split]
            nK, nC, nS = i_keep.numel(), i_clone.numel(), i_split.numel()
            # candidate list in the reference's final order: survivors, clones, first children, second children
            src = torch.cat((i_keep, i_clone, i_split, i_split))
            kind = torch.cat((torch.zeros(nK, dtype=torch.int8, device=dev), torch.ones(nC, dtype=torch.int8, device=dev),
                              torch.full((2 * nS,), 2, dtype=torch.int8, device=dev)))
            child = kind == 2
            # prune criteria evaluated on what each candidate will be (children are 1.6x smaller)
            c_smax = torch.where(child, smax[src] / 1.6, smax[src])
            dead = opac[src] < min_opacity
            if max_screen_size:
                dead = dead | (c_smax > 0.1 * extent)                       # (max_radii2D > max_screen_size) is identically False, see above
            # split children: position sampled from the parent's Gaussian, scale / (0.8 * 2)
            if nS:
                std = scal[i_split].repeat(2, 1)
                noise = (torch.randn(std.shape, device=dev, generator=generator) if noise is None else noise) * std
                rot = build_rotation(self._rotation.detach()[i_split]).repeat(2, 1, 1)
                child_xyz = torch.bmm(rot, noise.unsqueeze(-1)).squeeze(-1) + self._xyz.detach()[i_split].repeat(2, 1)
                child_scaling = self.scaling_inverse_activation(std / 1.6)
            alive = ~dead
            sel = torch.nonzero(alive, as_tuple=False).squeeze(-1)
            final_src, fresh = src[sel], (kind != 0)[sel]
            tensors = self._gather(final_src, fresh)
            if nS:
                pos = torch.cumsum(alive.to(torch.int64), 0) - 1                # candidate index -> index in the result
                cidx = torch.arange(nK + nC, nK + nC + 2 * nS, device=dev)
                ok = alive[cidx]
                with torch.no_grad():
                    self._xyz.data[pos[cidx][ok]] = child_xyz[ok]
                    self._scaling.data[pos[cidx][ok]] = child_scaling[ok]
            n = self._xyz.shape[0]
            self.xyz_gradient_accum = torch.zeros((n, 1), device=dev)
            self.denom = torch.zeros((n, 1), device=dev)
            self.max_radii2D = torch.zeros((n,), device=dev)
        return {"cloned": int(nC), "split": int(nS), "pruned": int((~alive).sum().item()), "points": int(n)}


def build_rotation(r):
    """unit-normalised quaternions (w, x, y, z) [M,4] -> rotation matrices [M,3,3] (reference general_utils build_rotation)"""
    q = r / r.norm(dim=1, keepdim=True)
    w, x, y, z = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
    R = torch.stack((1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y),
                     2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x),
                     2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)), dim=1)
    return R.view(-1, 3, 3)


def build_scaling_rotation(s, r):
    """L = R(q) diag(s) [M,3,3] (reference :104-113)"""
    return build_rotation(r) * s[:, None, :]


def strip_symmetric(sym):
    """upper triangle of symmetric [M,3,3] as [M,6] = (xx, xy, xz, yy, yz, zz): the cov3D_precomp layout of the rasterizer (reference :46-58)"""
    return torch.stack((sym[:, 0, 0], sym[:, 0, 1], sym[:, 0, 2], sym[:, 1, 1], sym[:, 1, 2], sym[:, 2, 2]), dim=1)


strip_lowerdiag = strip_symmetric


def build_covariance_from_scaling_rotation(scaling, scaling_modifier, rotation):
    """the reference's covariance_activation (:219-224): Sigma = L L^T with L = R(q) diag(modifier * s), as a 6-vector"""
    L = build_scaling_rotation(scaling_modifier * scaling, rotation)
    return strip_symmetric(L @ L.transpose(1, 2))


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
            if self.device.type == "cuda":
                # the reference's path: PointCloud -> create_from_pcd with the 3-nearest-neighbour scale (:823-828)
                self.gaussians.create_from_pcd(PointCloud(points=xyz, colors=SH2RGB(shs), normals=np.zeros((num_pts, 3))), 10)
            else:   # host-logic tests: mean nearest-neighbour spacing of a uniform ball as the isotropic scale
                spacing = radius * (4.0 / 3.0 * np.pi / max(num_pts, 1)) ** (1.0 / 3.0)
                self.gaussians.create_from_arrays(xyz, SH2RGB(shs), np.full((num_pts,), spacing), spatial_lr_scale=10)
        elif isinstance(input, PlyData):
            self.gaussians.create_from_ply(input)
        elif isinstance(input, PointCloud):
            self.gaussians.create_from_pcd(input, 1)
        elif isinstance(input, dict):
            self.gaussians.create_from_tensors(**input)
        elif hasattr(input, "v") and hasattr(input, "f"):        # mesh_processer.mesh.Mesh
            self.gaussians.create_from_mesh(input, num_pts)
        else:      # anything else is the UV-grid triple, as in the reference (:827-828)
            self.gaussians.create_from_uv_data(input)

    def raw_storage_ok(self):
        """can the raw-parameter kernels (render()'s fused path, render_views, the fused training step) read this model's storage?  SH degree 0-3, f_dc [N,1,3] and
        f_rest [N,K-1,3] with K in (1, 4, 9, 16) coefficients per channel, enough of them for the active degree (diff_gaussian_rasterization.raw_sh_coeffs)"""
        g = self.gaussians
        fd, fr = g._features_dc, g._features_rest
        if not (0 <= g.max_sh_degree <= 3 and fd.dim() == 3 and tuple(fd.shape[1:]) == (1, 3) and fr.dim() == 3 and fr.shape[2] == 3):
            return False
        K = int(fr.shape[1]) + 1
        return K in (1, 4, 9, 16) and (g.active_sh_degree + 1) ** 2 <= K

    def render_views(self, viewpoint_cameras, bg_colors=None, scaling_modifier=1.0, lanes=1, group=16, streams=4):
        """Inference-only batch of render(): all cameras in ONE library call (c3d_gs_render_views_raw; `group` views per launch of every stage, no host
        synchronisation between views).  Same per-view results as render() -- the orbit-renderer loop of the reference's
        nodes without its per-view launch gaps.  bg_colors: None, one [3] tensor or one per camera.  streams > 1 (default 4; at least 8 views per part): the views in that many parts on HIP streams of their own
        (c3d_hip.gs_step.FusedViewRender: one part's binning underneath another part's compositing; same bits, `streams` workspaces).
        -> dict(image [V,3,H,W] clamped, depth [V,1,H,W], alpha [V,1,H,W], radii [V,N], visibility_filter [V,N])"""
        from diff_gaussian_rasterization import GaussianRasterizationSettings
        from c3d_hip.gs_step import FusedViewRender
        g = self.gaussians
        if not g._xyz.is_cuda:
            raise RuntimeError("render_views needs a HIP-resident model; there is no CPU path")
        if not self.raw_storage_ok():      # the per-view loop over render() (which takes the accessor path for such a model)
            outs = [self.render(c, scaling_modifier, bg_color=(bg_colors[i] if isinstance(bg_colors, (list, tuple)) else bg_colors)) for i, c in enumerate(viewpoint_cameras)]
            with torch.no_grad():
                return {"image": torch.stack([o["image"] for o in outs]), "depth": torch.stack([o["depth"] for o in outs]), "alpha": torch.stack([o["alpha"] for o in outs]),
                        "radii": torch.stack([o["radii"] for o in outs]), "visibility_filter": torch.stack([o["visibility_filter"] for o in outs])}
        V = len(viewpoint_cameras)
        if bg_colors is None or torch.is_tensor(bg_colors):
            bg_colors = [self.bg_color if bg_colors is None else bg_colors] * V
        H, W = int(viewpoint_cameras[0].image_height), int(viewpoint_cameras[0].image_width)
        settings = [GaussianRasterizationSettings(H, W, math.tan(c.FoVx * 0.5), math.tan(c.FoVy * 0.5), bg, scaling_modifier, c.world_view_transform,
                                                  c.full_proj_transform, g.active_sh_degree, c.camera_center, False, False)
                    for c, bg in zip(viewpoint_cameras, bg_colors)]
        key = (g._xyz.shape[0], H, W, int(lanes), int(group), int(streams))
        if getattr(self, "_view_render_key", None) != key:
            self._view_render, self._view_render_key = FusedViewRender(key[0], H, W, self.device, lanes=lanes, group=group, streams=streams), key
        with torch.no_grad():
            color, depth, alpha, radii = self._view_render.run(settings, [g._xyz, g._features_dc, g._features_rest, g._opacity, g._scaling, g._rotation], want_radii=True)
            return {"image": color.clamp_(0, 1), "depth": depth, "alpha": alpha, "radii": radii, "visibility_filter": radii > 0}

    def render(self, viewpoint_camera, scaling_modifier=1.0, gaussain_idx=None, bg_color=None, override_color=None,
               compute_cov3D_python=False, convert_SHs_python=False):
        """-> dict(image[3,H,W] clamped, depth[1,H,W], alpha[1,H,W], viewspace_points[N,3], visibility_filter[N], radii[N])"""
        from diff_gaussian_rasterization import GaussianRasterizationSettings, GaussianRasterizer
        if convert_SHs_python:
            raise NotImplementedError("convert_SHs_python is dead code in the reference (:902 reads an attribute that does not exist)")
        g = self.gaussians
        settings = GaussianRasterizationSettings(
            image_height=int(viewpoint_camera.image_height), image_width=int(viewpoint_camera.image_width),
            tanfovx=math.tan(viewpoint_camera.FoVx * 0.5), tanfovy=math.tan(viewpoint_camera.FoVy * 0.5),
            bg=self.bg_color if bg_color is None else bg_color, scale_modifier=scaling_modifier,
            viewmatrix=viewpoint_camera.world_view_transform, projmatrix=viewpoint_camera.full_proj_transform,
            sh_degree=g.active_sh_degree, campos=viewpoint_camera.camera_center, prefiltered=False, debug=False)
        fused = (gaussain_idx is None and override_color is None and not compute_cov3D_python and self.raw_storage_ok() and g._xyz.is_cuda
                 and not self.force_unfused)
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
            cov = None
            if compute_cov3D_python:         # world covariances from the model (already carrying the modifier) instead of scales + rotations
                cov, scales, rots = g.get_covariance(scaling_modifier, gaussain_idx), None, None
            image, radii, depth, alpha = GaussianRasterizer(raster_settings=settings)(
                means3D=xyz, means2D=screenspace_points, shs=shs, colors_precomp=colors, opacities=opac,
                scales=scales, rotations=rots, cov3D_precomp=cov)
        return {"image": image.clamp(0, 1), "depth": depth, "alpha": alpha, "viewspace_points": screenspace_points,
                "visibility_filter": radii > 0, "radii": radii}
