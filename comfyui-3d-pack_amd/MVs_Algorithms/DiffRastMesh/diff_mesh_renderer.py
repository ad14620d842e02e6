"""DiffRastRenderer: differentiable textured-mesh renderer over the `nvdiffrast.torch` boundary.

Host-side mirror of /root/reference/MVs_Algorithms/DiffRastMesh/diff_mesh_renderer.py: constructor :38-54,
get_params :56-66, update_mesh :68-70, render :72-159 -- the same op sequence with the same arguments
(MVP transform -> rasterize -> antialias(alpha) -> interpolate(uv, diff all) -> texture(linear) -> sigmoid ->
interpolate(depth / normal) -> antialias(albedo) -> composite -> optional SSAA resize), the same result dict.
Written from scratch; the ops resolve to the MI355X HIP kernels (nvdiffrast/torch/__init__.py)."""
import math
import os

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F
import nvdiffrast.torch as dr

from mesh_processer.mesh import safe_normalize
from c3d_hip.mesh_fused import render_view, shade, transform_vertices, view_state_tensors


def inverse_sigmoid(x):
    return torch.log(x / (1 - x))


def scale_img_nhwc(x, size, mag='bilinear', min='bilinear'):
    """resize an NHWC batch.  Shrinking in both directions uses `min`; everything else `mag`, with align_corners for the two smooth
    modes (what the reference does, :14-26); magnifying one side while minifying the other is refused."""
    src_h, src_w = int(x.shape[1]), int(x.shape[2])
    dst_h, dst_w = int(size[0]), int(size[1])
    grows = src_h < dst_h and src_w < dst_w
    if not (grows or (src_h >= dst_h and src_w >= dst_w)):
        raise AssertionError("Trying to magnify image in one dimension and minify in the other")
    if src_h > dst_h and src_w > dst_w:
        mode, extra = min, {}
    else:
        mode, extra = mag, ({"align_corners": True} if mag in ("bilinear", "bicubic") else {})
    return F.interpolate(x.permute(0, 3, 1, 2), (dst_h, dst_w), mode=mode, **extra).permute(0, 2, 3, 1).contiguous()


def scale_img_hwc(x, size, mag='bilinear', min='bilinear'):
    return scale_img_nhwc(x.unsqueeze(0), size, mag, min).squeeze(0)


def scale_img_nhw(x, size, mag='bilinear', min='bilinear'):
    return scale_img_nhwc(x.unsqueeze(-1), size, mag, min).squeeze(-1)


def scale_img_hw(x, size, mag='bilinear', min='bilinear'):
    return scale_img_nhwc(x[None, ..., None], size, mag, min)[0, ..., 0]


def make_divisible(x, m=8):
    """the next multiple of m at or above x"""
    return -int(-x // m) * m if float(x).is_integer() else int(math.ceil(x / m)) * m


def vertex_normals_from_faces(v, f):
    """unit face normals summed onto their three vertices (not re-normalised here: the caller normalises after interpolation); vertices
    that collect nothing get +z -- the reference's rebuild when the geometry trains (:119-131)"""
    corner = [f[:, k].long() for k in range(3)]
    face_n = safe_normalize(torch.cross(v[corner[1]] - v[corner[0]], v[corner[2]] - v[corner[0]], dim=-1))
    acc = torch.zeros_like(v)
    for idx in corner:
        acc.index_add_(0, idx, face_n)
    empty = (acc * acc).sum(-1, keepdim=True) <= 1e-20
    return torch.where(empty, acc.new_tensor([0.0, 0.0, 1.0]), acc)


class LazyResults(dict):
    """dict whose deferred entries are computed on first access; iteration / `in` / len see every key"""

    def __init__(self):
        super().__init__()
        self._thunks = {}

    def defer(self, key, fn):
        self._thunks[key] = fn
        super().__setitem__(key, None)

    def __getitem__(self, key):
        fn = self._thunks.pop(key, None)
        if fn is not None:
            super().__setitem__(key, fn())
        return super().__getitem__(key)

    def get(self, key, default=None):
        return self[key] if key in self else default

    def items(self):
        return [(k, self[k]) for k in list(self.keys())]

    def values(self):
        return [self[k] for k in list(self.keys())]


class DiffRastRenderer(nn.Module):
    def __init__(self, mesh, force_cuda_rast):
        super().__init__()
        self.mesh = mesh
        self.glctx = dr.RasterizeCudaContext() if (force_cuda_rast or os.name != 'nt') else dr.RasterizeGLContext()
        self.v_offsets = nn.Parameter(torch.zeros_like(self.mesh.v), requires_grad=True)
        self.raw_albedo = nn.Parameter(inverse_sigmoid(self.mesh.albedo), requires_grad=True)
        self.train_geo = False
        self.fused_glue = True          # False: the reference's torch op chain around the nvdiffrast calls (tests compare both)
        self.fused_view = True          # the whole view as one library call each way (c3d_mesh_view_*); False: op by op over nvdiffrast.torch

    def get_params(self, texture_lr, train_geo, geom_lr):
        params = [{'params': self.raw_albedo, 'lr': texture_lr}]
        self.train_geo = train_geo
        if train_geo:
            params.append({'params': self.v_offsets, 'lr': geom_lr})
        return params

    def update_mesh(self):
        self.mesh.v = (self.mesh.v + self.v_offsets).detach()
        self.mesh.albedo = torch.sigmoid(self.raw_albedo.detach())

    @staticmethod
    def _bg_host(bg_color):
        """background as three host floats without a device round trip per view: a tensor's values are read once and kept on the tensor object
        (the camera controller hands the same white / black tensors over every step)"""
        if not torch.is_tensor(bg_color):
            return (float(bg_color),) * 3
        try:
            ver = bg_color._version
        except RuntimeError:                 # inference tensors (the orbit-renderer nodes run under torch.inference_mode) keep no version counter
            ver = None
        host = getattr(bg_color, "_c3d_host", None) if ver is not None else None
        if host is None or host[0] != ver:
            vals = [float(x) for x in bg_color.detach().reshape(-1).cpu().tolist()]
            host = (ver, tuple((vals * 3)[:3]))
            try:
                bg_color._c3d_host = host
            except AttributeError:
                pass
        return host[1]

    def _render_fused_view(self, v, pose_np, pose_inv_np, proj_np, h, w, bg_color, optional_render_types):
        """ssaa = 1, linear filter, HIP device: one library call forward (and one backward) for the whole op sequence of `render`"""
        import nvdiffrast.torch as _dr
        mesh = self.mesh
        f, ft = mesh.f.to(torch.int32).contiguous(), mesh.ft.to(torch.int32).contiguous()
        vt = mesh.vt.to(torch.float32).contiguous()
        clip = (proj_np @ pose_inv_np).astype(np.float32)
        image, alpha, hold = render_view(mesh.v, self.v_offsets if self.train_geo else None, self.raw_albedo, f, vt, ft, clip, self._bg_host(bg_color), h, w, self.glctx,
                                         _dr._topology(f))
        results = LazyResults()
        results['image'], results['alpha'] = image, alpha
        lazy_v = {}

        def cur_v():
            if 'v' not in lazy_v:
                lazy_v['v'] = v if v is not None else (mesh.v + self.v_offsets if self.train_geo else mesh.v)
            return lazy_v['v']

        def raster():
            # depth / normal are produced on demand; with autograd on and trainable geometry they get their own (differentiable) rasterization,
            # otherwise the one the fused call left in its state
            if torch.is_grad_enabled() and self.train_geo:
                v_clip = transform_vertices(cur_v(), torch.from_numpy(clip).to(mesh.v.device)).unsqueeze(0)
                return dr.rasterize(self.glctx, v_clip, mesh.f, (h, w))[0]
            return view_state_tensors(hold, h, w)[0]

        def depth_fn():
            vc = transform_vertices(cur_v(), torch.from_numpy(pose_inv_np).to(mesh.v.device)).unsqueeze(0)
            d, _ = dr.interpolate(-vc[..., [2]], raster(), mesh.f)
            return d.squeeze(0)

        shading = {}

        def normal_pair():
            if not shading:
                vn = vertex_normals_from_faces(cur_v(), mesh.f) if self.train_geo else mesh.vn
                normal, _ = dr.interpolate(vn.unsqueeze(0).contiguous(), raster(), mesh.fn)
                normal = safe_normalize(normal[0])
                viewcos = normal @ torch.from_numpy(pose_np[:3, :3].copy()).to(mesh.v.device)
                shading['normal'], shading['viewcos'] = (normal + 1) / 2, (viewcos + 1) / 2
            return shading
        if 'depth' in optional_render_types:
            results.defer('depth', depth_fn)
        if 'normal' in optional_render_types:
            results.defer('normal', lambda: normal_pair()['normal'])
            results.defer('viewcos', lambda: normal_pair()['viewcos'])
        return results

    def render(self, pose, proj, h0, w0, ssaa=1, bg_color=1, texture_filter='linear', optional_render_types=['depth', 'normal']):
        h, w = (make_divisible(h0 * ssaa, 8), make_divisible(w0 * ssaa, 8)) if ssaa != 1 else (h0, w0)
        mesh = self.mesh
        # the 4x4 inverse is taken on the host (the reference calls torch.inverse on the device: a solver launch + sync per view);
        # one upload carries pose, its inverse and the projection
        pose_np = pose.astype(np.float32)
        pose_inv_np = np.linalg.inv(pose_np).astype(np.float32)
        proj_np = proj.astype(np.float32)
        fused = mesh.v.is_cuda and ssaa == 1 and self.fused_glue
        if mesh.v.is_cuda and self.fused_glue and self.fused_view and texture_filter == 'linear':
            # the fused view adds the offsets inside its transform kernel; `v` (one more launch and autograd node per view) only exists if depth / normal are read
            res = self._render_fused_view(None, pose_np, pose_inv_np, proj_np, h, w, bg_color, optional_render_types)
            if ssaa == 1:
                return res
            # super-sampling (reference :142-149): the view runs at the super-sampled size -- still ONE library call each way -- and image / alpha (and,
            # on demand, depth / normal / viewcos) are brought down with the reference's bilinear scale_img_hwc.  The reference scales the composite and
            # clamps afterwards; composite and scaled composite of values in [0, 1] stay in [0, 1], the fused view's clamp before the scaling is the same image.
            out = LazyResults()
            out['image'] = scale_img_hwc(res['image'], (h0, w0)).clamp(0, 1)
            out['alpha'] = scale_img_hwc(res['alpha'], (h0, w0))
            for k in ('depth', 'normal', 'viewcos'):
                if k in res:
                    out.defer(k, (lambda kk: (lambda: scale_img_hwc(res[kk], (h0, w0))))(k))
            return out
        v = mesh.v + self.v_offsets if self.train_geo else mesh.v
        mats = torch.from_numpy(np.stack((pose_np, pose_inv_np, proj_np, proj_np @ pose_inv_np))).to(v.device)
        pose, pose_inv, proj, clip_from_world = mats[0], mats[1], mats[2], mats[3]
        if fused:
            # pad + two GEMMs of the reference as one kernel each way (c3d_mesh_transform_*); camera-space positions only if depth is asked for
            v_clip = transform_vertices(v, clip_from_world).unsqueeze(0)
            v_cam = None
        else:
            v_cam = torch.matmul(F.pad(v, pad=(0, 1), mode='constant', value=1.0), pose_inv.T).float().unsqueeze(0)
            v_clip = v_cam @ proj.T

        rast, rast_db = dr.rasterize(self.glctx, v_clip, mesh.f, (h, w))
        alpha = torch.clamp(rast[..., -1:], 0, 1).contiguous()                                   # [1,H,W,1]
        alpha = dr.antialias(alpha, rast, v_clip, mesh.f)                                        # silhouette gradients enter here
        if not fused:
            alpha = alpha.clamp(0, 1).squeeze(0)
        texc, texc_db = dr.interpolate(mesh.vt.unsqueeze(0).contiguous(), rast, mesh.ft, rast_db=rast_db, diff_attrs='all')
        albedo = torch.sigmoid(dr.texture(self.raw_albedo.unsqueeze(0), texc, uv_da=texc_db, filter_mode=texture_filter))

        # depth / normal / viewcos are produced on first access (LazyResults): the reference's trainer asks for them by default
        # (optional_render_types) and then only reads 'image' (diff_mesh.py:104-106) -- two interpolations, the vertex-normal rebuild and
        # their backward passes per view for nothing.  Accessing a key gives exactly the tensor the eager code would have returned.
        def depth_fn():
            vc = v_cam if v_cam is not None else transform_vertices(v, pose_inv).unsqueeze(0)
            d, _ = dr.interpolate(-vc[..., [2]], rast, mesh.f)
            d = d.squeeze(0)
            return scale_img_hwc(d, (h0, w0)) if ssaa != 1 else d

        shading = {}

        def normal_pair():
            if not shading:
                vn = vertex_normals_from_faces(v, mesh.f) if self.train_geo else mesh.vn
                normal, _ = dr.interpolate(vn.unsqueeze(0).contiguous(), rast, mesh.fn)
                normal = safe_normalize(normal[0])
                viewcos = normal @ pose[:3, :3]                                                   # [0,0,1] faces the camera
                if ssaa != 1:
                    normal, viewcos = scale_img_hwc(normal, (h0, w0)), scale_img_hwc(viewcos, (h0, w0))
                shading['normal'], shading['viewcos'] = (normal + 1) / 2, (viewcos + 1) / 2
            return shading

        albedo = dr.antialias(albedo, rast, v_clip, mesh.f).squeeze(0).contiguous()
        results = LazyResults()
        if fused:
            # clamp(alpha), composite over the background and the final clamp in one kernel each way (c3d_mesh_shade_*)
            bg = bg_color if torch.is_tensor(bg_color) else torch.full((3,), float(bg_color), dtype=torch.float32, device=albedo.device)
            image, alpha = shade(albedo, alpha.squeeze(0), bg.to(albedo.device, torch.float32).reshape(-1)[:3].expand(3).contiguous())
            results['image'] = image
        else:
            albedo = alpha * albedo + (1 - alpha) * bg_color
            if ssaa != 1:
                albedo, alpha = scale_img_hwc(albedo, (h0, w0)), scale_img_hwc(alpha, (h0, w0))
            results['image'] = albedo.clamp(0, 1)
        results['alpha'] = alpha
        if 'depth' in optional_render_types:
            results.defer('depth', depth_fn)
        if 'normal' in optional_render_types:
            results.defer('normal', lambda: normal_pair()['normal'])
            results.defer('viewcos', lambda: normal_pair()['viewcos'])
        return results
