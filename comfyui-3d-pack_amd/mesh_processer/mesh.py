"""Minimal torch `Mesh` container: the data contract DiffRastRenderer reads.

Mirrors the attribute names and tensor layouts of /root/reference/mesh_processer/mesh.py:15-66 (v [V,3] f32, f [T,3] i32,
vn/fn, vt/ft, albedo [Ht,Wt,3] in [0,1]) plus `auto_normal` (:471-494), `aabb` / `auto_size` (:450-469) and `set_new_albedo` (:442-447).  Asset I/O
(obj/glb/ply, xatlas UV unwrapping) is out of scope for the hot path."""
import torch


def safe_normalize(x, eps=1e-20):
    return x / torch.sqrt(torch.clamp(torch.sum(x * x, -1, keepdim=True), min=eps))


class Mesh:
    def __init__(self, v=None, f=None, vn=None, fn=None, vt=None, ft=None, albedo=None, vc=None, device=None):
        self.device = torch.device(device) if device is not None else (v.device if v is not None else torch.device("cpu"))
        self.v, self.vn, self.vt, self.f, self.fn, self.ft = v, vn, vt, f, fn, ft
        self.albedo, self.vc = albedo, vc
        self.ori_center, self.ori_scale = 0, 1

    def to(self, device):
        self.device = torch.device(device)
        for name in ("v", "f", "vn", "fn", "vt", "ft", "albedo", "vc"):
            t = getattr(self, name)
            if t is not None:
                setattr(self, name, t.to(device))
        return self

    def auto_normal(self):
        """area-weighted vertex normals; fn = f"""
        i0, i1, i2 = (self.f[:, k].long() for k in range(3))
        v0, v1, v2 = self.v[i0], self.v[i1], self.v[i2]
        face_n = torch.cross(v1 - v0, v2 - v0, dim=-1)
        vn = torch.zeros_like(self.v)
        for idx in (i0, i1, i2):
            vn.scatter_add_(0, idx[:, None].repeat(1, 3), face_n)
        vn = torch.where(torch.sum(vn * vn, -1, keepdim=True) > 1e-20, vn, torch.tensor([0.0, 0.0, 1.0], dtype=torch.float32, device=vn.device))
        self.vn = safe_normalize(vn)
        self.fn = self.f

    def aabb(self):
        """(min xyz, max xyz) of the vertices (reference :450-457)"""
        return torch.min(self.v, dim=0).values, torch.max(self.v, dim=0).values

    @torch.no_grad()
    def auto_size(self, bound=0.9):
        """centre the box and scale its longest side to [-bound, bound] (reference :460-469); remembers ori_center / ori_scale"""
        vmin, vmax = self.aabb()
        self.ori_center = (vmax + vmin) / 2
        self.ori_scale = 2 * bound / torch.max(vmax - vmin).item()
        self.v = (self.v - self.ori_center) * self.ori_scale

    def set_new_albedo(self, res_H, res_W):
        """no texture yet: mid-grey [res_H, res_W, 3]; otherwise the existing one resized bilinearly (reference :442-447)"""
        if self.albedo is None:
            self.albedo = torch.full((res_H, res_W, 3), 0.5, dtype=torch.float32, device=self.device)
        else:
            t = self.albedo.unsqueeze(0).permute(0, 3, 1, 2).to(self.device)
            t = torch.nn.functional.interpolate(t, (res_H, res_W), mode="bilinear", align_corners=False)
            self.albedo = t.squeeze(0).permute(1, 2, 0).contiguous()
