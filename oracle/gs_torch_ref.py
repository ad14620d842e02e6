"""Independent float64 torch-autograd restatement of the 3DGS rasterizer -- TEST INFRASTRUCTURE.

Purpose: pin oracle/gs_oracle.c (whose backward is hand-derived) against gradients that torch
derives by itself.  Dense [pixels x Gaussians] formulation, so only for tiny scenes
(N <= a few hundred, <= 64x64 px).  Only tests/ import this.  PARITY UNPINNED (see gs_oracle.c).

Semantics follow SURVEY.md Appendix A (steps 1-10); call-site contract:
/root/reference/MVs_Algorithms/GaussianSplatting/main_3DGS_renderer.py:849-862,927-936;
SH basis /root/reference/shared_utils/sh_utils.py:57-100.
Two places where the dependency's backward is *not* the autograd derivative are mimicked so the
comparison is exact: the alpha cap min(0.99, .) passes gradient straight through, and the
+-1.3 tan(fov) clamp of the projected centre blocks the gradient only of the clamped coordinate.
"""
import torch

C0 = 0.28209479177387814
C1 = 0.4886025119029199
C2 = [1.0925484305920792, -1.0925484305920792, 0.31539156525252005, -1.0925484305920792, 0.5462742152960396]
C3 = [-0.5900435899266435, 2.890611442640554, -0.4570457994644658, 0.3731763325901154, -0.4570457994644658,
      1.445305721320277, -0.5900435899266435]


def eval_sh(deg, sh, d):
    """sh [N,K,3], d [N,3] unit -> [N,3]"""
    x, y, z = d[:, 0:1], d[:, 1:2], d[:, 2:3]
    r = C0 * sh[:, 0]
    if deg > 0:
        r = r - C1 * y * sh[:, 1] + C1 * z * sh[:, 2] - C1 * x * sh[:, 3]
    if deg > 1:
        xx, yy, zz, xy, yz, xz = x * x, y * y, z * z, x * y, y * z, x * z
        r = (r + C2[0] * xy * sh[:, 4] + C2[1] * yz * sh[:, 5] + C2[2] * (2 * zz - xx - yy) * sh[:, 6]
             + C2[3] * xz * sh[:, 7] + C2[4] * (xx - yy) * sh[:, 8])
    if deg > 2:
        r = (r + C3[0] * y * (3 * xx - yy) * sh[:, 9] + C3[1] * xy * z * sh[:, 10]
             + C3[2] * y * (4 * zz - xx - yy) * sh[:, 11] + C3[3] * z * (2 * zz - 3 * xx - 3 * yy) * sh[:, 12]
             + C3[4] * x * (4 * zz - xx - yy) * sh[:, 13] + C3[5] * z * (xx - yy) * sh[:, 14]
             + C3[6] * x * (xx - 3 * yy) * sh[:, 15])
    return r


def quat_to_rot(q):
    r, x, y, z = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
    R = torch.stack([1 - 2 * (y * y + z * z), 2 * (x * y - r * z), 2 * (x * z + r * y),
                     2 * (x * y + r * z), 1 - 2 * (x * x + z * z), 2 * (y * z - r * x),
                     2 * (x * z - r * y), 2 * (y * z + r * x), 1 - 2 * (x * x + y * y)], dim=-1)
    return R.reshape(-1, 3, 3)


def render(means3D, opacities, settings, shs=None, colors_precomp=None, scales=None, rotations=None, cov3D_precomp=None):
    """All tensors float64 on CPU.  -> color[3,H,W], radii[N], depth[1,H,W], alpha[1,H,W]"""
    dt = means3D.dtype
    H, W = int(settings["image_height"]), int(settings["image_width"])
    tfx, tfy = float(settings["tanfovx"]), float(settings["tanfovy"])
    mod = float(settings.get("scale_modifier", 1.0))
    view = torch.as_tensor(settings["viewmatrix"], dtype=dt).reshape(4, 4)
    proj = torch.as_tensor(settings["projmatrix"], dtype=dt).reshape(4, 4)
    campos = torch.as_tensor(settings["campos"], dtype=dt).reshape(3)
    bg = torch.as_tensor(settings["bg"], dtype=dt).reshape(3)
    deg = int(settings["sh_degree"])
    N = means3D.shape[0]
    fx, fy = W / (2 * tfx), H / (2 * tfy)
    gx, gy = (W + 15) // 16, (H + 15) // 16

    hom = torch.cat([means3D, torch.ones(N, 1, dtype=dt)], dim=1)
    pv = (hom @ view)[:, :3]
    ph = hom @ proj
    pw = 1.0 / (ph[:, 3] + 1e-7)
    ndc = ph[:, :2] * pw[:, None]
    vis = pv[:, 2] > 0.2

    if cov3D_precomp is None:
        R = quat_to_rot(rotations)
        Mm = R * (mod * scales)[:, None, :]
        Sigma = Mm @ Mm.transpose(1, 2)
    else:
        c = cov3D_precomp
        Sigma = torch.stack([c[:, 0], c[:, 1], c[:, 2], c[:, 1], c[:, 3], c[:, 4], c[:, 2], c[:, 4], c[:, 5]], dim=-1).reshape(-1, 3, 3)

    tz = pv[:, 2]
    safe_tz = torch.where(vis, tz, torch.ones_like(tz))
    limx, limy = 1.3 * tfx, 1.3 * tfy
    txtz, tytz = pv[:, 0] / safe_tz, pv[:, 1] / safe_tz
    # clamped coordinate: value lim*tz, no gradient (dependency's x_grad_mul = 0 behaviour)
    tx = torch.where((txtz < -limx) | (txtz > limx), (txtz.clamp(-limx, limx) * safe_tz).detach(), pv[:, 0])
    ty = torch.where((tytz < -limy) | (tytz > limy), (tytz.clamp(-limy, limy) * safe_tz).detach(), pv[:, 1])
    zero = torch.zeros_like(tz)
    J = torch.stack([fx / safe_tz, zero, -(fx * tx) / (safe_tz * safe_tz),
                     zero, fy / safe_tz, -(fy * ty) / (safe_tz * safe_tz)], dim=-1).reshape(-1, 2, 3)
    Wm = view[:3, :3].t()  # w2c rotation
    T2 = J @ Wm
    cov = T2 @ Sigma @ T2.transpose(1, 2)
    a = cov[:, 0, 0] + 0.3
    b = cov[:, 0, 1]
    c_ = cov[:, 1, 1] + 0.3
    det = a * c_ - b * b
    vis = vis & (det != 0)
    sdet = torch.where(det != 0, det, torch.ones_like(det))
    conx, cony, conz = c_ / sdet, -b / sdet, a / sdet
    mid = 0.5 * (a + c_)
    lam = mid + torch.sqrt(torch.clamp(mid * mid - det, min=0.1))
    radius = torch.ceil(3.0 * torch.sqrt(lam)).detach()
    px = ((ndc[:, 0] + 1) * W - 1) * 0.5
    py = ((ndc[:, 1] + 1) * H - 1) * 0.5
    pxd, pyd = px.detach(), py.detach()
    trunc = lambda v: torch.trunc(v).to(torch.int64)
    x0 = trunc((pxd - radius) / 16).clamp(0, gx); x1 = trunc((pxd + radius + 15) / 16).clamp(0, gx)
    y0 = trunc((pyd - radius) / 16).clamp(0, gy); y1 = trunc((pyd + radius + 15) / 16).clamp(0, gy)
    vis = vis & (((x1 - x0) * (y1 - y0)) > 0)
    radii = torch.where(vis, radius, torch.zeros_like(radius)).to(torch.int32)

    if colors_precomp is None:
        d = means3D - campos[None]
        d = d / d.norm(dim=1, keepdim=True)
        rgb = torch.clamp_min(eval_sh(deg, shs, d) + 0.5, 0.0)
    else:
        rgb = colors_precomp

    order = torch.argsort(torch.where(vis, tz, torch.full_like(tz, float("inf"))).detach(), stable=True)
    vis_o = vis[order]
    ys, xs = torch.meshgrid(torch.arange(H), torch.arange(W), indexing="ij")
    X = xs.reshape(-1, 1).to(dt); Y = ys.reshape(-1, 1).to(dt)
    TX = (xs.reshape(-1, 1) // 16); TY = (ys.reshape(-1, 1) // 16)
    o = lambda t: t[order][None, :]
    dx = o(px) - X; dy = o(py) - Y
    power = -0.5 * (o(conx) * dx * dx + o(conz) * dy * dy) - o(cony) * dx * dy
    in_rect = (TX >= o(x0)) & (TX < o(x1)) & (TY >= o(y0)) & (TY < o(y1)) & vis_o[None, :]
    G = torch.exp(torch.clamp(power, max=0.0))
    araw = opacities.reshape(-1)[order][None, :] * G
    alpha = araw + (torch.clamp(araw, max=0.99) - araw).detach()
    valid = in_rect & (power <= 0) & (alpha >= 1.0 / 255.0)
    aeff = torch.where(valid, alpha, torch.zeros_like(alpha))
    one_m = 1 - aeff
    T_incl = torch.cumprod(one_m, dim=1)
    T_excl = torch.cat([torch.ones_like(T_incl[:, :1]), T_incl[:, :-1]], dim=1)
    stop = valid & ((T_excl * one_m).detach() < 1e-4)
    done = torch.cumsum(stop.to(torch.int64), dim=1) > 0
    aeff = torch.where(done, torch.zeros_like(aeff), aeff)
    one_m = 1 - aeff
    T_incl = torch.cumprod(one_m, dim=1)
    T_excl = torch.cat([torch.ones_like(T_incl[:, :1]), T_incl[:, :-1]], dim=1)
    w = aeff * T_excl
    T_final = T_incl[:, -1] if N > 0 else torch.ones(H * W, dtype=dt)
    color = w @ rgb[order] + T_final[:, None] * bg[None, :]
    depth = w @ tz[order]
    acc = w.sum(dim=1)
    return color.t().reshape(3, H, W), radii, depth.reshape(1, H, W), acc.reshape(1, H, W)
