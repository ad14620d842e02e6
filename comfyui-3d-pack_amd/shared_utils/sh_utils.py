"""SH helpers the renderer mirror needs (reference: shared_utils/sh_utils.py:26, 57-112, 114-117).  The rasterizer evaluates the SH
polynomial inside the HIP kernel (csrc/gs_math.h); `eval_sh` here is the host-side counterpart of the reference's function of that
name, written as one basis-matrix contraction, and is held to the reference's outputs by tests/test_ref_conventions.py."""
import torch

C0 = 0.28209479177387814
_C1 = 0.4886025119029199
_C2 = (1.0925484305920792, -1.0925484305920792, 0.31539156525252005, -1.0925484305920792, 0.5462742152960396)
_C3 = (-0.5900435899266435, 2.890611442640554, -0.4570457994644658, 0.3731763325901154, -0.4570457994644658, 1.445305721320277,
       -0.5900435899266435)
_C4 = (2.5033429417967046, -1.7701307697799304, 0.9461746957575601, -0.6690465435572892, 0.10578554691520431, -0.6690465435572892,
       0.47308734787878004, -1.7701307697799304, 0.6258357354491761)


def sh_basis(deg, dirs):
    """real SH basis values [..., (deg+1)^2] at unit directions [..., 3], in the coefficient order of the 3DGS code base"""
    if not 0 <= deg <= 4:
        raise ValueError("sh degree must be 0..4")      # the rasterizer kernels stop at 3; the reference's host-side eval_sh goes to 4 (sh_utils.py:101-111)
    x, y, z = dirs[..., 0], dirs[..., 1], dirs[..., 2]
    cols = [torch.full_like(x, C0)]
    if deg >= 1:
        cols += [-_C1 * y, _C1 * z, -_C1 * x]
    if deg >= 2:
        xx, yy, zz = x * x, y * y, z * z
        cols += [_C2[0] * x * y, _C2[1] * y * z, _C2[2] * (2 * zz - xx - yy), _C2[3] * x * z, _C2[4] * (xx - yy)]
    if deg >= 3:
        cols += [_C3[0] * y * (3 * xx - yy), _C3[1] * x * y * z, _C3[2] * y * (4 * zz - xx - yy), _C3[3] * z * (2 * zz - 3 * xx - 3 * yy),
                 _C3[4] * x * (4 * zz - xx - yy), _C3[5] * z * (xx - yy), _C3[6] * x * (xx - 3 * yy)]
    if deg >= 4:
        cols += [_C4[0] * x * y * (xx - yy), _C4[1] * y * z * (3 * xx - yy), _C4[2] * x * y * (7 * zz - 1), _C4[3] * y * z * (7 * zz - 3),
                 _C4[4] * (zz * (35 * zz - 30) + 3), _C4[5] * x * z * (7 * zz - 3), _C4[6] * (xx - yy) * (7 * zz - 1), _C4[7] * x * z * (xx - 3 * yy),
                 _C4[8] * (xx * (xx - 3 * yy) - yy * (3 * xx - yy))]
    return torch.stack(cols, dim=-1)


def eval_sh(deg, sh, dirs):
    """sh [..., C, >=(deg+1)^2], dirs [..., 3] (unit) -> [..., C]: sum_k basis_k(dir) * sh[..., k]"""
    K = (deg + 1) ** 2
    if sh.shape[-1] < K:
        raise ValueError("eval_sh: degree %d needs %d coefficients, got %d" % (deg, K, sh.shape[-1]))
    return (sh[..., :K] * sh_basis(deg, dirs).unsqueeze(-2)).sum(-1)


def RGB2SH(rgb):
    return (rgb - 0.5) / C0


def SH2RGB(sh):
    return sh * C0 + 0.5
