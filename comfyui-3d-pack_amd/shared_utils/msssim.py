"""Multi-scale SSIM.  Stand-in for pytorch_msssim.MS_SSIM(data_range=1, size_average=True, channel=3) used at
/root/reference/MVs_Algorithms/GaussianSplatting/main_3DGS.py:102,192 (the wheel is not vendored): Gaussian window 11,
sigma 1.5, five scales with the standard weights, 2x2 average pooling between scales.

Two implementations of the same published algorithm:
  * plain torch (dense separable grouped convolutions), the restatement everything is checked against and the path for CPU tensors and for
    gradients w.r.t. the first argument;
  * the fused HIP kernels of include/c3d_loss.h (csrc/msssim.hip) for the trainers' call pattern MS_SSIM(reference, rendered) on a HIP
    device -- value and d/d(rendered) in one library call: at 8 x 3 x 1080 x 1920 the torch chain costs 65 ms per step, nine times the
    rasterizer step it is the loss of (profiles/r02b)."""
import torch
import torch.nn.functional as F

_WEIGHTS = (0.0448, 0.2856, 0.3001, 0.2363, 0.1333)


def _window(size, sigma, device, dtype):
    x = torch.arange(size, device=device, dtype=dtype) - size // 2
    g = torch.exp(-(x ** 2) / (2 * sigma ** 2))
    return g / g.sum()


def _blur(x, w):
    c = x.shape[1]
    x = F.conv2d(x, w.view(1, 1, -1, 1).expand(c, 1, -1, 1), groups=c)
    return F.conv2d(x, w.view(1, 1, 1, -1).expand(c, 1, 1, -1), groups=c)


def _ssim_cs(x, y, w, data_range):
    c1, c2 = (0.01 * data_range) ** 2, (0.03 * data_range) ** 2
    mx, my = _blur(x, w), _blur(y, w)
    sxx, syy, sxy = _blur(x * x, w) - mx * mx, _blur(y * y, w) - my * my, _blur(x * y, w) - mx * my
    cs = (2 * sxy + c2) / (sxx + syy + c2)
    ssim = ((2 * mx * my + c1) / (mx * mx + my * my + c1)) * cs
    return ssim.flatten(2).mean(-1), cs.flatten(2).mean(-1)   # per image, per channel


class _MsSsimHip(torch.autograd.Function):
    """mean MS-SSIM(x, y) with the gradient w.r.t. y from c3d_msssim_value_grad (x is a constant: the reference image)"""

    @staticmethod
    def forward(ctx, x, y):
        import c3d_hip as _h
        B, C, H, W = y.shape
        lib = _h.lib()
        ws = torch.empty((lib.c3d_msssim_workspace_bytes(B, C, H, W),), dtype=torch.uint8, device=y.device)
        grad = torch.empty_like(y)
        val = torch.zeros((1,), dtype=torch.float32, device=y.device)
        with torch.cuda.device(y.device):
            _h.check(lib.c3d_msssim_value_grad(_h.ptr(x), _h.ptr(y), None, 0, B, C, H, W, 1.0, 0, _h.ptr(grad), _h.ptr(val), _h.ptr(ws), _h.stream(y.device)),
                     "c3d_msssim_value_grad")
        ctx.save_for_backward(grad)
        return val[0]

    @staticmethod
    def backward(ctx, g):
        (grad,) = ctx.saved_tensors
        return None, grad * g


def _hip_ok(x, y, weights, win_size, win_sigma, data_range):
    return (y.is_cuda and x.is_cuda and x.dtype == torch.float32 and y.dtype == torch.float32 and x.dim() == 4 and x.shape == y.shape and not x.requires_grad
            and tuple(weights) == _WEIGHTS and win_size == 11 and win_sigma == 1.5 and data_range == 1)


class MS_SSIM(torch.nn.Module):
    def __init__(self, data_range=1.0, size_average=True, channel=3, win_size=11, win_sigma=1.5, weights=_WEIGHTS):
        super().__init__()
        self.data_range, self.size_average, self.win_size, self.win_sigma, self.weights = data_range, size_average, win_size, win_sigma, weights
        self.use_hip = True             # False: always the torch restatement (tests compare the two)

    def forward(self, x, y):
        assert min(x.shape[-2:]) > (self.win_size - 1) * 2 ** (len(self.weights) - 1), "image too small for 5-scale MS-SSIM"
        if self.use_hip and self.size_average and _hip_ok(x, y, self.weights, self.win_size, self.win_sigma, self.data_range):
            return _MsSsimHip.apply(x.contiguous(), y.contiguous())
        w = _window(self.win_size, self.win_sigma, x.device, x.dtype)
        mcs = []
        for i in range(len(self.weights)):
            ssim, cs = _ssim_cs(x, y, w, self.data_range)
            if i < len(self.weights) - 1:
                mcs.append(torch.relu(cs))
                pad = [s % 2 for s in x.shape[2:]]
                x, y = F.avg_pool2d(x, 2, padding=pad), F.avg_pool2d(y, 2, padding=pad)
        vals = torch.stack(mcs + [torch.relu(ssim)], dim=0)                     # [levels, B, C]
        wts = torch.tensor(self.weights, device=x.device, dtype=x.dtype).view(-1, 1, 1)
        out = torch.prod(vals ** wts, dim=0)                                    # [B, C]
        return out.mean() if self.size_average else out.mean(1)
