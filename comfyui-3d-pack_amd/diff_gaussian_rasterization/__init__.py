"""`diff_gaussian_rasterization` -- drop-in Python boundary over the MI355X HIP rasterizer.

Same names, argument meaning and error behaviour as the package the reference imports at
MVs_Algorithms/GaussianSplatting/main_3DGS_renderer.py:840-843 and calls at :849-864 / :927-936
(ashawkey fork: returns colour, radii, depth, alpha).  The arithmetic runs in libc3d_hip.so
(include/c3d_gs.h); this file only allocates tensors and wires autograd.
"""
from typing import NamedTuple

import ctypes as C
import torch
import torch.nn as nn

import c3d_hip as _h

last_num_rendered = 0   # (tile, splat) pairs of the most recent forward whose count has reached the host -- bench/telemetry only

# ---- pair capacity: the forward pass without its host round trip -----------------------------------------------------------------------------------
# The wheel sizes its binning buffer from the exact pair count and stalls the host for that one number on every call (so did rounds 1-4 here:
# c3d_gs_forward_project).  Here the first call of a (device, N, H, W) shape takes that synchronous path and LEARNS the count; later calls of the shape go through
# c3d_gs_forward_nosync with buffers and launches sized for a capacity above the largest count seen, the count stays on the device, and the two status words
# of the call come back through pinned memory and are examined when a later call starts -- the host runs ahead of the GPU instead of waiting for it once per view.
# A view that needs more pairs than the capacity is rendered incompletely (pairs beyond it are dropped); that is noticed one call late: a RuntimeWarning says so
# and the capacity regrows at once (calls still pending when the interpreter exits are examined then: flush() / atexit).  Headroom: 1.5 x the largest count seen when the call is differentiated (a training loop: counts drift slowly, and one incomplete
# gradient step is harmless), 3 x when it is not (inference through LGM / TGS / TRELLIS-style callers: the next object may be larger, and an incomplete image is what the
# user gets) -- empty capacity costs a few early-exiting workgroups per sort pass and bytes of a 288 GB memory.  sync_free(False) restores the wheel's behaviour (exact
# count, one synchronisation per call) for callers that cannot accept any of that.
import threading
import warnings

_lock = threading.RLock()      # ComfyUI may run nodes on several threads: slot hand-out and examination are serialised (a few dictionary / list operations per call)
_SYNC_FREE = True
_HEADROOM_GRAD, _HEADROOM_NOGRAD, _SLACK = 1.5, 3.0, 1 << 16      # capacity = headroom x largest count seen + slack
_cap = {}        # (device index, N, H, W) -> largest pair count seen for the shape
_SLOTS = 64      # status slots per device: calls whose status words may be on their way to the host at once
_SENTINEL = -1   # 0xFFFFFFFF: neither a flag word (bits 0-1) nor a pair count (< 2^30)
_rings = {}      # device index -> _Ring


class _Ring:
    """per device: _SLOTS status slots, each two int32 words on the device (what the kernels write) and two in pinned host memory (where c3d_gs_forward_nosync
    copies them at the end of the call).  The host presets a pinned slot to the sentinel and later just LOOKS at it: no event, no allocation, no synchronisation per call.
    Slots are handed out round-robin and examined first-in first-out, so the slot about to be reused is always the oldest one still pending."""

    def __init__(self, dev):
        self.device = dev
        self.dev_words = torch.zeros((_SLOTS, 2), dtype=torch.int32, device=dev)
        self.pin = torch.full((_SLOTS, 2), _SENTINEL, dtype=torch.int32).pin_memory()
        self.host = self.pin.numpy()                      # same memory: plain loads / stores from Python
        self.dev_ptr, self.pin_ptr = self.dev_words.data_ptr(), self.pin.data_ptr()
        self.next = 0
        self.pending = []                                 # [(slot, key, capacity)] calls whose status words have not been examined, oldest first

    def examine(self, block=False, keep=_SLOTS):
        """retire the calls whose words have arrived; block: all of them, waiting for the GPU if need be; keep: wait until at most that many are left"""
        while self.pending:
            slot, key, cap = self.pending[0]
            w = self.host[slot]
            if w[0] == _SENTINEL or w[1] == _SENTINEL:    # still on its way
                if not block and len(self.pending) <= keep:
                    return
                torch.cuda.synchronize(self.device)       # rare: flush(), or _SLOTS calls ahead of the GPU
                w = self.host[slot]
                if w[0] == _SENTINEL or w[1] == _SENTINEL:     # the device is idle and the words never came: that call failed before its copy was enqueued (it raised there)
                    self.pending.pop(0)
                    continue
            self.pending.pop(0)
            flags, seen = int(w[0]), int(w[1]) & 0xFFFFFFFF
            if flags & 2:
                raise RuntimeError("diff_gaussian_rasterization (MI355X): a chained-scan look-back of an earlier forward call timed out in the binning stage (device fault or a wedged workgroup)")
            _learn(key, seen)
            if flags & 1:
                warnings.warn("diff_gaussian_rasterization (MI355X): an earlier sync-free forward call needed %d (tile, splat) pairs, its buffers held %d -- that image (and its "
                              "gradient) is incomplete.  The capacity has been regrown; diff_gaussian_rasterization.sync_free(False) restores the exact synchronous path."
                              % (seen, cap), RuntimeWarning, stacklevel=4)


def sync_free(on=True):
    """False: every forward takes the wheel's synchronous path (exact pair count read back per call).  Returns the previous setting."""
    global _SYNC_FREE
    prev, _SYNC_FREE = _SYNC_FREE, bool(on)
    return prev


def flush():
    """wait for the status words of every sync-free forward call issued so far and examine them (warns / raises as described above); afterwards
    last_num_rendered is the pair count of the most recent forward call"""
    with _lock:
        for ring in list(_rings.values()):
            ring.examine(block=True)


def _flush_at_exit():
    """a process whose LAST forward call overflowed would otherwise never hear of it: examine what is pending before the interpreter goes"""
    try:
        if pending_calls():
            flush()
    except Exception as e:      # a fault reported this late can only be printed
        warnings.warn("diff_gaussian_rasterization (MI355X): %s" % e, RuntimeWarning)


def pending_calls():
    """sync-free forward calls whose status words have not been examined yet"""
    with _lock:
        return sum(len(r.pending) for r in _rings.values())


def _learn(key, seen):
    """a pair count of this shape has reached the host"""
    global last_num_rendered
    with _lock:
        last_num_rendered = int(seen)
        if seen > _cap.get(key, -1):
            _cap[key] = int(seen)


def _capacity_for(key, differentiated):
    """-> pair capacity for a sync-free forward of this shape, or None: take the synchronous path (and learn the count)"""
    with _lock:
        ring = _rings.get(key[0])
        if ring is not None and ring.pending:
            ring.examine()
        seen = _cap.get(key)
        if seen is None or not _SYNC_FREE:
            return None
        return min(int(seen * (_HEADROOM_GRAD if differentiated else _HEADROOM_NOGRAD)) + _SLACK, 0x3FFFFFF0)


def _status_slot(dev, key, cap):
    """-> (device pointer, pinned host pointer) of the status words of one sync-free call, registered for examination"""
    with _lock:
        ring = _rings.get(dev.index)
        if ring is None:
            ring = _rings[dev.index] = _Ring(dev)
        if len(ring.pending) >= _SLOTS:
            ring.examine(keep=_SLOTS - 1)                 # the slot handed out next is the oldest pending one: retire it first
        slot = ring.next
        ring.next = (slot + 1) % _SLOTS
        ring.host[slot] = _SENTINEL
        ring.pending.append((slot, key, cap))
        return C.c_void_p(ring.dev_ptr + 8 * slot), C.c_void_p(ring.pin_ptr + 8 * slot)


import atexit
atexit.register(_flush_at_exit)


class GaussianRasterizationSettings(NamedTuple):
    image_height: int
    image_width: int
    tanfovx: float
    tanfovy: float
    bg: torch.Tensor
    scale_modifier: float
    viewmatrix: torch.Tensor
    projmatrix: torch.Tensor
    sh_degree: int
    campos: torch.Tensor
    prefiltered: bool
    debug: bool


def _settings_struct(rs, keep, sh_coeffs=0):
    """sh_coeffs: SH coefficients per channel the RAW parameters of the call store (f_rest.shape[1] + 1); 0 for the entry points that take M explicitly"""
    dev = rs.viewmatrix.device
    bg = _h.f32c(rs.bg.to(dev)); vm = _h.f32c(rs.viewmatrix); pm = _h.f32c(rs.projmatrix.to(dev)); cp = _h.f32c(rs.campos.to(dev))
    keep.extend([bg, vm, pm, cp])
    return _h.GsSettings(int(rs.image_height), int(rs.image_width), float(rs.tanfovx), float(rs.tanfovy),
                         float(rs.scale_modifier), int(rs.sh_degree), int(bool(rs.prefiltered)), int(bool(rs.debug)),
                         _h.ptr(bg), _h.ptr(vm), _h.ptr(pm), _h.ptr(cp), _h.GS_FLAG_EXACT_DSCALE if getattr(rs, "exact_dscale", False) else 0, int(sh_coeffs))


class _ExactDscaleSettings(GaussianRasterizationSettings):
    """the 12-field settings tuple of the dependency plus one attribute the C-ABI carries per call (C3D_GS_FLAG_EXACT_DSCALE)"""
    exact_dscale = True


def with_exact_dscale(rs):
    """-> the same raster settings, asking the backward pass for the exact d/dscale (x scale_modifier) instead of the dependency's convention.
    Extension; a plain GaussianRasterizationSettings keeps the dependency's behaviour.  No process-wide switch exists."""
    return _ExactDscaleSettings(*rs)


def rasterize_gaussians(means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp, raster_settings):
    return _RasterizeGaussians.apply(means3D, means2D, sh, colors_precomp, opacities, scales, rotations,
                                     cov3Ds_precomp, raster_settings)


class _RasterizeGaussians(torch.autograd.Function):
    @staticmethod
    def forward(ctx, means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp, raster_settings):
        lib = _h.lib()
        rs = raster_settings
        dev = means3D.device
        if not means3D.is_cuda:
            raise RuntimeError("diff_gaussian_rasterization (MI355X): tensors must live on a HIP device; there is no CPU path")
        H, W = int(rs.image_height), int(rs.image_width)
        means3D_c = _h.f32c(means3D)
        N = 0 if means3D_c is None else means3D_c.shape[0]
        sh_c, col_c = _h.f32c(sh), _h.f32c(colors_precomp)
        op_c, sc_c, rot_c, cov_c = _h.f32c(opacities), _h.f32c(scales), _h.f32c(rotations), _h.f32c(cov3Ds_precomp)
        M = 0 if sh_c is None else sh_c.shape[1]
        keep = []
        with torch.cuda.device(dev):
            st = _settings_struct(rs, keep)
            s = _h.stream(dev)
            u8 = dict(dtype=torch.uint8, device=dev)
            radii = torch.empty((N,), dtype=torch.int32, device=dev)
            geom = torch.empty((lib.c3d_gs_geom_bytes(N),), **u8)
            img = torch.empty((lib.c3d_gs_image_bytes(H, W),), **u8)
            color = torch.empty((3, H, W), dtype=torch.float32, device=dev)
            depth = torch.empty((1, H, W), dtype=torch.float32, device=dev)
            alpha = torch.empty((1, H, W), dtype=torch.float32, device=dev)
            key = (dev.index, N, H, W)
            cap = _capacity_for(key, any(ctx.needs_input_grad)) if (N > 0 and H > 0 and W > 0) else None
            if cap is not None:      # sync-free: launches sized for the capacity, the pair count stays on the device (see _cap above)
                num_rendered = cap
                binning = torch.empty((lib.c3d_gs_binning_bytes(cap, H, W),), **u8)
                st_dev, st_host = _status_slot(dev, key, cap)
                _h.check(lib.c3d_gs_forward_nosync(C.byref(st), N, M, _h.ptr(means3D_c), _h.ptr(sh_c), _h.ptr(col_c), _h.ptr(op_c), _h.ptr(sc_c), _h.ptr(rot_c),
                                                   _h.ptr(cov_c), _h.ptr(radii), _h.ptr(geom), cap, _h.ptr(binning), _h.ptr(img), _h.ptr(color), _h.ptr(depth),
                                                   _h.ptr(alpha), st_dev, st_host, s), "c3d_gs_forward_nosync")
            else:
                nr = C.c_int64(0)
                _h.check(lib.c3d_gs_forward_project(C.byref(st), N, M, _h.ptr(means3D_c), _h.ptr(sh_c), _h.ptr(col_c), _h.ptr(op_c),
                                                    _h.ptr(sc_c), _h.ptr(rot_c), _h.ptr(cov_c), _h.ptr(radii), _h.ptr(geom),
                                                    C.byref(nr), s), "c3d_gs_forward_project")
                num_rendered = int(nr.value)
                _learn(key, num_rendered)
                binning = torch.empty((lib.c3d_gs_binning_bytes(num_rendered, H, W),), **u8)
                _h.check(lib.c3d_gs_forward_render(C.byref(st), N, M, _h.ptr(radii), _h.ptr(geom), num_rendered, _h.ptr(binning),
                                                   _h.ptr(img), _h.ptr(color), _h.ptr(depth), _h.ptr(alpha), s), "c3d_gs_forward_render")
        ctx.raster_settings = rs
        ctx.num_rendered = num_rendered
        ctx.sizes = (N, M)
        ctx.present = (sh_c is not None, col_c is not None, sc_c is not None, cov_c is not None)
        e = torch.empty(0, device=dev)
        ctx.save_for_backward(*(t if t is not None else e for t in (col_c, means3D_c, sc_c, rot_c, cov_c, radii, sh_c, geom, binning, img)))
        ctx.mark_non_differentiable(radii)
        return color, radii, depth, alpha

    @staticmethod
    def backward(ctx, grad_color, grad_radii, grad_depth, grad_alpha):
        lib = _h.lib()
        rs = ctx.raster_settings
        N, M = ctx.sizes
        col_c, means3D_c, sc_c, rot_c, cov_c, radii, sh_c, geom, binning, img = ctx.saved_tensors
        has_sh, has_col, has_sc, has_cov = ctx.present
        dev = geom.device
        keep = []
        nn_ = lambda t, ok: t if ok else None
        with torch.cuda.device(dev):
            st = _settings_struct(rs, keep)
            s = _h.stream(dev)
            f = dict(dtype=torch.float32, device=dev)
            g_means2D = torch.empty((N, 3), **f)
            g_colors = torch.empty((N, 3), **f)
            g_opacity = torch.empty((N, 1), **f)
            g_means3D = torch.empty((N, 3), **f)
            g_cov3D = torch.empty((N, 6), **f) if has_cov else None   # NULL -> the kernel skips the 24 B/Gaussian store
            g_sh = torch.empty((N, M, 3), **f) if has_sh else None
            g_scales = torch.empty((N, 3), **f) if has_sc else None
            g_rot = torch.empty((N, 4), **f) if has_sc else None
            scratch = torch.empty((lib.c3d_gs_backward_scratch_bytes(N, ctx.num_rendered),), dtype=torch.uint8, device=dev)
            gc = _h.f32c(grad_color)
            if gc is None:
                gc = torch.zeros((3, int(rs.image_height), int(rs.image_width)), **f)
            _h.check(lib.c3d_gs_backward(C.byref(st), N, M, _h.ptr(means3D_c if N else None), _h.ptr(nn_(sh_c, has_sh)),
                                         _h.ptr(nn_(col_c, has_col)), _h.ptr(nn_(sc_c, has_sc)), _h.ptr(nn_(rot_c, has_sc)),
                                         _h.ptr(nn_(cov_c, has_cov)), _h.ptr(radii), _h.ptr(geom), ctx.num_rendered,
                                         _h.ptr(binning), _h.ptr(img), _h.ptr(gc), _h.ptr(_h.f32c(grad_depth)),
                                         _h.ptr(_h.f32c(grad_alpha)), _h.ptr(g_means2D), _h.ptr(g_colors), _h.ptr(g_opacity),
                                         _h.ptr(g_means3D), _h.ptr(g_cov3D), _h.ptr(g_sh), _h.ptr(g_scales), _h.ptr(g_rot),
                                         _h.ptr(scratch), s), "c3d_gs_backward")
        return (g_means3D, g_means2D, g_sh, g_colors if has_col else None, g_opacity, g_scales, g_rot,
                g_cov3D if has_cov else None, None)


class _RasterizeGaussiansRaw(torch.autograd.Function):
    """Extension: the same rasterizer fed with GaussianModel's RAW parameters; exp / sigmoid / normalize / cat and their backward
    passes run inside the projection kernels (c3d_gs_forward_project_raw / c3d_gs_backward_raw)."""

    @staticmethod
    def forward(ctx, means3D, means2D, f_dc, f_rest, opacity_raw, scaling_raw, rotation_raw, raster_settings):
        lib = _h.lib()
        rs = raster_settings
        dev = means3D.device
        if not means3D.is_cuda:
            raise RuntimeError("diff_gaussian_rasterization (MI355X): tensors must live on a HIP device; there is no CPU path")
        K = raw_sh_coeffs(f_dc, f_rest)
        H, W = int(rs.image_height), int(rs.image_width)
        t = [_h.f32c(x) for x in (means3D, f_dc, f_rest, opacity_raw, scaling_raw, rotation_raw)]
        N = means3D.shape[0]
        keep = []
        with torch.cuda.device(dev):
            st = _settings_struct(rs, keep, K)
            s = _h.stream(dev)
            u8 = dict(dtype=torch.uint8, device=dev)
            radii = torch.empty((N,), dtype=torch.int32, device=dev)
            geom = torch.empty((lib.c3d_gs_geom_bytes(N),), **u8)
            img = torch.empty((lib.c3d_gs_image_bytes(H, W),), **u8)
            color = torch.empty((3, H, W), dtype=torch.float32, device=dev)
            depth = torch.empty((1, H, W), dtype=torch.float32, device=dev)
            alpha = torch.empty((1, H, W), dtype=torch.float32, device=dev)
            key = (dev.index, N, H, W)
            cap = _capacity_for(key, any(ctx.needs_input_grad)) if (N > 0 and H > 0 and W > 0) else None
            if cap is not None:      # sync-free (see _cap above)
                num_rendered = cap
                binning = torch.empty((lib.c3d_gs_binning_bytes(cap, H, W),), **u8)
                st_dev, st_host = _status_slot(dev, key, cap)
                _h.check(lib.c3d_gs_forward_raw_nosync(C.byref(st), N, *[_h.ptr(x) for x in t], _h.ptr(radii), _h.ptr(geom), cap, _h.ptr(binning), _h.ptr(img),
                                                       _h.ptr(color), _h.ptr(depth), _h.ptr(alpha), st_dev, st_host, s), "c3d_gs_forward_raw_nosync")
            else:
                nr = C.c_int64(0)
                _h.check(lib.c3d_gs_forward_project_raw(C.byref(st), N, *[_h.ptr(x) for x in t], _h.ptr(radii), _h.ptr(geom), C.byref(nr), s),
                         "c3d_gs_forward_project_raw")
                num_rendered = int(nr.value)
                _learn(key, num_rendered)
                binning = torch.empty((lib.c3d_gs_binning_bytes(num_rendered, H, W),), **u8)
                _h.check(lib.c3d_gs_forward_render(C.byref(st), N, K, _h.ptr(radii), _h.ptr(geom), num_rendered, _h.ptr(binning), _h.ptr(img),
                                                   _h.ptr(color), _h.ptr(depth), _h.ptr(alpha), s), "c3d_gs_forward_render")
        ctx.raster_settings, ctx.num_rendered, ctx.N, ctx.K = rs, num_rendered, N, K
        e = torch.empty(0, device=dev)
        ctx.save_for_backward(*(x if x is not None else e for x in t), radii, geom, binning, img)
        ctx.mark_non_differentiable(radii)
        return color, radii, depth, alpha

    @staticmethod
    def backward(ctx, grad_color, grad_radii, grad_depth, grad_alpha):
        lib = _h.lib()
        rs, N = ctx.raster_settings, ctx.N
        means3D, f_dc, f_rest, opacity_raw, scaling_raw, rotation_raw, radii, geom, binning, img = ctx.saved_tensors
        dev = geom.device
        keep = []
        with torch.cuda.device(dev):
            st = _settings_struct(rs, keep, ctx.K)
            f = dict(dtype=torch.float32, device=dev)
            g_m2d, g_m3d = torch.empty((N, 3), **f), torch.empty((N, 3), **f)
            g_dc, g_rest = torch.empty((N, 1, 3), **f), torch.empty((N, ctx.K - 1, 3), **f)
            g_op, g_sc, g_rot = torch.empty((N, 1), **f), torch.empty((N, 3), **f), torch.empty((N, 4), **f)
            scratch = torch.empty((lib.c3d_gs_backward_scratch_bytes(N, ctx.num_rendered),), dtype=torch.uint8, device=dev)
            gc = _h.f32c(grad_color)
            if gc is None:
                gc = torch.zeros((3, int(rs.image_height), int(rs.image_width)), **f)
            nz = lambda x: x if (N and x.numel()) else None
            _h.check(lib.c3d_gs_backward_raw(C.byref(st), N, _h.ptr(nz(means3D)), _h.ptr(nz(f_dc)), _h.ptr(nz(f_rest)), _h.ptr(nz(scaling_raw)),
                                             _h.ptr(nz(rotation_raw)), _h.ptr(radii), _h.ptr(geom), ctx.num_rendered, _h.ptr(binning), _h.ptr(img),
                                             _h.ptr(gc), _h.ptr(_h.f32c(grad_depth)), _h.ptr(_h.f32c(grad_alpha)), _h.ptr(g_m2d), _h.ptr(g_m3d),
                                             _h.ptr(g_dc), _h.ptr(nz(g_rest)), _h.ptr(g_op), _h.ptr(g_sc), _h.ptr(g_rot), _h.ptr(scratch), 0,
                                             _h.stream(dev)), "c3d_gs_backward_raw")
        return g_m3d, g_m2d, g_dc, g_rest, g_op, g_sc, g_rot, None


def raw_sh_coeffs(f_dc, f_rest):
    """SH coefficients per channel a GaussianModel's split storage holds (f_dc [N,1,3] + f_rest [N,K-1,3]): 16, 9, 4 or 1 -- what c3d_gs_settings.sh_coeffs carries"""
    if f_dc.dim() != 3 or f_dc.shape[1:] != (1, 3) or f_rest.dim() != 3 or f_rest.shape[2] != 3 or f_rest.shape[1] + 1 not in (1, 4, 9, 16):
        raise ValueError("raw SH storage must be f_dc [N,1,3] + f_rest [N,K-1,3] with K in (1, 4, 9, 16); got %s / %s" % (tuple(f_dc.shape), tuple(f_rest.shape)))
    return int(f_rest.shape[1]) + 1


def rasterize_gaussians_raw(means3D, means2D, f_dc, f_rest, opacity_raw, scaling_raw, rotation_raw, raster_settings):
    """(color, radii, depth, alpha) from RAW GaussianModel parameters (SH storage of degree 0..3).  Equivalent to
    GaussianRasterizer(settings)(means3D, means2D, sigmoid(opacity_raw), shs=cat(f_dc, f_rest), scales=exp(scaling_raw),
    rotations=normalize(rotation_raw)) -- one kernel instead of five torch ops each way."""
    return _RasterizeGaussiansRaw.apply(means3D, means2D, f_dc, f_rest, opacity_raw, scaling_raw, rotation_raw, raster_settings)


class GaussianRasterizer(nn.Module):
    def __init__(self, raster_settings):
        super().__init__()
        self.raster_settings = raster_settings

    def markVisible(self, positions):
        with torch.no_grad():
            rs = self.raster_settings
            lib = _h.lib()
            pos = _h.f32c(positions)
            N = 0 if pos is None else pos.shape[0]
            present = torch.empty((N,), dtype=torch.uint8, device=positions.device)
            vm, pm = _h.f32c(rs.viewmatrix), _h.f32c(rs.projmatrix)
            with torch.cuda.device(positions.device):
                _h.check(lib.c3d_gs_mark_visible(N, _h.ptr(pos), _h.ptr(vm), _h.ptr(pm), _h.ptr(present), _h.stream(positions.device)),
                         "c3d_gs_mark_visible")
        return present.bool()

    def forward(self, means3D, means2D, opacities, shs=None, colors_precomp=None, scales=None, rotations=None, cov3D_precomp=None):
        rs = self.raster_settings
        if (shs is None and colors_precomp is None) or (shs is not None and colors_precomp is not None):
            raise Exception('Please provide excatly one of either SHs or precomputed colors!')
        if ((scales is None or rotations is None) and cov3D_precomp is None) or \
                ((scales is not None or rotations is not None) and cov3D_precomp is not None):
            raise Exception('Please provide exactly one of either scale/rotation pair or precomputed 3D covariance!')
        e = torch.Tensor([]).to(means3D.device)
        shs = e if shs is None else shs
        colors_precomp = e if colors_precomp is None else colors_precomp
        scales = e if scales is None else scales
        rotations = e if rotations is None else rotations
        cov3D_precomp = e if cov3D_precomp is None else cov3D_precomp
        return rasterize_gaussians(means3D, means2D, shs, colors_precomp, opacities, scales, rotations, cov3D_precomp, rs)
