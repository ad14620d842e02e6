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
redone_calls = 0        # sync-free forward calls whose view did not fit its buffers and was rendered again at the exact count before forward() returned -- telemetry
beyond_hint_calls = 0   # ... whose pair count exceeded the launch hint (workgroups looped; exact) -- telemetry
sync_free_calls = 0     # forward calls that went through c3d_gs_forward_nosync / c3d_gs_forward_raw_nosync -- telemetry

# ---- the forward pass without stalling the GPU, and still exact -------------------------------------------------------------------------------------
# The wheel sizes its binning buffer from the exact pair count: project, copy one number to the host, WAIT, allocate, enqueue the second half -- the GPU idles through the
# round trip on every view.  main_3DGS_renderer.py:927-936 never returns an incomplete image; neither does this module.  After the first call of a (device, H, W) shape:
#   * the whole forward is enqueued at once (c3d_gs_forward_nosync, include/c3d_gs.h): launches sized for a HINT (1.25 x the largest pair count seen, scaled with the point
#     count while a model densifies), buffers for _ROOM x that, workgroups that loop when the count exceeds the hint -- every count the buffers hold is rendered exactly, and a
#     count within the hint costs nothing extra;
#   * the kernel that finds the pair count stores it straight into pinned host memory, and forward() WAITS for that word before it returns (c3d_gs_wait_count): by then
#     emission, tile sort and compositing are queued behind it, so the GPU never idles -- and the host knows, before the image leaves, whether the view fitted its buffers.
#     If not (a count more than _HEADROOM x _ROOM = 5 x the largest seen), the second half is rendered again at the exact count on the geometry already projected.
# Result: exact for every call, differentiated or not, bit for bit the synchronous path -- and the host still runs ahead of the GPU by everything behind the scan.
# sync_free("unverified") skips the wait (the host may run whole calls ahead; status words are looked at one call late; a view beyond its buffers comes back as NaN planes
# and the examination raises); sync_free(False) is the wheel's synchronous path for every call.  Calls that are not differentiated (grad mode off, or no input requires a
# gradient) render with C3D_GS_FLAG_FORWARD_ONLY: nothing a backward pass would read is recorded.
import collections
import threading
import warnings

_lock = threading.RLock()      # ComfyUI may run nodes on several threads: slot hand-out and examination are serialised (a few dictionary / list operations per call)
_SYNC_FREE = "verified"        # "verified" | "unverified" | False
_FORWARD_ONLY = True           # calls that are not differentiated render with C3D_GS_FLAG_FORWARD_ONLY (forward_only(False): the A/B switch of bench.py)
_HEADROOM, _ROOM, _SLACK = 1.25, 4, 1 << 16      # launch hint = headroom x largest count seen + slack; buffers = _ROOM x hint
_MAX_PAIRS = 0x3FFFFFF0       # the library's own limit (the chained scans' status words)
_WAIT_US = 20_000_000         # c3d_gs_wait_count gives up after 20 s (a wedged device)
_MODELS = 8      # point counts remembered per (device, H, W)
_learnt = {}     # (device index, H, W) -> OrderedDict {N: largest pair count seen with N points}, most recently used last
_SLOTS = 64      # status slots per device: calls whose status words may be on their way to the host at once
_SENTINEL = -1   # 0xFFFFFFFF: neither a flag word (bits 0-2) nor a pair count (< 2^30)
_rings = {}      # device index -> _Ring


class _Ring:
    """per device: _SLOTS status slots.  Each has two int32 words on the device (what the kernels write), two in pinned host memory where c3d_gs_forward_nosync copies them
    at the END of the call (`pin`: the fault bit can be raised by any kernel), and two more pinned words the emit-offset scan stores {bits, pair count} into ITSELF, early
    (`cnt`).  The host presets pinned words to the sentinel and later just looks at them: no event, no allocation per call.  Slots are handed out round-robin and examined
    first-in first-out, so the slot about to be reused is always the oldest one still pending."""

    def __init__(self, dev):
        self.device = dev
        self.dev_words = torch.zeros((_SLOTS, 2), dtype=torch.int32, device=dev)
        self.pin = torch.full((_SLOTS, 2), _SENTINEL, dtype=torch.int32).pin_memory()
        self.cnt = torch.full((_SLOTS, 2), _SENTINEL, dtype=torch.int32).pin_memory()
        self.host, self.cnt_host = self.pin.numpy(), self.cnt.numpy()      # same memory: plain loads / stores from Python
        self.dev_ptr, self.pin_ptr, self.cnt_ptr = self.dev_words.data_ptr(), self.pin.data_ptr(), self.cnt.data_ptr()
        self.next = 0
        self.pending = []                                 # [[slot, key, N, capacity, verified]] calls whose final status words have not been examined, oldest first

    def examine(self, block=False, keep=_SLOTS):
        """retire the calls whose words have arrived; block: all of them, waiting for the GPU if need be; keep: wait until at most that many are left"""
        global beyond_hint_calls
        while self.pending:
            slot, key, n_points, cap, verified = self.pending[0]
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
            if verified:
                continue                                  # count and overflow were dealt with inside that call
            if flags & 4:
                beyond_hint_calls += 1
            _learn(key, n_points, seen)
            if flags & 1:
                raise RuntimeError("diff_gaussian_rasterization (MI355X): an earlier forward call (sync_free('unverified')) needed %d (tile, splat) pairs, more than the %d its "
                                   "buffers held (the count grew more than %g x from one call to the next): its colour / depth / alpha planes were returned as NaN.  The capacity "
                                   "has been regrown; the default sync_free('verified') renders such a view again at its exact count before the image leaves." % (seen, cap, _HEADROOM * _ROOM))


def sync_free(mode="verified"):
    """"verified" (default; True means the same): one enqueue per forward, the host waits for the pair count only and redoes a view that did not fit -- always exact.
    "unverified": no wait at all (see the head of this file).  False: the wheel's synchronous path (exact pair count read back between the two halves).  Returns the previous setting."""
    global _SYNC_FREE
    if mode is True:
        mode = "verified"
    if mode not in ("verified", "unverified", False):
        raise ValueError("sync_free: 'verified', 'unverified' or False")
    prev, _SYNC_FREE = _SYNC_FREE, mode
    return prev


def forward_only(on=True):
    """False: calls that are not differentiated still record the backward pass's state (what rounds 1-5 did; measurement switch).  Returns the previous setting."""
    global _FORWARD_ONLY
    prev, _FORWARD_ONLY = _FORWARD_ONLY, bool(on)
    return prev


def flush():
    """wait for the final status words of every sync-free forward call issued so far and examine them (raises as described above); afterwards
    last_num_rendered is the pair count of the most recent forward call"""
    with _lock:
        for ring in list(_rings.values()):
            ring.examine(block=True)


def _flush_at_exit():
    """a process whose LAST forward call faulted would otherwise never hear of it: examine what is pending before the interpreter goes"""
    try:
        if pending_calls():
            flush()
    except Exception as e:      # a fault reported this late can only be printed
        warnings.warn("diff_gaussian_rasterization (MI355X): %s" % e, RuntimeWarning)


def pending_calls():
    """sync-free forward calls whose final status words have not been examined yet"""
    with _lock:
        return sum(len(r.pending) for r in _rings.values())


def _learn(key, n_points, seen):
    """a pair count of a (device, H, W) shape rendered with n_points Gaussians has reached the host"""
    global last_num_rendered
    with _lock:
        last_num_rendered = int(seen)
        models = _learnt.setdefault(key, collections.OrderedDict())
        if seen > models.get(n_points, -1):
            models[n_points] = int(seen)
        models.move_to_end(n_points)
        while len(models) > _MODELS:
            models.popitem(last=False)


def _estimate(key, n_points):
    """-> the largest pair count seen for this shape at n_points Gaussians; for a point count not seen yet, that of the nearest one within a factor of two, scaled
    (a model that densifies or prunes keeps its pairs per Gaussian; looping workgroups and, beyond the buffers, the second rendering make a wrong guess exact); None: nothing to go by"""
    models = _learnt.get(key)
    if not models:
        return None
    if n_points in models:
        return models[n_points]
    n0 = min(models, key=lambda n: abs(n - n_points))
    if 2 * n0 < n_points or 2 * n_points < n0:
        return None
    return -(-models[n0] * n_points // n0)


def _capacity_for(key, n_points):
    """-> (launch hint, buffer capacity) of a sync-free forward of this shape (hint 0: launches sized for the buffers -- they hold every pair the view can have), or
    None: take the synchronous path (and learn the count)"""
    with _lock:
        ring = _rings.get(key[0])
        if ring is not None and ring.pending:
            ring.examine()
        seen = _estimate(key, n_points) if _SYNC_FREE else None
        if seen is None:
            return None
        bound = min(n_points * ((key[1] + 15) // 16) * ((key[2] + 15) // 16), _MAX_PAIRS)      # every Gaussian in every tile
        first = min(int(seen * _HEADROOM) + _SLACK, bound)
        cap = min(first * _ROOM, bound)
        return (first if first < cap else 0), max(cap, 1)


def _status_slot(dev, key, n_points, cap, verified):
    """-> (pending entry, device pointer, pinned pointer of the final words, pinned pointer of the early count words) of one sync-free call, registered for examination"""
    global sync_free_calls
    with _lock:
        sync_free_calls += 1
        ring = _rings.get(dev.index)
        if ring is None:
            ring = _rings[dev.index] = _Ring(dev)
        if len(ring.pending) >= _SLOTS:
            ring.examine(keep=_SLOTS - 1)                 # the slot handed out next is the oldest pending one: retire it first
        slot = ring.next
        ring.next = (slot + 1) % _SLOTS
        ring.host[slot] = _SENTINEL
        ring.cnt_host[slot] = _SENTINEL
        ring.pending.append([slot, key, n_points, cap, verified])
        return C.c_void_p(ring.dev_ptr + 8 * slot), C.c_void_p(ring.pin_ptr + 8 * slot), C.c_void_p(ring.cnt_ptr + 8 * slot)


def _differentiated(*tensors):
    """what autograd will do with the call: grad mode is off inside Function.forward, so it is looked at here, by the wrappers, before apply()"""
    return torch.is_grad_enabled() and any(isinstance(t, torch.Tensor) and t.requires_grad for t in tensors)


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


def _settings_struct(rs, keep, sh_coeffs=0, forward_only=False):
    """sh_coeffs: SH coefficients per channel the RAW parameters of the call store (f_rest.shape[1] + 1); 0 for the entry points that take M explicitly"""
    dev = rs.viewmatrix.device
    bg = _h.f32c(rs.bg.to(dev)); vm = _h.f32c(rs.viewmatrix); pm = _h.f32c(rs.projmatrix.to(dev)); cp = _h.f32c(rs.campos.to(dev))
    keep.extend([bg, vm, pm, cp])
    flags = (_h.GS_FLAG_EXACT_DSCALE if getattr(rs, "exact_dscale", False) else 0) | (_h.GS_FLAG_FORWARD_ONLY if forward_only else 0)
    return _h.GsSettings(int(rs.image_height), int(rs.image_width), float(rs.tanfovx), float(rs.tanfovy),
                         float(rs.scale_modifier), int(rs.sh_degree), int(bool(rs.prefiltered)), int(bool(rs.debug)),
                         _h.ptr(bg), _h.ptr(vm), _h.ptr(pm), _h.ptr(cp), flags, int(sh_coeffs))


class _ExactDscaleSettings(GaussianRasterizationSettings):
    """the 12-field settings tuple of the dependency plus one attribute the C-ABI carries per call (C3D_GS_FLAG_EXACT_DSCALE)"""
    exact_dscale = True


def with_exact_dscale(rs):
    """-> the same raster settings, asking the backward pass for the exact d/dscale (x scale_modifier) instead of the dependency's convention.
    Extension; a plain GaussianRasterizationSettings keeps the dependency's behaviour.  No process-wide switch exists."""
    return _ExactDscaleSettings(*rs)


def rasterize_gaussians(means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp, raster_settings):
    return _RasterizeGaussians.apply(means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp, raster_settings,
                                     _differentiated(means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp))


_NO_BACKWARD = ("diff_gaussian_rasterization (MI355X): this forward call was not differentiated when it ran (grad mode off, or no input required a gradient), so it was rendered "
                "forward-only (C3D_GS_FLAG_FORWARD_ONLY) and kept no state for a backward pass")


def _forward_state(lib, dev, N, H, W, st, project, render, nosync):
    """the part of forward() the plain and the raw-parameter entry points share: buffers, the choice of path, the library calls.
    project(radii, geom, nr_ref, s), render(radii, geom, num_rendered, binning, img, color, depth, alpha, s), nosync(radii, geom, cap, first, binning, img, color, depth, alpha,
    st_dev, st_host, cnt_host, s) issue the calls with the entry point's own parameter list; st: the call's settings struct (its flags are amended for a second rendering).
    -> (color, radii, depth, alpha, geom, binning, img, num_rendered)"""
    global redone_calls, beyond_hint_calls
    u8 = dict(dtype=torch.uint8, device=dev)
    s = _h.stream(dev)
    radii = torch.empty((N,), dtype=torch.int32, device=dev)
    geom = torch.empty((lib.c3d_gs_geom_bytes(N),), **u8)
    img = torch.empty((lib.c3d_gs_image_bytes(H, W),), **u8)
    color = torch.empty((3, H, W), dtype=torch.float32, device=dev)
    depth = torch.empty((1, H, W), dtype=torch.float32, device=dev)
    alpha = torch.empty((1, H, W), dtype=torch.float32, device=dev)
    key = (dev.index, H, W)
    caps = _capacity_for(key, N) if (N > 0 and H > 0 and W > 0) else None
    if caps is not None:      # one enqueue for the whole forward (see the head of this file)
        first, cap = caps
        verified = _SYNC_FREE == "verified"
        num_rendered = cap
        binning = torch.empty((lib.c3d_gs_binning_bytes(cap, H, W),), **u8)
        st_dev, st_host, cnt_host = _status_slot(dev, key, N, cap, verified)
        nosync(radii, geom, cap, first, binning, img, color, depth, alpha, st_dev, st_host, cnt_host if verified else None, s)
        if verified:          # the pair count, straight from the kernel that found it; everything behind that kernel is still queued or running
            bits, count = C.c_uint32(0), C.c_uint32(0)
            _h.check(lib.c3d_gs_wait_count(cnt_host, 0xFFFFFFFF, _WAIT_US, C.byref(bits), C.byref(count)), "c3d_gs_wait_count")
            _learn(key, N, count.value)
            if bits.value & 4:
                beyond_hint_calls += 1
            if bits.value & 1:      # more pairs than the buffers hold: the second half again, on the geometry already projected, at the exact count
                redone_calls += 1
                num_rendered = int(count.value)
                binning = torch.empty((lib.c3d_gs_binning_bytes(num_rendered, H, W),), **u8)
                st.flags |= _h.GS_FLAG_KEEP_RECORD_BASES
                try:
                    render(radii, geom, num_rendered, binning, img, color, depth, alpha, s)
                finally:
                    st.flags &= ~_h.GS_FLAG_KEEP_RECORD_BASES
    else:                     # the wheel's way: the exact pair count comes back to the host between the two halves
        nr = C.c_int64(0)
        project(radii, geom, nr, s)
        num_rendered = int(nr.value)
        if N > 0 and H > 0 and W > 0:
            _learn(key, N, num_rendered)
        binning = torch.empty((lib.c3d_gs_binning_bytes(num_rendered, H, W),), **u8)
        render(radii, geom, num_rendered, binning, img, color, depth, alpha, s)
    return color, radii, depth, alpha, geom, binning, img, num_rendered


class _RasterizeGaussians(torch.autograd.Function):
    @staticmethod
    def forward(ctx, means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp, raster_settings, differentiated=True):
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
            st = _settings_struct(rs, keep, forward_only=_FORWARD_ONLY and not differentiated)
            inputs = [_h.ptr(x) for x in (means3D_c, sh_c, col_c, op_c, sc_c, rot_c, cov_c)]

            def project(radii, geom, nr, s):
                _h.check(lib.c3d_gs_forward_project(C.byref(st), N, M, *inputs, _h.ptr(radii), _h.ptr(geom), C.byref(nr), s), "c3d_gs_forward_project")

            def render(radii, geom, num_rendered, binning, img, color, depth, alpha, s):
                _h.check(lib.c3d_gs_forward_render(C.byref(st), N, M, _h.ptr(radii), _h.ptr(geom), num_rendered, _h.ptr(binning), _h.ptr(img), _h.ptr(color), _h.ptr(depth),
                                                   _h.ptr(alpha), s), "c3d_gs_forward_render")

            def nosync(radii, geom, cap, first, binning, img, color, depth, alpha, st_dev, st_host, cnt_host, s):
                _h.check(lib.c3d_gs_forward_nosync(C.byref(st), N, M, *inputs, _h.ptr(radii), _h.ptr(geom), cap, first, _h.ptr(binning), _h.ptr(img), _h.ptr(color),
                                                   _h.ptr(depth), _h.ptr(alpha), st_dev, st_host, cnt_host, s), "c3d_gs_forward_nosync")

            color, radii, depth, alpha, geom, binning, img, num_rendered = _forward_state(lib, dev, N, H, W, st, project, render, nosync)
        ctx.raster_settings = rs
        ctx.num_rendered = num_rendered
        ctx.sizes = (N, M)
        ctx.forward_only = not differentiated
        ctx.present = (sh_c is not None, col_c is not None, sc_c is not None, cov_c is not None)
        ctx.mark_non_differentiable(radii)
        if differentiated:
            e = torch.empty(0, device=dev)
            ctx.save_for_backward(*(t if t is not None else e for t in (col_c, means3D_c, sc_c, rot_c, cov_c, radii, sh_c, geom, binning, img)))
        return color, radii, depth, alpha

    @staticmethod
    def backward(ctx, grad_color, grad_radii, grad_depth, grad_alpha):
        if ctx.forward_only:
            raise RuntimeError(_NO_BACKWARD)
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
                g_cov3D if has_cov else None, None, None)


class _RasterizeGaussiansRaw(torch.autograd.Function):
    """Extension: the same rasterizer fed with GaussianModel's RAW parameters; exp / sigmoid / normalize / cat and their backward
    passes run inside the projection kernels (c3d_gs_forward_project_raw / c3d_gs_backward_raw)."""

    @staticmethod
    def forward(ctx, means3D, means2D, f_dc, f_rest, opacity_raw, scaling_raw, rotation_raw, raster_settings, differentiated=True):
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
            st = _settings_struct(rs, keep, K, forward_only=_FORWARD_ONLY and not differentiated)
            inputs = [_h.ptr(x) for x in t]

            def project(radii, geom, nr, s):
                _h.check(lib.c3d_gs_forward_project_raw(C.byref(st), N, *inputs, _h.ptr(radii), _h.ptr(geom), C.byref(nr), s), "c3d_gs_forward_project_raw")

            def render(radii, geom, num_rendered, binning, img, color, depth, alpha, s):
                _h.check(lib.c3d_gs_forward_render(C.byref(st), N, K, _h.ptr(radii), _h.ptr(geom), num_rendered, _h.ptr(binning), _h.ptr(img), _h.ptr(color), _h.ptr(depth),
                                                   _h.ptr(alpha), s), "c3d_gs_forward_render")

            def nosync(radii, geom, cap, first, binning, img, color, depth, alpha, st_dev, st_host, cnt_host, s):
                _h.check(lib.c3d_gs_forward_raw_nosync(C.byref(st), N, *inputs, _h.ptr(radii), _h.ptr(geom), cap, first, _h.ptr(binning), _h.ptr(img), _h.ptr(color),
                                                       _h.ptr(depth), _h.ptr(alpha), st_dev, st_host, cnt_host, s), "c3d_gs_forward_raw_nosync")

            color, radii, depth, alpha, geom, binning, img, num_rendered = _forward_state(lib, dev, N, H, W, st, project, render, nosync)
        ctx.raster_settings, ctx.num_rendered, ctx.N, ctx.K = rs, num_rendered, N, K
        ctx.forward_only = not differentiated
        ctx.mark_non_differentiable(radii)
        if differentiated:
            e = torch.empty(0, device=dev)
            ctx.save_for_backward(*(x if x is not None else e for x in t), radii, geom, binning, img)
        return color, radii, depth, alpha

    @staticmethod
    def backward(ctx, grad_color, grad_radii, grad_depth, grad_alpha):
        if ctx.forward_only:
            raise RuntimeError(_NO_BACKWARD)
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
        return g_m3d, g_m2d, g_dc, g_rest, g_op, g_sc, g_rot, None, None


def raw_sh_coeffs(f_dc, f_rest):
    """SH coefficients per channel a GaussianModel's split storage holds (f_dc [N,1,3] + f_rest [N,K-1,3]): 16, 9, 4 or 1 -- what c3d_gs_settings.sh_coeffs carries"""
    if f_dc.dim() != 3 or f_dc.shape[1:] != (1, 3) or f_rest.dim() != 3 or f_rest.shape[2] != 3 or f_rest.shape[1] + 1 not in (1, 4, 9, 16):
        raise ValueError("raw SH storage must be f_dc [N,1,3] + f_rest [N,K-1,3] with K in (1, 4, 9, 16); got %s / %s" % (tuple(f_dc.shape), tuple(f_rest.shape)))
    return int(f_rest.shape[1]) + 1


def rasterize_gaussians_raw(means3D, means2D, f_dc, f_rest, opacity_raw, scaling_raw, rotation_raw, raster_settings):
    """(color, radii, depth, alpha) from RAW GaussianModel parameters (SH storage of degree 0..3).  Equivalent to
    GaussianRasterizer(settings)(means3D, means2D, sigmoid(opacity_raw), shs=cat(f_dc, f_rest), scales=exp(scaling_raw),
    rotations=normalize(rotation_raw)) -- one kernel instead of five torch ops each way."""
    return _RasterizeGaussiansRaw.apply(means3D, means2D, f_dc, f_rest, opacity_raw, scaling_raw, rotation_raw, raster_settings,
                                        _differentiated(means3D, means2D, f_dc, f_rest, opacity_raw, scaling_raw, rotation_raw))


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
