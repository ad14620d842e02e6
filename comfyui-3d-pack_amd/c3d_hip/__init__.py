"""ctypes binding of libc3d_hip.so -- the C-ABI declared in include/c3d_gs.h / include/c3d_mesh.h.

PyTorch is used here only as the owner of device memory and of the current HIP stream; every
tensor crosses the boundary as a raw device pointer.  There is NO CPU or eager fallback: if the
library cannot be loaded, or a tensor is not on a HIP device, the call raises.
"""
import ctypes as C
import os
import threading

# The library sets no process-wide runtime switches.  HIP_FORCE_DEV_KERNARG=1 (kernel arguments in device memory: ~2 us off every dependent dispatch,
# profiles/r03/r03u_kernarg.txt) is a recommendation for the HOST process, documented in INTEGRATION.md; bench.py sets it for itself.

import torch

_PKG = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB_PATH = os.path.join(_PKG, "lib", "libc3d_hip.so")
_lock = threading.Lock()
_lib = None

vp, i32, i64, f32, sz = C.c_void_p, C.c_int32, C.c_int64, C.c_float, C.c_size_t


class GsSettings(C.Structure):
    """struct c3d_gs_settings (include/c3d_gs.h)"""
    _fields_ = [("image_height", i32), ("image_width", i32), ("tanfovx", f32), ("tanfovy", f32),
                ("scale_modifier", f32), ("sh_degree", i32), ("prefiltered", i32), ("debug", i32),
                ("bg", vp), ("viewmatrix", vp), ("projmatrix", vp), ("campos", vp), ("flags", i32), ("sh_coeffs", i32)]


GS_FLAG_EXACT_DSCALE = 1      # C3D_GS_FLAG_EXACT_DSCALE
GS_FLAG_FORWARD_ONLY = 2      # C3D_GS_FLAG_FORWARD_ONLY
GS_FLAG_KEEP_RECORD_BASES = 4      # C3D_GS_FLAG_KEEP_RECORD_BASES


class GsLoss(C.Structure):
    """struct c3d_gs_loss"""
    _fields_ = [("w_l1", f32), ("w_l2", f32), ("w_alpha_mse", f32), ("scale", f32), ("w_ssim", f32)]


class AdamTensor(C.Structure):
    """struct c3d_adam_tensor (include/c3d_optim.h)"""
    _fields_ = [("param", vp), ("grad", vp), ("exp_avg", vp), ("exp_avg_sq", vp), ("n", i64), ("step", i64),
                ("lr", C.c_double), ("beta1", C.c_double), ("beta2", C.c_double), ("eps", C.c_double)]


ADAM_MAX_TENSORS = 16      # C3D_ADAM_MAX_TENSORS

_SIGNATURES = {
    # name: (restype, argtypes)
    "c3d_last_error": (C.c_char_p, []),
    "c3d_version": (C.c_int, []),
    "c3d_gs_geom_bytes": (sz, [i32]),
    "c3d_gs_binning_bytes": (sz, [i64, i32, i32]),
    "c3d_gs_image_bytes": (sz, [i32, i32]),
    "c3d_gs_backward_scratch_bytes": (sz, [i32, i64]),
    "c3d_gs_forward_project": (C.c_int, [C.POINTER(GsSettings), i32, i32] + [vp] * 7 + [vp, vp, C.POINTER(i64), vp]),
    "c3d_gs_forward_render": (C.c_int, [C.POINTER(GsSettings), i32, i32, vp, vp, i64, vp, vp, vp, vp, vp, vp]),
    "c3d_gs_forward_nosync": (C.c_int, [C.POINTER(GsSettings), i32, i32] + [vp] * 7 + [vp, vp, i64, i64, vp, vp, vp, vp, vp, vp, vp, vp, vp]),
    "c3d_gs_forward_raw_nosync": (C.c_int, [C.POINTER(GsSettings), i32] + [vp] * 6 + [vp, vp, i64, i64, vp, vp, vp, vp, vp, vp, vp, vp, vp]),
    "c3d_gs_wait_count": (C.c_int, [vp, C.c_uint32, i64, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]),
    "c3d_gs_backward": (C.c_int, [C.POINTER(GsSettings), i32, i32] + [vp] * 6 + [vp, vp, i64, vp, vp] + [vp] * 3 + [vp] * 8 + [vp, vp]),
    "c3d_gs_forward_project_raw": (C.c_int, [C.POINTER(GsSettings), i32] + [vp] * 6 + [vp, vp, C.POINTER(i64), vp]),
    "c3d_gs_backward_raw": (C.c_int, [C.POINTER(GsSettings), i32] + [vp] * 5 + [vp, vp, i64, vp, vp] + [vp] * 3 + [vp] * 7 + [vp, i32, vp]),
    "c3d_gs_step_workspace_bytes": (sz, [i32, i32, i32, i64, i32]),
    "c3d_gs_render_workspace_bytes": (sz, [i32, i32, i32, i64, i32]),
    "c3d_gs_train_views_raw": (C.c_int, [C.POINTER(GsSettings), i32, i32] + [vp] * 6 + [C.POINTER(vp), C.POINTER(vp), C.POINTER(vp), C.POINTER(GsLoss)] + [vp] * 6 + [vp, i64, i32, i32, vp, vp, vp, vp]),
    "c3d_gs_render_views_raw": (C.c_int, [C.POINTER(GsSettings), i32, i32] + [vp] * 6 + [C.POINTER(vp)] * 4 + [i64, i32, vp, i64, vp, vp]),
    "c3d_gs_forward_views_raw": (C.c_int, [C.POINTER(GsSettings), i32, i32] + [vp] * 6 + [C.POINTER(vp)] * 4 + [i64, i32, vp, vp, vp]),
    "c3d_gs_step_param_backward_range": (C.c_int, [C.POINTER(GsSettings), i32, i32] + [vp] * 5 + [vp] * 6 + [i64, i32, vp, i32, i32, vp]),
    "c3d_gs_backward_views_raw": (C.c_int, [C.POINTER(GsSettings), i32, i32] + [vp] * 5 + [C.POINTER(vp)] * 3 + [vp] * 6 + [i64, i32, i32, vp, vp]),
    "c3d_gs_step_read_view": (C.c_int, [i32, i32, i32, i64, vp, i32, vp, vp, vp]),
    "c3d_gs_step_accumulate_densify_stats": (C.c_int, [i32, i32, i32, i64, vp, i32, vp, vp, vp, vp]),
    "c3d_gs_mark_visible": (C.c_int, [i32, vp, vp, vp, vp, vp]),
    "c3d_gs_debug_state": (C.c_int, [i32, i32, i32, vp, i64, vp] + [vp] * 7 + [vp]),
    "c3d_adam_step": (C.c_int, [vp, vp, vp, vp, i64, C.c_double, C.c_double, C.c_double, C.c_double, i64, vp]),
    "c3d_adam_step_multi": (C.c_int, [C.POINTER(AdamTensor), i32, vp]),
    "c3d_knn_scratch_bytes": (sz, [i32]),
    "c3d_knn3_mean_dist2": (C.c_int, [vp, i32, C.POINTER(C.c_float), C.POINTER(C.c_float), vp, vp, vp]),
    "c3d_msssim_workspace_bytes": (sz, [i32, i32, i32, i32]),
    "c3d_msssim_value_grad": (C.c_int, [vp, vp, vp, i32, i32, i32, i32, i32, f32, i32, vp, vp, vp, vp]),
    "c3d_reduce_ranks_f32": (C.c_int, [vp, vp, i32, i64, C.c_float, vp]),
    "c3d_densify_plan_bytes": (sz, [i32]),
    "c3d_densify_plan": (C.c_int, [i32, vp, vp, vp, vp, f32, f32, f32, f32, vp, vp, vp]),
    "c3d_densify_fill": (C.c_int, [i32, vp, C.POINTER(C.c_uint32), vp, vp, vp, vp]),
    "c3d_gather_rows": (C.c_int, [i32, C.POINTER(vp), C.POINTER(vp), C.POINTER(i32), C.POINTER(i32), vp, vp, i64, vp]),
    "c3d_prof_enable": (C.c_int, [C.c_int]),
    "c3d_prof_select": (C.c_int, [C.c_ulonglong]),
    "c3d_prof_slots": (C.c_int, []),
    "c3d_prof_name": (C.c_char_p, [C.c_int]),
    "c3d_prof_read": (C.c_int, [C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_longlong)]),
    "c3d_test_scan_u32": (C.c_int, [vp, vp, i64, i32, vp]),
    "c3d_test_sort_pairs_u32": (C.c_int, [vp, vp, i64, i32, vp]),
    "c3d_test_sort_iota_u32": (C.c_int, [vp, vp, i64, i32, vp]),
    "c3d_test_sort_phases": (C.c_int, [vp]),
    "c3d_test_scan_wave": (C.c_int, [vp, vp, vp, vp, i64, vp]),
}


ABI_VERSION = 600      # c3d_version() of the library these signatures describe


def exported_symbols():
    """Names every include/*.h header declares (checked against the .so by the CPU test-suite)."""
    names = dict(_SIGNATURES)
    try:
        from . import mesh_sigs
        names.update(mesh_sigs.SIGNATURES)
    except ImportError:
        pass
    return names


def lib():
    """Load libc3d_hip.so.  With hipcc present build.build() runs first: a no-op when the source digest (csrc/*, include/*.h, flags)
    matches the one the .so was built from, a rebuild otherwise -- a stale library is never loaded under new ctypes signatures.  Without
    hipcc the .so must exist and report the ABI version this binding was written for.  Raises on failure; there is no fallback."""
    global _lib
    if _lib is not None:
        return _lib
    with _lock:
        if _lib is not None:
            return _lib
        from . import build as _b
        if os.path.exists(_b.HIPCC):
            try:
                _b.build()
            except Exception as e:  # loud failure, never a fallback
                raise RuntimeError("c3d_hip: %s could not be built: %s" % (LIB_PATH, e))
        elif not os.path.exists(LIB_PATH):
            raise RuntimeError("c3d_hip: %s is missing and there is no hipcc (%s) to build it" % (LIB_PATH, _b.HIPCC))
        l = C.CDLL(LIB_PATH)
        l.c3d_version.restype = C.c_int
        if l.c3d_version() != ABI_VERSION:
            raise RuntimeError("c3d_hip: %s reports ABI version %d, this binding needs %d (stale build: run c3d_hip/build.py --force)"
                               % (LIB_PATH, l.c3d_version(), ABI_VERSION))
        for name, (res, args) in exported_symbols().items():
            fn = getattr(l, name)
            fn.restype = res
            fn.argtypes = args
        _lib = l
    return _lib


def code_digest():
    """first 16 hex digits of the source digest of libc3d_hip.so (csrc/*, include/*.h, flags): stamped into every bench line and profile summary"""
    from . import build as _b
    return _b.code_digest()[:16]


def check(rc, what):
    if rc != 0:
        msg = lib().c3d_last_error()
        raise RuntimeError("%s failed (code %d): %s" % (what, rc, msg.decode() if msg else "?"))


def ptr(t):
    """device pointer of a tensor (None -> NULL).  Refuses non-HIP tensors: no CPU fallback."""
    if t is None:
        return None
    if not t.is_cuda:
        raise RuntimeError("c3d_hip: expected a tensor on a HIP device, got %s (the MI355X path has no CPU fallback)" % t.device)
    if not t.is_contiguous():
        raise RuntimeError("c3d_hip: tensor must be contiguous")
    return C.c_void_p(t.data_ptr())


def stream(device=None):
    return C.c_void_p(torch.cuda.current_stream(device).cuda_stream)


def f32c(t):
    """contiguous float32 view/copy of t (None and empty tensors -> None, like the dependency's 'empty = absent')."""
    if t is None or t.numel() == 0:
        return None
    if t.dtype != torch.float32:
        t = t.float()
    return t.contiguous()


def prof_enable(on=True, only=None):
    """only: names of the kernel groups to time (None = all).  Every timed launch costs two event records on its stream."""
    l = lib()
    mask = (1 << 64) - 1
    if only is not None:
        names = [l.c3d_prof_name(i).decode() for i in range(l.c3d_prof_slots())]
        mask = 0
        for n in only:
            mask |= 1 << names.index(n)
    l.c3d_prof_select(mask)
    l.c3d_prof_enable(1 if on else 0)


def prof_read():
    """{kernel group name: (total_ms, launches)} accumulated since prof_enable(True)."""
    l = lib()
    out = {}
    for i in range(l.c3d_prof_slots()):
        ms, n = C.c_double(0), C.c_longlong(0)
        l.c3d_prof_read(i, C.byref(ms), C.byref(n))
        if n.value:
            out[l.c3d_prof_name(i).decode()] = (ms.value, n.value)
    return out
