"""ctypes front-end of oracle/gs_oracle.c  (TEST INFRASTRUCTURE -- see that file's header).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this module.
PARITY UNPINNED (the reference ships no golden vectors for this path) except the input conventions, which
tests/test_ref_conventions.py holds to the reference's own Python (see gs_oracle.c).

The call signature mirrors the boundary the reference uses
(MVs_Algorithms/GaussianSplatting/main_3DGS_renderer.py:849-862, 927-936):
settings = image_height, image_width, tanfovx, tanfovy, bg, scale_modifier, viewmatrix,
projmatrix, sh_degree, campos; tensors = means3D, shs | colors_precomp, opacities,
scales+rotations | cov3D_precomp.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIBS = {}


def build(quiet=True):
    """Compile the oracle shared objects with gcc (Makefile in this directory)."""
    subprocess.run(["make", "-C", _HERE, "gs"], check=True,
                   stdout=subprocess.DEVNULL if quiet else None)


def _lib(dtype):
    dtype = np.dtype(dtype)
    if dtype in _LIBS:
        return _LIBS[dtype]
    name = "libgs_oracle_f32.so" if dtype == np.float32 else "libgs_oracle_f64.so"
    path = os.path.join(_HERE, "_build", name)
    build()                     # make: a no-op when the shared objects are newer than gs_oracle.c
    lib = C.CDLL(path)
    real = C.c_float if dtype == np.float32 else C.c_double
    P = C.c_void_p
    lib.gs_oracle_forward.restype = P
    lib.gs_oracle_forward.argtypes = [C.c_int] * 5 + [real] * 3 + [P] * 11 + [C.c_int] + [P] * 4 + [C.c_int]
    lib.gs_oracle_backward.restype = None
    lib.gs_oracle_backward.argtypes = [P] * 20 + [C.c_int]
    lib.gs_oracle_free.argtypes = [P]
    lib.gs_oracle_num_rendered.restype = C.c_int64
    lib.gs_oracle_num_rendered.argtypes = [P]
    for fn in ("point_list", "ranges", "xy", "depths", "conic_opacity", "rgb", "tiles_touched",
               "final_T", "n_contrib"):
        f = getattr(lib, "gs_oracle_" + fn)
        f.restype = P
        f.argtypes = [P]
    lib.gs_oracle_mark_visible.argtypes = [C.c_int, P, P, P, P]
    lib.gs_oracle_last_sort_seconds.restype = C.c_double
    lib.gs_oracle_last_sort_seconds.argtypes = []
    lib.gs_oracle_taint.restype = None
    lib.gs_oracle_taint.argtypes = [P, P, P]
    assert lib.gs_oracle_sizeof_real() == dtype.itemsize
    _LIBS[dtype] = lib
    return lib


def last_sort_seconds(dtype=np.float32):
    """wall seconds of the serial pair sort inside the last forward() of the `dtype` build (the CPU baseline's one single-threaded stage)"""
    return float(_lib(dtype).gs_oracle_last_sort_seconds())


def set_exact_dscale(on):
    """dL/dscale convention of both builds: False (default) = as the dependency's backward returns it (no scale_modifier factor),
    True = exact derivative.  -> previous setting"""
    old = False
    for dt in (np.float32, np.float64):
        old = bool(_lib(dt).gs_oracle_set_exact_dscale(1 if on else 0))
    return old


def _arr(x, dtype, shape=None):
    if x is None:
        return None
    a = np.ascontiguousarray(np.asarray(x, dtype=dtype))
    if shape is not None:
        a = a.reshape(shape)
    return a


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


class State:
    def __init__(self, lib, handle, inputs, settings, dtype):
        self.lib, self.handle, self.inputs, self.settings, self.dtype = lib, handle, inputs, settings, dtype

    def _view(self, name, ctype, count):
        ptr = getattr(self.lib, "gs_oracle_" + name)(self.handle)
        if count == 0:
            return np.zeros(0, dtype=ctype)
        buf = (C.c_char * (count * np.dtype(ctype).itemsize)).from_address(ptr)
        return np.frombuffer(buf, dtype=ctype, count=count).copy()

    @property
    def num_rendered(self):
        return int(self.lib.gs_oracle_num_rendered(self.handle))

    def geometry(self):
        N = self.inputs["N"]
        return {
            "xy": self._view("xy", self.dtype, 2 * N).reshape(N, 2),
            "depths": self._view("depths", self.dtype, N),
            "conic_opacity": self._view("conic_opacity", self.dtype, 4 * N).reshape(N, 4),
            "rgb": self._view("rgb", self.dtype, 3 * N).reshape(N, 3),
            "tiles_touched": self._view("tiles_touched", np.int32, N),
        }

    def binning(self):
        s = self.settings
        tiles = ((s["image_width"] + 15) // 16) * ((s["image_height"] + 15) // 16)
        return {
            "point_list": self._view("point_list", np.uint32, self.num_rendered),
            "ranges": self._view("ranges", np.uint32, 2 * tiles).reshape(tiles, 2),
        }

    def image_state(self):
        s = self.settings
        P = s["image_width"] * s["image_height"]
        return {"final_T": self._view("final_T", self.dtype, P), "n_contrib": self._view("n_contrib", np.uint32, P)}

    def __del__(self):
        if self.handle:
            self.lib.gs_oracle_free(self.handle)
            self.handle = None


def forward(means3D, opacities, settings, shs=None, colors_precomp=None, scales=None, rotations=None,
            cov3D_precomp=None, dtype=np.float32, nthreads=1):
    """-> (color[3,H,W], radii[N] i32, depth[1,H,W], alpha[1,H,W], State)"""
    if (shs is None) == (colors_precomp is None):
        raise Exception('Please provide excatly one of either SHs or precomputed colors!')
    if ((scales is None or rotations is None) and cov3D_precomp is None) or \
            ((scales is not None or rotations is not None) and cov3D_precomp is not None):
        raise Exception('Please provide exactly one of either scale/rotation pair or precomputed 3D covariance!')
    lib = _lib(dtype)
    real = C.c_float if np.dtype(dtype) == np.float32 else C.c_double
    means3D = _arr(means3D, dtype)
    N = means3D.shape[0] if means3D.ndim == 2 else 0
    H, W = int(settings["image_height"]), int(settings["image_width"])
    shs = _arr(shs, dtype)
    M = shs.shape[1] if shs is not None and shs.ndim == 3 else 0
    inp = dict(N=N, means3D=means3D, shs=shs, colors_precomp=_arr(colors_precomp, dtype),
               opacities=_arr(opacities, dtype), scales=_arr(scales, dtype), rotations=_arr(rotations, dtype),
               cov3D_precomp=_arr(cov3D_precomp, dtype))
    bg = _arr(settings["bg"], dtype, (3,))
    view = _arr(settings["viewmatrix"], dtype, (16,))
    proj = _arr(settings["projmatrix"], dtype, (16,))
    campos = _arr(settings["campos"], dtype, (3,))
    color = np.zeros((3, H, W), dtype)
    depth = np.zeros((1, H, W), dtype)
    alpha = np.zeros((1, H, W), dtype)
    radii = np.zeros((N,), np.int32)
    h = lib.gs_oracle_forward(N, M, int(settings["sh_degree"]), W, H, real(settings["tanfovx"]),
                              real(settings["tanfovy"]), real(settings.get("scale_modifier", 1.0)),
                              _p(bg), _p(view), _p(proj), _p(campos), _p(inp["means3D"]), _p(inp["shs"]),
                              _p(inp["colors_precomp"]), _p(inp["opacities"]), _p(inp["scales"]),
                              _p(inp["rotations"]), _p(inp["cov3D_precomp"]), int(settings.get("prefiltered", False)),
                              _p(color), _p(depth), _p(alpha), _p(radii), int(nthreads))
    st = State(lib, h, inp, dict(settings, image_height=H, image_width=W), np.dtype(dtype))
    return color, radii, depth, alpha, st


def backward(st, dL_dcolor, dL_ddepth=None, dL_dalpha=None, nthreads=1):
    """-> dict of gradients with the names the dependency's backward returns."""
    lib, dtype, inp = st.lib, st.dtype, st.inputs
    N = inp["N"]
    M = inp["shs"].shape[1] if inp["shs"] is not None else 0
    H, W = st.settings["image_height"], st.settings["image_width"]
    dL_dcolor = _arr(dL_dcolor, dtype, (3, H, W))
    dL_ddepth = _arr(dL_ddepth, dtype, (H, W)) if dL_ddepth is not None else np.zeros((H, W), dtype)
    dL_dalpha = _arr(dL_dalpha, dtype, (H, W)) if dL_dalpha is not None else np.zeros((H, W), dtype)
    g = {
        "means2D": np.zeros((N, 3), dtype), "conic": np.zeros((N, 4), dtype), "opacities": np.zeros((N, 1), dtype),
        "colors": np.zeros((N, 3), dtype), "depths": np.zeros((N,), dtype), "means3D": np.zeros((N, 3), dtype),
        "cov3D": np.zeros((N, 6), dtype), "shs": np.zeros((N, max(M, 1), 3), dtype)[:, :M],
        "scales": np.zeros((N, 3), dtype), "rotations": np.zeros((N, 4), dtype),
    }
    g["shs"] = np.ascontiguousarray(g["shs"])
    lib.gs_oracle_backward(st.handle, _p(inp["means3D"]), _p(inp["shs"]), _p(inp["colors_precomp"]),
                           _p(inp["scales"]), _p(inp["rotations"]), _p(inp["cov3D_precomp"]),
                           _p(dL_dcolor), _p(dL_ddepth), _p(dL_dalpha),
                           _p(g["means2D"]), _p(g["conic"]), _p(g["opacities"]), _p(g["colors"]), _p(g["depths"]),
                           _p(g["means3D"]), _p(g["cov3D"]), _p(g["shs"]), _p(g["scales"]), _p(g["rotations"]), int(nthreads))
    return g


def taint(st, pixel_flags, gauss_flags=None):
    """pixel_flags [H,W] bool -> gauss_flags [N] bool, OR-ed into `gauss_flags` when given: the Gaussians this view's composite loop can blend into a
    flagged pixel (gs_oracle.c: gs_oracle_taint).  Used by the full-size parity tests to attribute the float32-vs-float64 gradient tail."""
    H, W = st.settings["image_height"], st.settings["image_width"]
    pf = np.ascontiguousarray(np.asarray(pixel_flags).reshape(H, W).astype(np.uint8))
    gf = np.zeros((st.inputs["N"],), np.uint8) if gauss_flags is None else np.ascontiguousarray(gauss_flags.astype(np.uint8))
    st.lib.gs_oracle_taint(st.handle, _p(pf), _p(gf))
    return gf.astype(bool)


def mark_visible(means3D, viewmatrix, projmatrix, dtype=np.float32):
    lib = _lib(dtype)
    m = _arr(means3D, dtype)
    out = np.zeros((m.shape[0],), np.uint8)
    lib.gs_oracle_mark_visible(m.shape[0], _p(m), _p(_arr(viewmatrix, dtype, (16,))), _p(_arr(projmatrix, dtype, (16,))), _p(out))
    return out.astype(bool)
