"""ctypes front-end of oracle/mesh_oracle.c (TEST INFRASTRUCTURE; PARITY UNPINNED except rasterize/interpolate forward -- see that file's header).
Function names and argument meaning follow `nvdiffrast.torch` as the reference calls it
(/root/reference/MVs_Algorithms/DiffRastMesh/diff_mesh_renderer.py:97-138)."""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIBS = {}


def build(quiet=True):
    subprocess.run(["make", "-C", _HERE, "mesh"], check=True, stdout=subprocess.DEVNULL if quiet else None)


def _lib(dtype):
    dtype = np.dtype(dtype)
    if dtype not in _LIBS:
        path = os.path.join(_HERE, "_build", "libmesh_oracle_f32.so" if dtype == np.float32 else "libmesh_oracle_f64.so")
        if not os.path.exists(path):
            build()
        lib = C.CDLL(path)
        assert lib.mesh_oracle_sizeof_real() == dtype.itemsize
        _LIBS[dtype] = lib
    return _LIBS[dtype]


def _a(x, dt):
    return np.ascontiguousarray(np.asarray(x, dtype=dt))


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


I = C.c_int


def rasterize(pos, tri, resolution, dtype=np.float32):
    """pos [B,V,4] clip space, tri [T,3] int32, resolution (H,W) -> rast [B,H,W,4], rast_db [B,H,W,4]"""
    lib = _lib(dtype)
    pos, tri = _a(pos, dtype), _a(tri, np.int32)
    B, V, _ = pos.shape
    H, W = resolution
    rast = np.zeros((B, H, W, 4), dtype); db = np.zeros((B, H, W, 4), dtype)
    lib.mesh_rasterize_fwd(_p(pos), _p(tri), I(B), I(V), I(tri.shape[0]), I(H), I(W), _p(rast), _p(db))
    return rast, db


def rasterize_ranges(pos, tri, resolution, ranges, dtype=np.float32):
    """range ("instanced") mode as the dependency documents it: pos [V,4] shared; item b draws tri[start_b : start_b + count_b];
    ids index the full `tri`"""
    pos, tri = _a(pos, dtype), _a(tri, np.int32)
    rasts, dbs = [], []
    for start, count in np.asarray(ranges).tolist():
        r, d = rasterize(pos[None], tri[start:start + count], resolution, dtype=dtype)
        r[..., 3] += (r[..., 3] > 0) * start
        rasts.append(r); dbs.append(d)
    return np.concatenate(rasts, 0), np.concatenate(dbs, 0)


def rasterize_next_layer(pos, tri, resolution, prev_rast, dtype=np.float32):
    """one depth-peeling step: prev_rast = the previous layer's rast (None: first layer = rasterize)"""
    lib = _lib(dtype)
    pos, tri = _a(pos, dtype), _a(tri, np.int32)
    B, V, _ = pos.shape
    H, W = resolution
    rast = np.zeros((B, H, W, 4), dtype); db = np.zeros((B, H, W, 4), dtype)
    prev = None if prev_rast is None else _a(prev_rast, dtype)
    lib.mesh_rasterize_peel_fwd(_p(pos), _p(tri), I(B), I(V), I(tri.shape[0]), I(H), I(W), _p(prev), _p(rast), _p(db))
    return rast, db


def rasterize_bwd(pos, tri, rast, dy, ddb=None, dtype=np.float32):
    """ddb: gradient w.r.t. rast_db (optional; the dependency's grad_db path)"""
    lib = _lib(dtype)
    pos, tri, rast, dy = _a(pos, dtype), _a(tri, np.int32), _a(rast, dtype), _a(dy, dtype)
    ddb = None if ddb is None else _a(ddb, dtype)
    B, V, _ = pos.shape
    _, H, W, _ = rast.shape
    dpos = np.zeros_like(pos)
    lib.mesh_rasterize_bwd(_p(pos), _p(tri), _p(rast), _p(dy), _p(ddb) if ddb is not None else None, I(B), I(V), I(tri.shape[0]), I(H), I(W), _p(dpos))
    return dpos


def interpolate(attr, rast, tri, rast_db=None, diff_attrs=None, dtype=np.float32):
    """attr [Ba,V,A] -> out [B,H,W,A], out_da [B,H,W,2*nd] (empty when diff_attrs is None)"""
    lib = _lib(dtype)
    attr, rast, tri = _a(attr, dtype), _a(rast, dtype), _a(tri, np.int32)
    if attr.ndim == 2:
        attr = attr[None]
    Ba, V, A = attr.shape
    B, H, W, _ = rast.shape
    if diff_attrs is None:
        diff = np.zeros((0,), np.int32)
    elif isinstance(diff_attrs, str):
        assert diff_attrs == "all"
        diff = np.arange(A, dtype=np.int32)
    else:
        diff = _a(diff_attrs, np.int32)
    nd = diff.shape[0]
    out = np.zeros((B, H, W, A), dtype); out_da = np.zeros((B, H, W, 2 * nd), dtype)
    db = _a(rast_db, dtype) if nd else None
    lib.mesh_interpolate_fwd(_p(attr), I(Ba), _p(rast), _p(tri), _p(db), _p(diff), I(nd), I(B), I(V), I(A), I(H), I(W), _p(out), _p(out_da))
    return out, out_da


def interpolate_bwd(attr, rast, tri, dy, dtype=np.float32):
    lib = _lib(dtype)
    attr, rast, tri, dy = _a(attr, dtype), _a(rast, dtype), _a(tri, np.int32), _a(dy, dtype)
    squeeze = attr.ndim == 2
    if squeeze:
        attr = attr[None]
    Ba, V, A = attr.shape
    B, H, W, _ = rast.shape
    dattr = np.zeros_like(attr); drast = np.zeros_like(rast)
    lib.mesh_interpolate_bwd(_p(attr), I(Ba), _p(rast), _p(tri), _p(dy), I(B), I(V), I(A), I(H), I(W), _p(dattr), _p(drast))
    return (dattr[0] if squeeze else dattr), drast


def interpolate_da_bwd(attr, rast, tri, rast_db, diff_attrs, dout_da, dtype=np.float32):
    """backward of interpolate()'s pixel differentials -> (dattr, drast_db)"""
    lib = _lib(dtype)
    attr, rast, tri, db, g = _a(attr, dtype), _a(rast, dtype), _a(tri, np.int32), _a(rast_db, dtype), _a(dout_da, dtype)
    squeeze = attr.ndim == 2
    if squeeze:
        attr = attr[None]
    Ba, V, A = attr.shape
    B, H, W, _ = rast.shape
    diff = np.arange(A, dtype=np.int32) if (isinstance(diff_attrs, str) and diff_attrs == "all") else _a(diff_attrs, np.int32)
    dattr = np.zeros_like(attr); ddb = np.zeros_like(db)
    lib.mesh_interpolate_da_bwd(_p(attr), I(Ba), _p(rast), _p(tri), _p(db), _p(diff), I(diff.shape[0]), _p(g), I(B), I(V), I(A), I(H), I(W), _p(dattr), _p(ddb))
    return (dattr[0] if squeeze else dattr), ddb


_FILTER = {"nearest": 0, "linear": 1}
_BOUNDARY = {"wrap": 0, "clamp": 1, "zero": 2}


def texture(tex, uv, filter_mode="linear", boundary_mode="wrap", dtype=np.float32):
    lib = _lib(dtype)
    tex, uv = _a(tex, dtype), _a(uv, dtype)
    Bt, Ht, Wt, Cc = tex.shape
    B, H, W, _ = uv.shape
    out = np.zeros((B, H, W, Cc), dtype)
    lib.mesh_texture_fwd(_p(tex), I(Bt), _p(uv), I(B), I(H), I(W), I(Ht), I(Wt), I(Cc), I(_FILTER[filter_mode]), I(_BOUNDARY[boundary_mode]), _p(out))
    return out


def texture_bwd(tex, uv, dy, filter_mode="linear", boundary_mode="wrap", dtype=np.float32):
    lib = _lib(dtype)
    tex, uv, dy = _a(tex, dtype), _a(uv, dtype), _a(dy, dtype)
    Bt, Ht, Wt, Cc = tex.shape
    B, H, W, _ = uv.shape
    dtex = np.zeros_like(tex); duv = np.zeros_like(uv)
    lib.mesh_texture_bwd(_p(tex), I(Bt), _p(uv), _p(dy), I(B), I(H), I(W), I(Ht), I(Wt), I(Cc), I(_FILTER[filter_mode]), I(_BOUNDARY[boundary_mode]),
                         _p(dtex), _p(duv))
    return dtex, duv


_MIP_FILTER = {"linear-mipmap-nearest": 2, "linear-mipmap-linear": 3}
LL = C.c_longlong


def mip_info(Ht, Wt, max_mip_level=None):
    """-> [(h, w)] per level including the base, texels in levels 1..L.  Raises ValueError where an odd extent > 1 would have to be halved."""
    lib = _lib(np.float32)
    hw = (C.c_int * 34)(); tot = LL(0)
    L = lib.mesh_mip_info(I(Ht), I(Wt), I(-1 if max_mip_level is None else max_mip_level), hw, C.byref(tot))
    if L < 0:
        raise ValueError("mip pyramid: odd extent > 1 in %dx%d (limit the depth with max_mip_level)" % (Ht, Wt))
    return [(hw[2 * l], hw[2 * l + 1]) for l in range(L + 1)], int(tot.value)


def mip_build(tex, max_mip_level=None, dtype=np.float32):
    """tex [Bt,Ht,Wt,C] -> stack [Bt, texels of levels 1..L, C]"""
    lib = _lib(dtype)
    tex = _a(tex, dtype)
    Bt, Ht, Wt, Cc = tex.shape
    _, tot = mip_info(Ht, Wt, max_mip_level)
    stack = np.zeros((Bt, tot, Cc), dtype)
    assert lib.mesh_mip_build(_p(tex), I(Bt), I(Ht), I(Wt), I(Cc), I(-1 if max_mip_level is None else max_mip_level), _p(stack)) == 0
    return stack


def mip_build_bwd(dstack, tex_shape, max_mip_level=None, dtype=np.float32):
    lib = _lib(dtype)
    dstack = _a(dstack, dtype).copy()
    Bt, Ht, Wt, Cc = tex_shape
    dtex = np.zeros(tex_shape, dtype)
    assert lib.mesh_mip_build_bwd(_p(dstack), I(Bt), I(Ht), I(Wt), I(Cc), I(-1 if max_mip_level is None else max_mip_level), _p(dtex)) == 0
    return dtex


def texture_mip(tex, uv, uv_da=None, mip_level_bias=None, stack=None, filter_mode="linear-mipmap-linear", boundary_mode="wrap", max_mip_level=None,
                dtype=np.float32):
    lib = _lib(dtype)
    tex, uv = _a(tex, dtype), _a(uv, dtype)
    Bt, Ht, Wt, Cc = tex.shape
    B, H, W, _ = uv.shape
    stack = mip_build(tex, max_mip_level, dtype) if stack is None else _a(stack, dtype)
    da = None if uv_da is None else _a(uv_da, dtype)
    bias = None if mip_level_bias is None else _a(mip_level_bias, dtype)
    out = np.zeros((B, H, W, Cc), dtype)
    assert lib.mesh_texture_mip_fwd(_p(tex), _p(stack), I(Bt), _p(uv), _p(da), _p(bias), I(B), I(H), I(W), I(Ht), I(Wt), I(Cc), I(_MIP_FILTER[filter_mode]),
                                    I(_BOUNDARY[boundary_mode]), I(-1 if max_mip_level is None else max_mip_level), _p(out)) == 0
    return out


def texture_mip_bwd(tex, uv, dy, uv_da=None, mip_level_bias=None, stack=None, filter_mode="linear-mipmap-linear", boundary_mode="wrap",
                    max_mip_level=None, dtype=np.float32, level_grads=False):
    """-> dtex (level-0 taps only), dstack (levels >= 1), duv [, d uv_da, d mip_level_bias with level_grads=True].  The gradient of the base texture
    of an internally built pyramid is dtex + mip_build_bwd(dstack)."""
    lib = _lib(dtype)
    tex, uv, dy = _a(tex, dtype), _a(uv, dtype), _a(dy, dtype)
    Bt, Ht, Wt, Cc = tex.shape
    B, H, W, _ = uv.shape
    stack = mip_build(tex, max_mip_level, dtype) if stack is None else _a(stack, dtype)
    da = None if uv_da is None else _a(uv_da, dtype)
    bias = None if mip_level_bias is None else _a(mip_level_bias, dtype)
    dtex = np.zeros_like(tex); dstack = np.zeros_like(stack); duv = np.zeros_like(uv)
    dda = np.zeros((B, H, W, 4), dtype) if level_grads else None
    dbias = np.zeros((B, H, W), dtype) if level_grads else None
    assert lib.mesh_texture_mip_bwd(_p(tex), _p(stack), I(Bt), _p(uv), _p(da), _p(bias), _p(dy), I(B), I(H), I(W), I(Ht), I(Wt), I(Cc),
                                    I(_MIP_FILTER[filter_mode]), I(_BOUNDARY[boundary_mode]), I(-1 if max_mip_level is None else max_mip_level),
                                    _p(dtex), _p(dstack), _p(duv), _p(dda), _p(dbias)) == 0
    return (dtex, dstack, duv, dda, dbias) if level_grads else (dtex, dstack, duv)


def antialias(color, rast, pos, tri, dtype=np.float32):
    lib = _lib(dtype)
    color, rast, pos, tri = _a(color, dtype), _a(rast, dtype), _a(pos, dtype), _a(tri, np.int32)
    B, H, W, Cc = color.shape
    out = np.zeros_like(color)
    lib.mesh_antialias_fwd(_p(color), _p(rast), _p(pos), _p(tri), I(B), I(pos.shape[1]), I(tri.shape[0]), I(H), I(W), I(Cc), _p(out))
    return out


def antialias_bwd(color, rast, pos, tri, dy, dtype=np.float32):
    lib = _lib(dtype)
    color, rast, pos, tri, dy = _a(color, dtype), _a(rast, dtype), _a(pos, dtype), _a(tri, np.int32), _a(dy, dtype)
    B, H, W, Cc = color.shape
    dcolor = np.zeros_like(color); dpos = np.zeros_like(pos)
    lib.mesh_antialias_bwd(_p(color), _p(rast), _p(pos), _p(tri), _p(dy), I(B), I(pos.shape[1]), I(tri.shape[0]), I(H), I(W), I(Cc), _p(dcolor), _p(dpos))
    return dcolor, dpos
