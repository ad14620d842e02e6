"""`nvdiffrast.torch` -- drop-in Python boundary over the MI355X HIP mesh ops (include/c3d_mesh.h).

Same function names, argument meaning and return shapes as the module the reference imports as `dr`
(/root/reference/MVs_Algorithms/DiffRastMesh/diff_mesh_renderer.py:8, call sites :46,97,101,104,105,110,131,138;
 also MVs_Algorithms/FlexiCubes/flexicubes_renderer.py:46-66 and mesh_processer/mesh_utils.py:527-541):
    RasterizeCudaContext(device=None) / RasterizeGLContext(...)      rasterize(glctx, pos, tri, resolution, ranges=None, grad_db=True)
    interpolate(attr, rast, tri, rast_db=None, diff_attrs=None)      texture(tex, uv, uv_da=None, ..., filter_mode='auto', boundary_mode='wrap')
    antialias(color, rast, pos, tri, topology_hash=None, pos_gradient_boost=1.0)
    texture_construct_mip(tex, max_mip_level=None)                    filter modes nearest | linear | linear-mipmap-nearest | linear-mipmap-linear
    DepthPeeler(glctx, pos, tri, resolution).rasterize_next_layer()
The arithmetic runs in libc3d_hip.so; this file allocates tensors and wires autograd.  There is no CPU path.
    rasterize(..., ranges=[B,2]) / antialias with pos [V,4]: range (instanced) mode, composed on the host from the B = 1 kernels
    boundary modes 'wrap' | 'clamp' | 'zero' ('zero' for the nearest / linear filters, composed from a zero-padded texture + 'clamp')
Not built (raise NotImplementedError): cube maps (boundary 'cube').  Gradients w.r.t. uv_da and
mip_level_bias ('linear-mipmap-linear') and through interpolate's pixel differentials out_da to the attributes and to rast_db are propagated, and
rasterize(grad_db=True, the default) passes what arrives at rast_db on to the positions.
"""
import torch

import c3d_hip as _h

_FILTER = {"nearest": 0, "linear": 1}
_BOUNDARY = {"wrap": 0, "clamp": 1}


def _dev_check(t, what):
    if not t.is_cuda:
        raise RuntimeError("nvdiffrast.torch (MI355X): %s must live on a HIP device; there is no CPU path" % what)


ATOMIC_FREE_BACKWARD = True     # False: the scatter-add (float atomics) formulation of rasterize backward, kept for comparison


class RasterizeCudaContext:
    """Owns the rasterizer's scratch (64-bit depth|id buffer + large-triangle queue), grown on demand."""

    def __init__(self, device=None):
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        self._scratch = None
        self.active_depth_peeler = None

    def scratch(self, nbytes, device):
        if self._scratch is None or self._scratch.numel() < nbytes or self._scratch.device != device:
            self._scratch = torch.empty((nbytes,), dtype=torch.uint8, device=device)
        return self._scratch

    def vertex_topology(self, tri, V):
        """vertex -> (triangle, corner) adjacency of `tri` for the atomic-free backward; built once per index buffer"""
        key = (tri.data_ptr(), tri._version, int(tri.shape[0]), int(V), tri.device)
        if getattr(self, "_vtopo_key", None) != key:
            lib = _h.lib()
            T = int(tri.shape[0])
            buf = torch.empty((lib.c3d_mesh_vertex_topology_bytes(V, T),), dtype=torch.uint8, device=tri.device)
            with torch.cuda.device(tri.device):
                _h.check(lib.c3d_mesh_build_vertex_topology(_h.ptr(tri if T else None), V, T, _h.ptr(buf), _h.stream(tri.device)), "c3d_mesh_build_vertex_topology")
            self._vtopo, self._vtopo_key, self._vtopo_tri = buf, key, tri      # keep `tri` alive: the key holds its address
        return self._vtopo


class RasterizeGLContext(RasterizeCudaContext):
    """The reference picks the GL context on Windows unless force_cuda_rast (diff_mesh_renderer.py:45-48): same engine here."""

    def __init__(self, output_db=True, mode='automatic', device=None):
        super().__init__(device)


class _Rasterize(torch.autograd.Function):
    @staticmethod
    def forward(ctx, glctx, pos, tri, resolution, grad_db, peeler=None):
        lib = _h.lib()
        _dev_check(pos, "pos")
        if pos.dim() != 3 or pos.shape[-1] != 4:
            raise ValueError("rasterize: pos must be [B,V,4] (pass ranges=... for one shared vertex buffer [V,4])")
        pos_c, tri_c = _h.f32c(pos), tri.to(torch.int32).contiguous()
        B, V = pos.shape[0], pos.shape[1]
        T = tri_c.shape[0]
        H, W = int(resolution[0]), int(resolution[1])
        dev = pos.device
        with torch.cuda.device(dev):
            nbytes = lib.c3d_mesh_raster_scratch_bytes(B, H, W, T)
            rast = torch.empty((B, H, W, 4), dtype=torch.float32, device=dev)
            rast_db = torch.empty((B, H, W, 4), dtype=torch.float32, device=dev)
            if peeler is None:
                scratch = glctx.scratch(nbytes, dev)
                _h.check(lib.c3d_mesh_rasterize_fwd(_h.ptr(pos_c), _h.ptr(tri_c if T else None), B, V, T, H, W, _h.ptr(scratch), _h.ptr(rast),
                                                    _h.ptr(rast_db), _h.stream(dev)), "c3d_mesh_rasterize_fwd")
            else:
                prev, scratch = peeler._next_buffers(nbytes, dev)       # the previous layer's depth|id words, and where this layer's go
                _h.check(lib.c3d_mesh_rasterize_peel_fwd(_h.ptr(pos_c), _h.ptr(tri_c if T else None), B, V, T, H, W, _h.ptr(prev), _h.ptr(scratch),
                                                         _h.ptr(rast), _h.ptr(rast_db), _h.stream(dev)), "c3d_mesh_rasterize_peel_fwd")
        ctx.save_for_backward(pos_c if pos_c is not None else pos, tri_c, rast)
        ctx.dims = (B, V, T, H, W)
        ctx.glctx = glctx
        ctx.grad_db = bool(grad_db)
        return rast, rast_db

    @staticmethod
    def backward(ctx, dy, ddb):
        lib = _h.lib()
        pos, tri, rast = ctx.saved_tensors
        B, V, T, H, W = ctx.dims
        dev = pos.device
        # grad_db (the dependency's default): what arrives at rast_db -- interpolate's backward for its pixel differentials -- reaches the positions too
        dy_c = _h.f32c(dy) if dy is not None else None
        ddb_c = _h.f32c(ddb) if (ddb is not None and ctx.grad_db) else None
        with torch.cuda.device(dev):
            dpos = torch.empty((B, V, 4), dtype=torch.float32, device=dev)
            if T == 0 or (dy_c is None and ddb_c is None):
                dpos.zero_()                                          # nothing was drawn (an empty range) / nothing arrived
            elif ATOMIC_FREE_BACKWARD:
                # gather formulation: per-triangle corner records, then a fixed-order sum per vertex (no atomics, bit-reproducible)
                topo = ctx.glctx.vertex_topology(tri, V)
                scratch = torch.empty((lib.c3d_mesh_rasterize_bwd_scratch_bytes(B, T),), dtype=torch.uint8, device=dev)
                _h.check(lib.c3d_mesh_rasterize_bwd_gather(_h.ptr(pos), _h.ptr(tri), _h.ptr(rast), _h.ptr(dy_c), _h.ptr(ddb_c), B, V, T, H, W, _h.ptr(topo),
                                                           _h.ptr(scratch), _h.ptr(dpos), _h.stream(dev)), "c3d_mesh_rasterize_bwd_gather")
            else:
                _h.check(lib.c3d_mesh_rasterize_bwd(_h.ptr(pos), _h.ptr(tri if T else None), _h.ptr(rast), _h.ptr(dy_c), _h.ptr(ddb_c), B, V, T, H, W,
                                                    _h.ptr(dpos), _h.stream(dev)), "c3d_mesh_rasterize_bwd")
        return None, dpos, None, None, None, None


def rasterize(glctx, pos, tri, resolution, ranges=None, grad_db=True):
    """-> (rast [B,H,W,4] = (u, v, z/w, triangle_id+1), rast_db [B,H,W,4] = (du/dX, du/dY, dv/dX, dv/dY))"""
    if not isinstance(glctx, RasterizeCudaContext):
        raise TypeError("rasterize: glctx must be a RasterizeCudaContext / RasterizeGLContext")
    if glctx.active_depth_peeler is not None:
        raise RuntimeError("rasterize: cannot be called on a context while a DepthPeeler is active on it")
    if ranges is not None:
        return _rasterize_ranges(glctx, pos, tri, resolution, ranges, grad_db)
    return _Rasterize.apply(glctx, pos, tri, resolution, grad_db, None)


def _rasterize_ranges(glctx, pos, tri, resolution, ranges, grad_db):
    """Range ("instanced") mode: one vertex buffer pos [V,4]; minibatch item b draws triangles tri[start_b : start_b + count_b]
    (ranges [B,2] int32 on the CPU); triangle ids in the output index the full `tri`.  Composed on the host from the B = 1 kernels:
    the items share nothing but the vertex buffer, whose gradient autograd sums over the items."""
    if pos.dim() != 2 or pos.shape[-1] != 4:
        raise ValueError("rasterize: with ranges, pos must be [V,4]")
    r = torch.as_tensor(ranges)
    if r.is_cuda or r.dim() != 2 or r.shape[1] != 2 or r.dtype not in (torch.int32, torch.int64):
        raise ValueError("rasterize: ranges must be an integer CPU tensor of shape [B,2]")
    tri_c = tri.to(torch.int32).contiguous()
    T = int(tri_c.shape[0])
    rasts, dbs = [], []
    for start, count in r.tolist():
        if start < 0 or count < 0 or start + count > T:
            raise ValueError("rasterize: range (%d, %d) does not fit the %d triangles" % (start, count, T))
        rast, db = _Rasterize.apply(glctx, pos.unsqueeze(0), tri_c[start:start + count], resolution, grad_db, None)
        if start:
            shift = torch.zeros_like(rast)
            shift[..., 3] = (rast[..., 3] > 0).to(rast.dtype) * float(start)
            rast = rast + shift
        rasts.append(rast); dbs.append(db)
    if not rasts:
        H, W = int(resolution[0]), int(resolution[1])
        z = pos.new_zeros((0, H, W, 4), dtype=torch.float32)
        return z, z.clone()
    return torch.cat(rasts, 0), torch.cat(dbs, 0)


class DepthPeeler:
    """`with DepthPeeler(glctx, pos, tri, resolution) as peeler: rast, db = peeler.rasterize_next_layer()` -- layer 0 is what rasterize()
    returns; every further layer keeps, per pixel, the nearest surface strictly behind the previous layer's (empty where that layer was
    empty).  Layers are differentiable like rasterize() outputs.  (Reference use: InstantMesh/models/geometry/render/neural_render.py:103.)"""

    def __init__(self, glctx, pos, tri, resolution, ranges=None, grad_db=True):
        if not isinstance(glctx, RasterizeCudaContext):
            raise TypeError("DepthPeeler: glctx must be a RasterizeCudaContext / RasterizeGLContext")
        self.raster_ctx, self.pos, self.tri, self.resolution, self.grad_db = glctx, pos, tri, resolution, grad_db
        self._bufs, self._layer, self._active = [None, None], 0, False
        # range ("instanced") mode, as rasterize(..., ranges=): one vertex buffer pos [V,4], item b peels triangles tri[start_b : start_b + count_b].
        # Composed on the host from B = 1 peelers that share nothing but the vertex buffer (whose gradient autograd sums over the items).
        self._items = None
        if ranges is not None:
            if pos.dim() != 2 or pos.shape[-1] != 4:
                raise ValueError("DepthPeeler: with ranges, pos must be [V,4]")
            r = torch.as_tensor(ranges)
            if r.is_cuda or r.dim() != 2 or r.shape[1] != 2 or r.dtype not in (torch.int32, torch.int64):
                raise ValueError("DepthPeeler: ranges must be an integer CPU tensor of shape [B,2]")
            tri_c = tri.to(torch.int32).contiguous()
            T = int(tri_c.shape[0])
            self._items = []
            for start, count in r.tolist():
                if start < 0 or count < 0 or start + count > T:
                    raise ValueError("DepthPeeler: range (%d, %d) does not fit the %d triangles" % (start, count, T))
                sub = DepthPeeler.__new__(DepthPeeler)
                sub.raster_ctx, sub.pos, sub.tri, sub.resolution, sub.grad_db = glctx, pos.unsqueeze(0), tri_c[start:start + count], resolution, grad_db
                sub._bufs, sub._layer, sub._active, sub._items = [None, None], 0, False, None
                self._items.append((int(start), sub))

    def __enter__(self):
        if self.raster_ctx.active_depth_peeler is not None:
            raise RuntimeError("DepthPeeler: another depth peeling operation is active on this context")
        self.raster_ctx.active_depth_peeler = self
        self._layer, self._active = 0, True
        for _, sub in (self._items or []):
            sub._layer, sub._active = 0, True
        return self

    def __exit__(self, *args):
        self.raster_ctx.active_depth_peeler = None
        self._bufs, self._active = [None, None], False
        for _, sub in (self._items or []):
            sub._bufs, sub._active = [None, None], False
        return False

    def _next_buffers(self, nbytes, device):
        cur = self._layer & 1
        if self._bufs[cur] is None or self._bufs[cur].numel() < nbytes or self._bufs[cur].device != device:
            self._bufs[cur] = torch.empty((nbytes,), dtype=torch.uint8, device=device)
        prev = self._bufs[cur ^ 1] if self._layer > 0 else None
        self._layer += 1
        return prev, self._bufs[cur]

    def rasterize_next_layer(self):
        if not self._active:
            raise RuntimeError("DepthPeeler: rasterize_next_layer() must be called inside the `with` block")
        if self._items is None:
            return _Rasterize.apply(self.raster_ctx, self.pos, self.tri, self.resolution, self.grad_db, self)
        rasts, dbs = [], []
        for start, sub in self._items:          # every item peels its own triangle range; ids in the output index the full `tri`
            rast, db = _Rasterize.apply(sub.raster_ctx, sub.pos, sub.tri, sub.resolution, sub.grad_db, sub)
            if start:
                shift = torch.zeros_like(rast)
                shift[..., 3] = (rast[..., 3] > 0).to(rast.dtype) * float(start)
                rast = rast + shift
            rasts.append(rast); dbs.append(db)
        if not rasts:
            H, W = int(self.resolution[0]), int(self.resolution[1])
            z = self.pos.new_zeros((0, H, W, 4), dtype=torch.float32)
            return z, z.clone()
        return torch.cat(rasts, 0), torch.cat(dbs, 0)


class _Interpolate(torch.autograd.Function):
    @staticmethod
    def forward(ctx, attr, rast, tri, rast_db, diff):
        lib = _h.lib()
        _dev_check(attr, "attr")
        a = _h.f32c(attr)
        a3 = a if a.dim() == 3 else a.unsqueeze(0)
        Ba, V, A = a3.shape
        rast_c, tri_c = _h.f32c(rast), tri.to(torch.int32).contiguous()
        B, H, W, _ = rast_c.shape
        dev = rast_c.device
        nd = 0 if diff is None else int(diff.numel())
        with torch.cuda.device(dev):
            out = torch.empty((B, H, W, A), dtype=torch.float32, device=dev)
            out_da = torch.empty((B, H, W, 2 * nd), dtype=torch.float32, device=dev)
            _h.check(lib.c3d_mesh_interpolate_fwd(_h.ptr(a3), Ba, _h.ptr(rast_c), _h.ptr(tri_c), _h.ptr(_h.f32c(rast_db)) if nd else None,
                                                  _h.ptr(diff) if nd else None, nd, B, V, A, H, W, _h.ptr(out), _h.ptr(out_da) if nd else None,
                                                  _h.stream(dev)), "c3d_mesh_interpolate_fwd")
        ctx.save_for_backward(a3, rast_c, tri_c, _h.f32c(rast_db) if nd else None, diff if nd else None)
        ctx.attr_shape = tuple(attr.shape)
        ctx.nd = nd
        return out, out_da

    @staticmethod
    def backward(ctx, dy, dda):
        lib = _h.lib()
        a3, rast, tri, rast_db, diff = ctx.saved_tensors
        Ba, V, A = a3.shape
        B, H, W, _ = rast.shape
        dev = rast.device
        need_attr = ctx.needs_input_grad[0]      # constant attributes (texture coordinates) need no vertex scatter at all
        drast_db = None
        with torch.cuda.device(dev):
            dattr = torch.empty_like(a3) if need_attr else None
            drast = torch.empty_like(rast)
            _h.check(lib.c3d_mesh_interpolate_bwd(_h.ptr(a3), Ba, _h.ptr(rast), _h.ptr(tri), _h.ptr(_h.f32c(dy)), B, V, A, H, W,
                                                  _h.ptr(dattr) if need_attr else None, _h.ptr(drast), _h.stream(dev)), "c3d_mesh_interpolate_bwd")
            if ctx.nd and dda is not None and (need_attr or ctx.needs_input_grad[3]):
                # the pixel differentials out_da are linear in the attributes and in rast_db: what a mip-mapped texture() sends back for uv_da goes on to both
                drast_db = torch.empty_like(rast_db)
                _h.check(lib.c3d_mesh_interpolate_da_bwd(_h.ptr(a3), Ba, _h.ptr(rast), _h.ptr(tri), _h.ptr(rast_db), _h.ptr(diff), ctx.nd, _h.ptr(_h.f32c(dda)), B, V, A, H, W,
                                                         _h.ptr(dattr) if need_attr else None, _h.ptr(drast_db), _h.stream(dev)), "c3d_mesh_interpolate_da_bwd")
        return (dattr.reshape(ctx.attr_shape) if need_attr else None), drast, None, (drast_db if ctx.needs_input_grad[3] else None), None


def interpolate(attr, rast, tri, rast_db=None, diff_attrs=None):
    """-> (out [B,H,W,A], out_da [B,H,W,2*len(diff_attrs)])   (out_da empty when diff_attrs is None)"""
    diff = None
    if diff_attrs is not None:
        if rast_db is None:
            raise ValueError("interpolate: diff_attrs requires rast_db")
        A = attr.shape[-1]
        idx = list(range(A)) if (isinstance(diff_attrs, str) and diff_attrs == 'all') else list(diff_attrs)
        if len(idx):
            diff = torch.tensor(idx, dtype=torch.int32, device=rast.device)
    return _Interpolate.apply(attr, rast, tri, rast_db, diff)


class _Texture(torch.autograd.Function):
    @staticmethod
    def forward(ctx, tex, uv, filter_id, boundary_id):
        lib = _h.lib()
        _dev_check(tex, "tex")
        tex_c, uv_c = _h.f32c(tex), _h.f32c(uv)
        Bt, Ht, Wt, C = tex_c.shape
        B, H, W, _ = uv_c.shape
        dev = uv_c.device
        with torch.cuda.device(dev):
            out = torch.empty((B, H, W, C), dtype=torch.float32, device=dev)
            _h.check(lib.c3d_mesh_texture_fwd(_h.ptr(tex_c), Bt, _h.ptr(uv_c), B, H, W, Ht, Wt, C, filter_id, boundary_id, _h.ptr(out), _h.stream(dev)),
                     "c3d_mesh_texture_fwd")
        ctx.save_for_backward(tex_c, uv_c)
        ctx.modes = (filter_id, boundary_id)
        return out

    @staticmethod
    def backward(ctx, dy):
        lib = _h.lib()
        tex, uv = ctx.saved_tensors
        Bt, Ht, Wt, C = tex.shape
        B, H, W, _ = uv.shape
        dev = uv.device
        with torch.cuda.device(dev):
            dtex, duv = torch.empty_like(tex), torch.empty_like(uv)
            _h.check(lib.c3d_mesh_texture_bwd(_h.ptr(tex), Bt, _h.ptr(uv), _h.ptr(_h.f32c(dy)), B, H, W, Ht, Wt, C, ctx.modes[0], ctx.modes[1],
                                              _h.ptr(dtex), _h.ptr(duv), _h.stream(dev)), "c3d_mesh_texture_bwd")
        return dtex, duv, None, None


_MIP_FILTER = {"linear-mipmap-nearest": 2, "linear-mipmap-linear": 3}


def _mip_levels(Ht, Wt, max_mip_level):
    """[(h, w)] per level including the base, texels of levels 1..L.  ValueError where an odd extent > 1 would have to be halved."""
    import ctypes
    hw = (ctypes.c_int32 * 34)()
    tot = ctypes.c_int64(0)
    L = _h.lib().c3d_mesh_mip_info(Ht, Wt, -1 if max_mip_level is None else int(max_mip_level), ctypes.addressof(hw), ctypes.addressof(tot))
    if L < 0:
        msg = _h.lib().c3d_last_error()
        raise ValueError("texture mip pyramid for %dx%d: %s" % (Ht, Wt, msg.decode() if msg else "?"))
    return [(hw[2 * l], hw[2 * l + 1]) for l in range(L + 1)], int(tot.value)


class _MipBuild(torch.autograd.Function):
    """tex [Bt,Ht,Wt,C] -> stack [Bt, texels of levels 1..L, C] (2x2 box pyramid); backward = its transpose"""

    @staticmethod
    def forward(ctx, tex, max_level):
        lib = _h.lib()
        _dev_check(tex, "tex")
        tex_c = _h.f32c(tex)
        Bt, Ht, Wt, C = tex_c.shape
        _, total = _mip_levels(Ht, Wt, None if max_level < 0 else max_level)
        dev = tex_c.device
        with torch.cuda.device(dev):
            stack = torch.empty((Bt, total, C), dtype=torch.float32, device=dev)
            _h.check(lib.c3d_mesh_mip_build(_h.ptr(tex_c), Bt, Ht, Wt, C, max_level, _h.ptr(stack), _h.stream(dev)), "c3d_mesh_mip_build")
        ctx.shape, ctx.max_level = (Bt, Ht, Wt, C), max_level
        return stack

    @staticmethod
    def backward(ctx, dstack):
        lib = _h.lib()
        Bt, Ht, Wt, C = ctx.shape
        dev = dstack.device
        with torch.cuda.device(dev):
            scratch = _h.f32c(dstack).clone()                      # the fold runs in place, level by level
            dtex = torch.empty(ctx.shape, dtype=torch.float32, device=dev)
            _h.check(lib.c3d_mesh_mip_build_bwd(_h.ptr(scratch), Bt, Ht, Wt, C, ctx.max_level, _h.ptr(dtex), _h.stream(dev)), "c3d_mesh_mip_build_bwd")
        return dtex, None


class TextureMipWrapper:
    """What texture_construct_mip returns: the packed levels 1..L of `tex` (differentiable w.r.t. `tex`) and the shape they belong to."""

    def __init__(self, stack, shape, max_mip_level):
        self.stack, self.shape, self.max_mip_level = stack, tuple(shape), max_mip_level


def texture_construct_mip(tex, max_mip_level=None, cube_mode=False):
    """Pre-built mip stack for texture(..., mip=...), reusable across calls while `tex` is unchanged."""
    if cube_mode or tex.dim() != 4:
        raise NotImplementedError("texture_construct_mip: cube maps are not built")
    return TextureMipWrapper(_MipBuild.apply(tex, -1 if max_mip_level is None else int(max_mip_level)), tex.shape, max_mip_level)


class _TextureMip(torch.autograd.Function):
    @staticmethod
    def forward(ctx, tex, stack, uv, uv_da, bias, filter_id, boundary_id, max_level):
        lib = _h.lib()
        _dev_check(tex, "tex")
        tex_c, stack_c, uv_c = _h.f32c(tex), _h.f32c(stack), _h.f32c(uv)
        da_c = None if uv_da is None else _h.f32c(uv_da)
        bias_c = None if bias is None else _h.f32c(bias)
        Bt, Ht, Wt, C = tex_c.shape
        B, H, W, _ = uv_c.shape
        dev = uv_c.device
        with torch.cuda.device(dev):
            out = torch.empty((B, H, W, C), dtype=torch.float32, device=dev)
            _h.check(lib.c3d_mesh_texture_mip_fwd(_h.ptr(tex_c), _h.ptr(stack_c), Bt, _h.ptr(uv_c), _h.ptr(da_c), _h.ptr(bias_c),
                                                  B, H, W, Ht, Wt, C, filter_id, boundary_id, max_level, _h.ptr(out), _h.stream(dev)), "c3d_mesh_texture_mip_fwd")
        ctx.save_for_backward(tex_c, stack_c, uv_c, da_c, bias_c)
        ctx.modes = (filter_id, boundary_id, max_level)
        ctx.stack_shape = tuple(stack.shape)
        ctx.level_grads = (uv_da is not None and uv_da.requires_grad, bias is not None and bias.requires_grad)
        ctx.in_shapes = (None if uv_da is None else tuple(uv_da.shape), None if bias is None else tuple(bias.shape))
        return out

    @staticmethod
    def backward(ctx, dy):
        lib = _h.lib()
        tex, stack, uv, da, bias = ctx.saved_tensors
        Bt, Ht, Wt, C = tex.shape
        B, H, W, _ = uv.shape
        dev = uv.device
        with torch.cuda.device(dev):
            dtex, duv = torch.empty_like(tex), torch.empty_like(uv)
            dstack = torch.empty(ctx.stack_shape, dtype=torch.float32, device=dev)
            # the pixel differentials and the level bias steer the level: 'linear-mipmap-linear' has a gradient w.r.t. both (the dependency propagates it)
            dda = torch.empty((B, H, W, 4), dtype=torch.float32, device=dev) if ctx.level_grads[0] else None
            dbias = torch.empty((B, H, W), dtype=torch.float32, device=dev) if ctx.level_grads[1] else None
            _h.check(lib.c3d_mesh_texture_mip_bwd(_h.ptr(tex), _h.ptr(stack), Bt, _h.ptr(uv), _h.ptr(da), _h.ptr(bias), _h.ptr(_h.f32c(dy)),
                                                  B, H, W, Ht, Wt, C, ctx.modes[0], ctx.modes[1], ctx.modes[2], _h.ptr(dtex),
                                                  _h.ptr(dstack if dstack.numel() else None), _h.ptr(duv), _h.ptr(dda), _h.ptr(dbias), _h.stream(dev)), "c3d_mesh_texture_mip_bwd")
        if dda is not None:
            dda = dda.reshape(ctx.in_shapes[0])
        if dbias is not None:
            dbias = dbias.reshape(ctx.in_shapes[1])
        return dtex, dstack, duv, dda, dbias, None, None, None


def _mip_stack(tex, mip, max_mip_level):
    """-> (stack [Bt, texels, C], max level for the kernels) from None / a TextureMipWrapper / a list of custom level tensors"""
    Bt, Ht, Wt, C = tex.shape
    ml = -1 if max_mip_level is None else int(max_mip_level)
    if mip is None:
        return _MipBuild.apply(tex, ml), ml
    if isinstance(mip, TextureMipWrapper):
        if mip.shape != tuple(tex.shape):
            raise ValueError("texture: mip stack was built for a texture of shape %s, got %s" % (mip.shape, tuple(tex.shape)))
        wl = -1 if mip.max_mip_level is None else int(mip.max_mip_level)
        if ml >= 0 and (wl < 0 or ml < wl):
            raise ValueError("texture: max_mip_level must match the one the mip stack was built with")
        return mip.stack, wl
    # custom stack: a list of level tensors [Bt, h_l, w_l, C], l = 1..L; they receive their own gradients (not folded into `tex`)
    levels = list(mip)
    L = len(levels)
    if ml >= 0:
        L = min(L, ml)
    shapes, _ = _mip_levels(Ht, Wt, L)
    if len(shapes) - 1 != L:
        raise ValueError("texture: %d custom mip levels given, the texture supports %d" % (L, len(shapes) - 1))
    for l in range(L):
        if tuple(levels[l].shape) != (Bt, shapes[l + 1][0], shapes[l + 1][1], C):
            raise ValueError("texture: custom mip level %d has shape %s, expected %s" % (l + 1, tuple(levels[l].shape), (Bt,) + shapes[l + 1] + (C,)))
    if L == 0:
        return tex.new_zeros((Bt, 0, C)), 0
    return torch.cat([_h.f32c(levels[l]).reshape(Bt, -1, C) for l in range(L)], dim=1), L


def zero_boundary_as_clamp(tex, uv):
    """boundary_mode='zero' (the texture continued by zeros in every direction) expressed with what the kernels have: the texture padded
    by one zero texel per side and fetched in 'clamp' mode at coordinates moved by that texel -- every tap that falls outside lands on,
    or is clamped to, the zero border.  -> (padded texture [Bt,Ht+2,Wt+2,C], remapped uv); differentiable through both."""
    Ht, Wt = int(tex.shape[1]), int(tex.shape[2])
    padded = torch.nn.functional.pad(tex, (0, 0, 1, 1, 1, 1))
    scale = uv.new_tensor([Wt / (Wt + 2.0), Ht / (Ht + 2.0)])
    shift = uv.new_tensor([1.0 / (Wt + 2.0), 1.0 / (Ht + 2.0)])
    return padded, uv * scale + shift


def texture(tex, uv, uv_da=None, mip_level_bias=None, mip=None, filter_mode='auto', boundary_mode='wrap', max_mip_level=None):
    """-> [B,H,W,C].  'auto' = 'linear' without uv_da / mip_level_bias, 'linear-mipmap-linear' with (as the dependency documents).
    Mip-mapped modes propagate gradients to `tex` (through every level of the pyramid), to a custom `mip` list and to `uv`; uv_da and
    mip_level_bias receive none."""
    if filter_mode == 'auto':
        filter_mode = 'linear' if (uv_da is None and mip_level_bias is None) else 'linear-mipmap-linear'
    if filter_mode not in _FILTER and filter_mode not in _MIP_FILTER:
        raise ValueError("texture: unknown filter_mode %r" % (filter_mode,))
    if boundary_mode not in _BOUNDARY and boundary_mode != 'zero':
        raise NotImplementedError("texture: boundary_mode %r is not built" % (boundary_mode,))
    if tex.dim() != 4:
        raise NotImplementedError("texture: cube maps are not built")
    if boundary_mode == 'zero' and filter_mode in _FILTER:
        padded, uv_p = zero_boundary_as_clamp(tex, uv)
        return _Texture.apply(padded, uv_p, _FILTER[filter_mode], _BOUNDARY['clamp'])
    if filter_mode in _FILTER:
        return _Texture.apply(tex, uv, _FILTER[filter_mode], _BOUNDARY[boundary_mode])
    if uv_da is None and mip_level_bias is None:
        raise ValueError("texture: filter_mode %r needs uv_da or mip_level_bias" % (filter_mode,))
    stack, ml = _mip_stack(tex, mip, max_mip_level)
    # 'zero' with the mip-mapped filters: the kernels read 0 for a tap outside the level it belongs to (boundary code 2), level by level as nvdiffrast does
    return _TextureMip.apply(tex, stack, uv, uv_da, mip_level_bias, _MIP_FILTER[filter_mode], 2 if boundary_mode == 'zero' else _BOUNDARY[boundary_mode], ml)


# topology (edge hash) cache: the reference rebuilds it on every antialias call because it never passes topology_hash; the table only
# depends on `tri`, so it is kept per (storage, version, shape) of the triangle tensor and rebuilt when that changes.
_TOPO_CACHE = {}


def _topology(tri_c):
    lib = _h.lib()
    key = (tri_c.data_ptr(), tri_c._version, tuple(tri_c.shape), tri_c.device)
    hit = _TOPO_CACHE.get(key)
    if hit is not None:
        return hit[0]
    T = tri_c.shape[0]
    table = torch.empty((lib.c3d_mesh_antialias_scratch_bytes(T),), dtype=torch.uint8, device=tri_c.device)
    with torch.cuda.device(tri_c.device):
        _h.check(lib.c3d_mesh_antialias_build_topology(_h.ptr(tri_c if T else None), T, _h.ptr(table), _h.stream(tri_c.device)),
                 "c3d_mesh_antialias_build_topology")
    if len(_TOPO_CACHE) >= 8:
        _TOPO_CACHE.pop(next(iter(_TOPO_CACHE)))
    _TOPO_CACHE[key] = (table, tri_c)     # keeps tri_c alive so the pointer in the key cannot be recycled
    return table


class _Antialias(torch.autograd.Function):
    @staticmethod
    def forward(ctx, color, rast, pos, tri, pos_gradient_boost):
        lib = _h.lib()
        _dev_check(color, "color")
        color_c, rast_c, pos_c, tri_c = _h.f32c(color), _h.f32c(rast), _h.f32c(pos), tri.to(torch.int32).contiguous()
        B, H, W, C = color_c.shape
        V, T = pos_c.shape[1], tri_c.shape[0]
        dev = color_c.device
        with torch.cuda.device(dev):
            table = _topology(tri_c)
            out = torch.empty_like(color_c)
            _h.check(lib.c3d_mesh_antialias_fwd(_h.ptr(color_c), _h.ptr(rast_c), _h.ptr(pos_c), _h.ptr(tri_c if T else None), B, V, T, H, W, C,
                                                _h.ptr(table), _h.ptr(out), _h.stream(dev)), "c3d_mesh_antialias_fwd")
        ctx.save_for_backward(color_c, rast_c, pos_c, tri_c, table)
        ctx.boost = float(pos_gradient_boost)
        return out

    @staticmethod
    def backward(ctx, dy):
        lib = _h.lib()
        color, rast, pos, tri, table = ctx.saved_tensors
        B, H, W, C = color.shape
        V, T = pos.shape[1], tri.shape[0]
        dev = color.device
        with torch.cuda.device(dev):
            dcolor, dpos = torch.empty_like(color), torch.empty_like(pos)
            _h.check(lib.c3d_mesh_antialias_bwd(_h.ptr(color), _h.ptr(rast), _h.ptr(pos), _h.ptr(tri if T else None), _h.ptr(_h.f32c(dy)), B, V, T,
                                                H, W, C, _h.ptr(table), _h.ptr(dcolor), _h.ptr(dpos), _h.stream(dev)), "c3d_mesh_antialias_bwd")
        if ctx.boost != 1.0:
            dpos = dpos * ctx.boost
        return dcolor, None, dpos, None, None


def antialias(color, rast, pos, tri, topology_hash=None, pos_gradient_boost=1.0):
    """-> antialiased colour [B,H,W,C].  topology_hash is accepted and ignored (the edge hash is rebuilt per call, as it is
    for the reference, which never passes one)."""
    if pos.dim() == 2:      # range mode: one vertex buffer shared by the minibatch (see _rasterize_ranges)
        p1 = pos.unsqueeze(0)
        return torch.cat([_Antialias.apply(color[b:b + 1].contiguous(), rast[b:b + 1].contiguous(), p1, tri, pos_gradient_boost)
                          for b in range(color.shape[0])], 0) if color.shape[0] else color
    return _Antialias.apply(color, rast, pos, tri, pos_gradient_boost)


def antialias_construct_topology_hash(tri):
    """builds (and caches) the edge hash of `tri`; the returned handle may be passed as topology_hash (it is looked up again anyway)"""
    return _topology(tri.to(torch.int32).contiguous())


def get_log_level():
    return 1


def set_log_level(level):
    pass
