"""Fused multi-view training step of DiffMesh: all views of a step -- render, image loss, backward, gradients summed over the views -- in ONE sync-free
library call (include/c3d_mesh.h: c3d_mesh_train_views), the views going through every stage together (one launch per stage for up to 16 views; `lanes`
is kept for the round-2/3 signature and no longer changes anything).  MI355X-first replacement of the per-view loop
of the reference's trainer (MVs_Algorithms/DiffRastMesh/diff_mesh.py:98-125): the per-view autograd path is host bound at the BASELINE size (~35
launches per view enqueued from Python plus torch's own for the loss and the gradient accumulation)."""
import ctypes as C

import torch

import c3d_hip as _h
from c3d_hip.mesh_sigs import MeshStepLoss, MeshView


class FusedMeshStep:
    def __init__(self, device, lanes=4):
        self.device, self.lanes = torch.device(device), max(1, min(8, int(lanes)))
        self.loss = torch.zeros(1, dtype=torch.float32, device=self.device)
        self._key, self.workspace = None, None

    def run(self, views, v, v_offsets, f, vt, ft, raw_albedo, glctx, targets, masks, d_raw_albedo, d_v_offsets, H, W, w_mse=1.0, w_ssim=0.0, scale=1.0, accumulate=False):
        """views: [(clip_from_world 4x4 (numpy / sequence, row-major), bg (3 floats))]; targets / masks: per view [3,H,W] / [1,H,W] float32 device tensors (masks or
        entries None); v [V,3], v_offsets [V,3] | None, f / ft [T,3] int32, vt [Vt,2], raw_albedo [Ht,Wt,3].  d_raw_albedo / d_v_offsets (None: geometry not trained)
        are written (accumulate=False) or added to.  -> the loss of the step as a device scalar tensor (a view on a buffer the next run() reuses: clone to keep)."""
        import nvdiffrast.torch as _dr
        lib = _h.lib()
        n = len(views)
        if n == 0:
            if not accumulate:
                d_raw_albedo.zero_()
                if d_v_offsets is not None:
                    d_v_offsets.zero_()
            return torch.zeros(1, dtype=torch.float32, device=self.device)
        V, T, Vt = int(v.shape[0]), int(f.shape[0]), int(vt.shape[0])
        Ht, Wt = int(raw_albedo.shape[0]), int(raw_albedo.shape[1])
        key = (V, T, int(H), int(W), Ht, Wt, n, self.lanes)
        if key != self._key:
            nbytes = lib.c3d_mesh_step_workspace_bytes(V, T, int(H), int(W), Ht, Wt, n, self.lanes)
            self.workspace, self._key = torch.empty((nbytes,), dtype=torch.uint8, device=self.device), key
        arr = (MeshView * n)()
        for i, (clip, bg) in enumerate(views):
            arr[i] = MeshView(V, T, Vt, int(H), int(W), Ht, Wt, (C.c_float * 16)(*[float(x) for x in clip.reshape(-1)]), (C.c_float * 3)(*[float(x) for x in bg]))
        keep = [_h.f32c(t) for t in targets]
        tg = (C.c_void_p * n)(*[t.data_ptr() for t in keep])
        mk = None
        if masks is not None:
            keep_m = [(_h.f32c(m) if m is not None else None) for m in masks]
            mk = (C.c_void_p * n)(*[(m.data_ptr() if m is not None else None) for m in keep_m])
        geo = d_v_offsets is not None
        aa = _dr._topology(f)
        topo = glctx.vertex_topology(f, V) if geo else None
        loss = MeshStepLoss(float(w_mse), float(w_ssim), float(scale))
        self.loss.zero_()
        v_c, ra_c = _h.f32c(v), _h.f32c(raw_albedo)
        vo_c = _h.f32c(v_offsets) if v_offsets is not None else None
        with torch.cuda.device(self.device):
            _h.check(lib.c3d_mesh_train_views(arr, n, _h.ptr(v_c), _h.ptr(vo_c), _h.ptr(f), _h.ptr(vt), _h.ptr(ft), _h.ptr(ra_c), _h.ptr(aa), _h.ptr(topo), tg, mk, C.byref(loss),
                                              _h.ptr(d_raw_albedo), _h.ptr(d_v_offsets), _h.ptr(self.loss), 1 if accumulate else 0, self.lanes, _h.ptr(self.workspace),
                                              _h.stream(self.device)), "c3d_mesh_train_views")
        return self.loss
