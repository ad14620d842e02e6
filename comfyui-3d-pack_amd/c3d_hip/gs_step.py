"""Fused multi-view 3DGS training step: V views forward + backward in sync-free library calls, the parameter gradients written in place.
MI355X-first replacement of the per-view Python loop of the reference's trainer (main_3DGS.py:158-207).  Two forms:

  FusedViewStep.run()                     c3d_gs_train_views_raw: forward + the L1 / L2 / alpha-MSE pixel loss + backward in ONE call
  FusedViewStep.forward() / .backward()   c3d_gs_forward_views_raw / c3d_gs_backward_views_raw: the same step split at the image, so that any loss
                                          torch can differentiate (the reference's default adds MS-SSIM, main_3DGS.py:184-192) runs in between;
                                          every view carries its own background (camera_utils.py:246-249)

With 288 GB of HBM the (tile, splat) pair buffers are simply sized for a generous capacity and grown on the rare overflow."""
import ctypes as C
import time

import torch

import c3d_hip as _h


class FusedViewStep:
    def __init__(self, N, H, W, device, pair_capacity=None, lanes=1, views=1):
        """lanes: number of view GROUPS the step is cut into (1..8): ceil(V / lanes) views go through every stage of the chain together (one launch per stage), the
        groups one after the other on the caller's stream.  1 (default) = all views of the step in one group.  views: views per step the workspace is sized for (it
        grows on demand); see include/c3d_gs.h."""
        self.N, self.H, self.W, self.device = N, H, W, torch.device(device)
        self.lanes = max(1, min(8, int(lanes)))
        self.views = max(1, int(views))
        self.capacity = int(pair_capacity or max(8 * N, 1 << 22))
        # status words and the loss value of a step side by side, every step a FRESH, zero-initialised slot of a ring (a new ring when one is used up: old loss tensors keep
        # theirs alive): no fill launch and no clone launch per step -- the node's default scene is ~30 launches of 4-80 us per iteration, and these were two of them
        self._ring, self._ring_next = None, 0
        self._host_ring, self._host_next = None, 0      # pinned, device-mapped: where the step's loss launch leaves the status words for the host (c3d_gs_train_views_raw: status_host)
        self._new_words()
        self._fitted = False
        self._chunks_went_out = False
        self._fwd = None
        self.time_events = False
        # defer_status: once the capacity is fitted, run() does not wait for its own status words -- they are copied to pinned memory asynchronously and
        # examined when the NEXT run() (or finish()) starts, so the host enqueues step k+1 while the GPU still works on step k (the single sync per step
        # left a 0.26 ms bubble at every step boundary, 3.5 % of the 8-view step).  A device fault still raises; a pair overflow between two consecutive
        # steps of a fitted scene is then noticed one step late: the capacity is regrown and a warning says that the previous step's gradient was
        # incomplete.  To make that rare the capacity follows the scene: whenever a step is seen to use more than GROW_AT of it (the status word carries
        # the largest pair count of the step), it is regrown to 1.3 x that count BEFORE the next step -- splat footprints that grow gradually over a run
        # never reach the limit; only a jump of > 25 % between two consecutive steps can still overflow.  Off by default; the trainer and bench.py turn it on.
        self.defer_status = False
        # multi-GPU: a callable that replaces three int32 device words by their MAXIMUM over the ranks, in place and in stream order (c3d_hip.parallel.status_max(group)).
        # Every decision about a step -- fit, regrow, redo -- is then taken from the same numbers on every rank; see run().
        self.status_sync = None
        self._pending = None
        self._pinned = None
        self._flip = 0
        self._alloc()

    GROW_AT = 0.8      # regrow the pair capacity when a step used more than this share of it
    RING = 1024        # steps per ring of device words / pinned host words

    def _new_words(self):
        if self._ring is None or self._ring_next >= self.RING:
            self._ring, self._ring_next = torch.zeros((self.RING, 4), dtype=torch.int32, device=self.device), 0
        self._words = self._ring[self._ring_next]
        self._ring_next += 1
        self.status = self._words[0:2]
        self.loss = self._words[2:3].view(torch.float32)

    def _host_slot(self):
        """-> (numpy view of two pinned int32 words preset to the sentinel, their address)"""
        if self._host_ring is None:
            self._host_pin = torch.full((self.RING, 2), -1, dtype=torch.int32).pin_memory()
            self._host_ring = self._host_pin.numpy()
        i = self._host_next
        self._host_next = (i + 1) % self.RING
        self._host_ring[i] = -1
        return self._host_ring[i], self._host_pin.data_ptr() + 8 * i

    def _alloc(self):
        nbytes = _h.lib().c3d_gs_step_workspace_bytes(self.N, self.H, self.W, self.capacity, self.views)
        self.workspace = torch.empty((nbytes,), dtype=torch.uint8, device=self.device)

    def _follow(self, seen):
        """keep headroom above what the scene needs: called with the largest pair count of a CLEAN step"""
        if self._fitted and seen > self.GROW_AT * self.capacity:
            self.capacity = int(seen * 1.3) + 4096
            self._alloc()       # the finished step's buffers stay alive through self._last until the next step replaces them

    @staticmethod
    def _settings(rs_list, keep, params=None):
        """params: the raw parameter list of the call -- its f_dc / f_rest pair says how many SH coefficients the storage holds (any PLY degree 0..3)"""
        import diff_gaussian_rasterization as dgr
        K = dgr.raw_sh_coeffs(params[1], params[2]) if params is not None else 0
        arr = (_h.GsSettings * len(rs_list))()
        for i, rs in enumerate(rs_list):
            arr[i] = dgr._settings_struct(rs, keep, K)
        return arr

    def _status_words(self):
        """-> the step's status on the device: this rank's own two words, or as three int32 words {overflow, fault, largest pair count} -- with status_sync set (multi-GPU) -- the maximum over
        the ranks, so that every rank takes the same decision about the step (and sizes the same capacity)"""
        st = self.status
        if self.status_sync is None:
            return st                     # {flags, count}: decoded on the host (_decode), no extra launch on the single-GPU path
        words = torch.stack([st[0] & 1, (st[0] >> 1) & 1, st[1]])
        self.status_sync(words)
        return words

    @staticmethod
    def _decode(words):
        """host copy of _status_words() -> (overflow, fault, largest pair count)"""
        w = [int(x) for x in words]
        if len(w) == 2:
            return w[0] & 1, w[0] & 2, w[1] & 0xFFFFFFFF
        return w[0], w[1], w[2] & 0xFFFFFFFF

    def run(self, raster_settings, params, grads, target_color, target_alpha=None, color_mask=None, w_l1=1.0, w_l2=0.0, w_alpha_mse=0.0, scale=1.0, max_retries=3,
            accumulate=True, w_ssim=0.0, param_chunks=1, after_chunk=None):
        """params / grads: (xyz, f_dc, f_rest, opacity_raw, scaling_raw, rotation_raw) tensors; accumulate=True adds to the grads (zero them per
        step), False overwrites them (no zero-fill needed).  -> loss tensor (device scalar, the sum over the views).  Synchronises once, at
        the end, to read the overflow flag (not at all with defer_status).
        after_chunk(g0, g1): called when the gradient rows [g0, g1) of every tensor have been ENQUEUED in final form on the current stream, for the
        SAME sequence of `param_chunks` ranges on every call (so that ranks which start a collective per range stay in step, whatever their local
        state): with a fitted capacity and accumulate=False the per-Gaussian pass runs range by range and the callback follows each range (the
        multi-GPU trainer starts that range's collective there, underneath the next range's kernels); otherwise -- first step, no views on this
        rank -- the ranges are handed over one after the other once the whole pass has been enqueued and found good.
        Multi-GPU (status_sync set): the status words every decision below is taken from are the MAXIMUM over the ranks, so a view that exceeds the pair capacity on one rank
        makes ALL ranks regrow and redo the step together (the ranges' collectives are then issued a second time, by everybody) instead of raising on that rank and
        leaving the others in a collective nobody will join."""
        lib = _h.lib()
        V = len(raster_settings)
        K = max(1, int(param_chunks)) if self.N >= 1024 * max(1, int(param_chunks)) else 1
        bounds = [min(self.N, (self.N * i // K + 255) // 256 * 256) for i in range(K)] + [self.N]
        ranges = list(zip(bounds[:-1], bounds[1:]))
        together = self.status_sync is not None

        def hand_over():
            if after_chunk is not None:
                for g0, g1 in ranges:
                    after_chunk(g0, g1)
        prev, self._pending = self._pending, None      # a deferred previous step: examined AFTER this one is enqueued, so the GPU never waits for the host
        if V == 0 and not together:
            self._examine(prev)                       # a rank without views this step still owns well-defined gradients
            if not accumulate:
                for g in grads:
                    g.zero_()
            hand_over()
            return torch.zeros(1, dtype=torch.float32, device=self.device)
        if V > self.views:
            self.views = V
            self._alloc()
        for attempt in range(max_retries + 1):
            keep = []
            if V:
                views = self._settings(raster_settings, keep, params)
                tc = (C.c_void_p * V)(*[t.data_ptr() for t in target_color])
                ta = (C.c_void_p * V)(*[t.data_ptr() for t in target_alpha]) if target_alpha is not None else None
                cm = (C.c_void_p * V)(*[t.data_ptr() for t in color_mask]) if color_mask is not None else None
                loss = _h.GsLoss(float(w_l1), float(w_l2), float(w_alpha_mse), float(scale), float(w_ssim))
            if accumulate:
                snapshot = [g.clone() for g in grads] if attempt == 0 else snapshot     # to redo the step after an overflow
            self._new_words()
            t_host = time.perf_counter()
            if self.time_events:
                ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                ev0.record(torch.cuda.current_stream(self.device))
            chunked = K > 1 and after_chunk is not None and self._fitted and not accumulate
            self._chunks_went_out = chunked
            deferred = self.defer_status and self._fitted and not accumulate
            # one GPU, deferred status, nobody timing the call: the step's loss launch stores the status words into pinned memory itself, and the NEXT step waits for those
            # words only (they are final two kernels before the step ends) -- no copy launch, no event
            early = deferred and not together and not self.time_events and V > 0
            host_words, host_ptr = self._host_slot() if early else (None, None)
            if V:
                with torch.cuda.device(self.device):
                    pp = [_h.ptr(_h.f32c(p)) for p in params]
                    _h.check(lib.c3d_gs_train_views_raw(views, V, self.N, *pp, tc, ta, cm, C.byref(loss),
                                                        *[_h.ptr(g) for g in grads], _h.ptr(self.loss), self.capacity, self.lanes, (1 if accumulate else 0) | (2 if chunked else 0),
                                                        _h.ptr(self.workspace), _h.ptr(self.status), C.c_void_p(host_ptr) if early else None, _h.stream(self.device)),
                             "c3d_gs_train_views_raw")
                    if chunked:
                        for g0, g1 in ranges:
                            _h.check(lib.c3d_gs_step_param_backward_range(views, V, self.N, pp[0], pp[1], pp[2], pp[4], pp[5], *[_h.ptr(g) for g in grads], self.capacity, 0,
                                                                          _h.ptr(self.workspace), g0, g1 - g0, _h.stream(self.device)), "c3d_gs_step_param_backward_range")
                            after_chunk(g0, g1)
            else:      # a rank without views this step (multi-GPU): zero gradients, and the same sequence of range hand-overs at the same point as everybody else
                if not accumulate:
                    for g in grads:
                        g.zero_()
                if chunked:
                    hand_over()
            self.last_host_ms = (time.perf_counter() - t_host) * 1e3      # host time to enqueue the whole step (no sync inside)
            if self.time_events:
                ev1.record(torch.cuda.current_stream(self.device))
            if deferred:
                if early:
                    self._pending = (None, None, host_words)
                else:
                    words = self._status_words()
                    if self._pinned is None:
                        self._pinned = [torch.empty((3,), dtype=torch.int32).pin_memory() for _ in range(2)]
                    self._flip ^= 1
                    pin = self._pinned[self._flip][:words.numel()]
                    pin.copy_(words, non_blocking=True)
                    ev = torch.cuda.Event()
                    ev.record(torch.cuda.current_stream(self.device))
                    self._pending = (ev, (ev0, ev1) if self.time_events else None, pin)
                self._last = (self.workspace, self.capacity)
                out = self.loss
                if prev is not None and not self._examine(prev):
                    prev = None
                    self._pending = None
                    continue             # the previous step had overflowed (capacity regrown): this one certainly did too -- redo it, synchronously (multi-GPU: on every rank)
                if not chunked:
                    hand_over()
                return out
            if prev is not None:
                self._examine(prev)
                prev = None
            words = self._status_words()
            ovf, fault, seen = self._decode(words.tolist())       # the single host sync of the step
            if self.time_events:
                self.last_gpu_ms = ev0.elapsed_time(ev1)                  # GPU span of the library call: wall time beyond it is host-side bubble
            self._raise_on_fault(fault, seen)
            if not ovf:
                self._last = (self.workspace, self.capacity)     # what read_view() looks into
                if not self._fitted and self.capacity > 1.6 * max(seen, 1 << 16):      # (every launch of the binning chain is sized for the capacity: more than 1.6 x the need is worth one reallocation)
                    # first successful step: every launch is sized for the capacity, so bring it down to what the scene needs (+30 %)
                    self.capacity = int(seen * 1.3) + 4096
                    self._alloc()
                self._fitted = True
                self._follow(seen)
                if not chunked:
                    hand_over()
                return self.loss
            if chunked and not together:
                raise RuntimeError("c3d FusedViewStep: a step whose gradient ranges had already been handed to after_chunk exceeded the pair capacity (%d pairs)" % seen)
            self.capacity = int(max(seen, self.capacity) * 1.25) + 1024
            self._alloc()
            if accumulate:
                for g, s0 in zip(grads, snapshot):
                    g.copy_(s0)
        raise RuntimeError("c3d FusedViewStep: pair capacity still exceeded after %d retries" % max_retries)

    def finish(self):
        """examine the status of a deferred run() (see defer_status): whoever needs the last step to be known-good calls this"""
        prev, self._pending = self._pending, None
        self._examine(prev)

    def _examine(self, pending):
        """-> True when the deferred step was clean (or there was none)"""
        if pending is None:
            return True
        ev, tev, pin = pending
        if ev is None:      # the words come straight from the step's loss launch (pinned memory): wait for them, not for the step
            _h.check(_h.lib().c3d_gs_wait_count(C.c_void_p(pin.ctypes.data), 0xFFFFFFFF, 20_000_000, None, None), "c3d_gs_wait_count")
        else:
            ev.synchronize()
        if tev is not None:
            self.last_gpu_ms = tev[0].elapsed_time(tev[1])
        ovf, fault, seen = self._decode(pin.tolist())
        self._raise_on_fault(fault, seen)
        if ovf:
            import warnings
            self.capacity = int(max(seen, self.capacity) * 1.25) + 1024
            self._alloc()
            if self._chunks_went_out and self.status_sync is None:      # a redo would issue the ranges' collectives a second time on this rank only: fail here, not in a hung collective
                raise RuntimeError("c3d FusedViewStep: a step whose gradient ranges had already been handed to after_chunk needed %d (tile, splat) pairs, more "
                                   "than the fitted capacity; set status_sync (c3d_hip.parallel.status_max) so that the ranks redo such a step together" % seen)
            warnings.warn("c3d FusedViewStep: the previous step needed %d (tile, splat) pairs, more than the fitted capacity; its gradient was incomplete "
                          "(noticed one step late because defer_status is on); capacity regrown to %d" % (seen, self.capacity), RuntimeWarning)
            return False
        self._follow(seen)
        return True

    # ---- the step split at the image --------------------------------------------------------------------------------------------------------
    def _raise_on_fault(self, fault, seen=0):
        if fault:
            raise RuntimeError("c3d: a bounded inter-workgroup wait of the binning stage timed out (pair count word %r): device fault" % (seen,))

    def forward(self, raster_settings, params, want_depth=False, want_radii=False, max_retries=3):
        """All V views forward, state kept per view for backward().  params: (xyz, f_dc, f_rest, opacity_raw, scaling_raw, rotation_raw).
        -> color [V,3,H,W] (unclamped), depth [V,1,H,W] | None, alpha [V,1,H,W], radii [V,N] | None.  One host sync (overflow flag)."""
        lib = _h.lib()
        V = len(raster_settings)
        f32 = dict(dtype=torch.float32, device=self.device)
        color, alpha = torch.empty((V, 3, self.H, self.W), **f32), torch.empty((V, 1, self.H, self.W), **f32)
        depth = torch.empty((V, 1, self.H, self.W), **f32) if want_depth else None
        radii = torch.empty((V, self.N), dtype=torch.int32, device=self.device) if want_radii else None
        self._fwd = None
        if V == 0:
            return color, depth, alpha, radii
        if V > self.views:
            self.views = V
            self._alloc()
        arr = lambda t: (C.c_void_p * V)(*[t[i].data_ptr() for i in range(V)]) if t is not None else None
        pin = [_h.f32c(p) for p in params]
        for attempt in range(max_retries + 1):
            keep = []
            views = self._settings(raster_settings, keep, params)
            self.status.zero_()
            t_host = time.perf_counter()
            with torch.cuda.device(self.device):
                _h.check(lib.c3d_gs_forward_views_raw(views, V, self.N, *[_h.ptr(p) for p in pin], arr(color), arr(depth), arr(alpha), arr(radii), self.capacity,
                                                      self.lanes, _h.ptr(self.workspace), _h.ptr(self.status), _h.stream(self.device)), "c3d_gs_forward_views_raw")
            self.last_host_ms = (time.perf_counter() - t_host) * 1e3
            st = self.status.tolist()       # the single host sync of the forward half
            self._raise_on_fault(st[0] & 2, st[1])
            seen = st[1] & 0xFFFFFFFF
            if st[0] == 0:
                if not self._fitted and self.capacity > 1.6 * max(seen, 1 << 16):      # (every launch of the binning chain is sized for the capacity: more than 1.6 x the need is worth one reallocation)
                    self._fitted = True
                    self.capacity = int(seen * 1.3) + 4096      # first success: every launch is sized for the capacity -> fit it (+30 %) and redo
                    self._alloc()
                    continue
                self._fitted = True
                self._last = (self.workspace, self.capacity)
                self._fwd = (keep, views, V, pin)               # backward() must see the very same settings / workspace
                return color, depth, alpha, radii
            self.capacity = int(max(seen, self.capacity) * 1.25) + 1024
            self._alloc()
        raise RuntimeError("c3d FusedViewStep.forward: pair capacity still exceeded after %d retries" % max_retries)

    def backward(self, grads, dL_dcolor, dL_dalpha=None, dL_ddepth=None, accumulate=False):
        """Backward of the views of the last forward(): dL_dcolor [V,3,H,W] (w.r.t. the unclamped colour), dL_dalpha / dL_ddepth [V,1,H,W] | None
        -> parameter gradients into `grads` (overwritten, or added to with accumulate=True).  No host synchronisation."""
        if self._fwd is None:
            raise RuntimeError("c3d FusedViewStep.backward: no forward() state to differentiate")
        keep, views, V, pin = self._fwd
        g = lambda t: _h.f32c(t) if t is not None else None
        dc, da, dd = g(dL_dcolor), g(dL_dalpha), g(dL_ddepth)
        arr = lambda t: (C.c_void_p * V)(*[t[i].data_ptr() for i in range(V)]) if t is not None else None
        t_host = time.perf_counter()
        with torch.cuda.device(self.device):
            _h.check(_h.lib().c3d_gs_backward_views_raw(views, V, self.N, _h.ptr(pin[0]), _h.ptr(pin[1]), _h.ptr(pin[2]), _h.ptr(pin[4]), _h.ptr(pin[5]), arr(dc), arr(dd), arr(da),
                                                        *[_h.ptr(q) for q in grads], self.capacity, self.lanes, 1 if accumulate else 0, _h.ptr(self.workspace),
                                                        _h.stream(self.device)), "c3d_gs_backward_views_raw")
        self.last_host_ms_bwd = (time.perf_counter() - t_host) * 1e3

    def read_view(self, view):
        """-> (radii [N] int32, dL/dmeans2D [N,3]) of view `view` of the last run(): the densification statistics of the reference trainer"""
        ws, cap = self._last
        radii = torch.empty((self.N,), dtype=torch.int32, device=self.device)
        g2 = torch.empty((self.N, 3), dtype=torch.float32, device=self.device)
        with torch.cuda.device(self.device):
            _h.check(_h.lib().c3d_gs_step_read_view(self.N, self.H, self.W, cap, _h.ptr(ws), int(view), _h.ptr(radii), _h.ptr(g2),
                                                    _h.stream(self.device)), "c3d_gs_step_read_view")
        return radii, g2


    def accumulate_densify_stats(self, view, grad_accum, denom, max_radii=None):
        """the reference trainer's densification statistics of view `view` of the last run() (GaussianModel.add_densification_stats + the max_radii2D update), added in place,
        straight from the workspace: one launch (c3d_gs_step_accumulate_densify_stats) instead of read_view()'s two copies and the torch ops on them"""
        ws, cap = self._last
        for t in (grad_accum, denom, max_radii):
            if t is not None and (t.dtype != torch.float32 or not t.is_contiguous() or t.numel() != self.N):
                raise ValueError("accumulate_densify_stats: float32 contiguous tensors of N elements")
        with torch.cuda.device(self.device):
            _h.check(_h.lib().c3d_gs_step_accumulate_densify_stats(self.N, self.H, self.W, cap, _h.ptr(ws), int(view), _h.ptr(grad_accum), _h.ptr(denom), _h.ptr(max_radii),
                                                                  _h.stream(self.device)), "c3d_gs_step_accumulate_densify_stats")


class FusedViewRender:
    """Forward only: V views of one cloud per library call (c3d_gs_render_views_raw), `group` views per launch of every stage, no host synchronisation between views --
    the orbit-rendering loop of the reference's renderer nodes in one call.
    streams > 1 (round 6): the views are cut into up to `streams` contiguous parts of >= 8 views, each part one library call on a HIP stream of its own (own workspace),
    forked from and joined back into the caller's stream.  The binning chain of a part -- launches that are latency or memory bound -- then runs underneath the compositing of
    another part, which is bound by instruction issue: BASELINE config 2 (64 cameras) 15.15 -> 14.39 ms on one box with four streams, the images bit-identical
    (profiles/r06/r06s_t_small_launches_and_streams.txt, r06t_streams_*.txt).  The library itself still owns no stream."""

    MIN_PART = 8       # views per part: narrower launches cost more than further overlap brings (4 parts of 8 views: +35 % on a 10 k-Gaussian 512^2 orbit, +3.5 % at 200 k / 1024^2, +6 % at config 2)

    def __init__(self, N, H, W, device, pair_capacity=None, lanes=1, group=16, streams=1):
        """group: views that go through the chain together (one launch per stage; <= 16); the groups follow each other on their stream and reuse the workspace's
        `group` forward-only slices.  lanes: kept for callers of earlier rounds (a lower bound on the number of groups a call is cut into).  streams: see the class."""
        self.N, self.H, self.W, self.device = N, H, W, torch.device(device)
        self.lanes = max(1, min(8, int(lanes)))
        self.group = max(1, min(16, int(group)))
        self.streams = max(1, min(8, int(streams)))
        self.slices = self.group
        self.capacity = int(pair_capacity or max(8 * N, 1 << 22))
        self.status = torch.zeros(2, dtype=torch.int32, device=self.device)
        self._fitted = False
        self._parts, self._side = [], []
        self.workspace = None
        if self.streams == 1:
            self._alloc()

    def _alloc(self):
        nbytes = _h.lib().c3d_gs_render_workspace_bytes(self.N, self.H, self.W, self.capacity, self.slices)     # `group` forward-only slices: see c3d_gs_render_views_raw
        self.workspace = torch.empty((nbytes,), dtype=torch.uint8, device=self.device)

    def _enqueue(self, raster_settings, params, color, depth, alpha, radii, keep):
        """one library call for these views on the CURRENT stream; nothing is waited for.  keep: list that holds what the call's arguments point at"""
        V = len(raster_settings)
        arr = lambda t: (C.c_void_p * V)(*[t[i].data_ptr() for i in range(V)])
        views = FusedViewStep._settings(raster_settings, keep, params)
        keep.append(views)
        self.status.zero_()
        with torch.cuda.device(self.device):
            _h.check(_h.lib().c3d_gs_render_views_raw(views, V, self.N, *[_h.ptr(_h.f32c(p)) for p in params], arr(color), arr(depth), arr(alpha),
                                                      arr(radii) if radii is not None else None, self.capacity, self.lanes, _h.ptr(self.workspace), self.workspace.numel(),
                                                      _h.ptr(self.status), _h.stream(self.device)), "c3d_gs_render_views_raw")

    def _plan(self, V):
        """-> (number of parts, views per part)"""
        S = max(1, min(self.streams, V // self.MIN_PART))
        return S, (V + S - 1) // S

    def _make_parts(self, S, per):
        g = min(self.group, per)
        if len(self._parts) != S or any(p.group != g or p.capacity != self.capacity for p in self._parts):
            self._parts = []      # (frees the old workspaces first)
            self._parts = [FusedViewRender(self.N, self.H, self.W, self.device, pair_capacity=self.capacity, lanes=self.lanes, group=g) for _ in range(S)]
        while len(self._side) < S:
            self._side.append(torch.cuda.Stream(self.device))

    def run(self, raster_settings, params, want_radii=False, max_retries=3):
        """params: (xyz, f_dc, f_rest, opacity_raw, scaling_raw, rotation_raw) -> color [V,3,H,W], depth [V,1,H,W], alpha [V,1,H,W], radii [V,N] | None"""
        V = len(raster_settings)
        f32 = dict(dtype=torch.float32, device=self.device)
        color, depth, alpha = torch.empty((V, 3, self.H, self.W), **f32), torch.empty((V, 1, self.H, self.W), **f32), torch.empty((V, 1, self.H, self.W), **f32)
        radii = torch.empty((V, self.N), dtype=torch.int32, device=self.device) if want_radii else None
        if V == 0:
            return color, depth, alpha, radii
        S, per = self._plan(V)
        for attempt in range(max_retries + 1):
            keep = []
            if S == 1:
                if self.workspace is None:
                    self._alloc()
                self._enqueue(raster_settings, params, color, depth, alpha, radii, keep)
                st = self.status.tolist()       # the single host sync of the call
            else:
                self._make_parts(S, per)
                cur, ran = torch.cuda.current_stream(self.device), []
                try:
                    for h, part in enumerate(self._parts):      # fork: every part on its own stream, behind what the caller's stream holds so far
                        v0, v1 = h * per, min(V, (h + 1) * per)
                        if v0 >= v1:
                            continue
                        ran.append(part)
                        side = self._side[h]
                        side.wait_stream(cur)
                        with torch.cuda.stream(side):
                            part._enqueue(raster_settings[v0:v1], params, color[v0:v1], depth[v0:v1], alpha[v0:v1], radii[v0:v1] if radii is not None else None, keep)
                finally:                                         # join, also when a part's call raised: the caller's stream carries on when everything that was enqueued is
                    for h in range(S):                           # done (outputs, parameters and workspaces are safe to reuse or free)
                        cur.wait_stream(self._side[h])
                sts = torch.stack([part.status for part in ran]).tolist()       # the single host sync of the call
                st = [0, 0]
                for w in sts:
                    st = [st[0] | w[0], max(st[1], w[1] & 0xFFFFFFFF)]
            if st[0] & 2:
                raise RuntimeError("c3d: a bounded inter-workgroup wait of the binning stage timed out (status %r): device fault" % (st,))
            seen = st[1] & 0xFFFFFFFF
            if st[0] == 0:
                if not self._fitted and self.capacity > 1.6 * max(seen, 1 << 16):      # (every launch of the binning chain is sized for the capacity: more than 1.6 x the need is worth one reallocation)
                    self.capacity = int(seen * 1.3) + 4096
                    self._realloc(S == 1)
                self._fitted = True
                return color, depth, alpha, radii
            self.capacity = int(max(seen, self.capacity) * 1.25) + 1024
            self._realloc(S == 1)
        raise RuntimeError("c3d FusedViewRender: pair capacity still exceeded after %d retries" % max_retries)

    def _realloc(self, single=None):
        """the buffers again at self.capacity: the one-call workspace at once when that is what the caller uses (single; None: when the object has one stream), the parts'
        workspaces when the next multi-stream run builds them"""
        self._parts = []
        self.workspace = None
        if single or (single is None and self.streams == 1):
            self._alloc()
