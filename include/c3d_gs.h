/*
 * c3d_gs.h -- C-ABI of the MI355X-native 3D-Gaussian-Splatting rasterizer (libc3d_hip.so).
 *
 * Drop-in boundary: these entry points are what a replacement for the native extension of
 * `diff_gaussian_rasterization` must export.  The reference reaches that extension at
 *   /root/reference/MVs_Algorithms/GaussianSplatting/main_3DGS_renderer.py:840-843  (import)
 *   /root/reference/MVs_Algorithms/GaussianSplatting/main_3DGS_renderer.py:849-864  (settings)
 *   /root/reference/MVs_Algorithms/GaussianSplatting/main_3DGS_renderer.py:927-936  (rasterizer call)
 * and, through the same Python package, from Gen_3D_Modules/LGM/core/gs.py:27-80,
 * Gen_3D_Modules/TriplaneGaussian/models/renderer.py:209-270 and
 * Gen_3D_Modules/TRELLIS/trellis/renderers/gaussian_render.py:62-130.
 * The dependency's native layer exposes three functions (SURVEY.md 8b "Level 2"):
 *   rasterize_gaussians           -> c3d_gs_forward_project + c3d_gs_forward_render
 *   rasterize_gaussians_backward  -> c3d_gs_backward
 *   mark_visible                  -> c3d_gs_mark_visible
 * The forward is split in two because the number of (tile, splat) pairs is data dependent: the
 * caller allocates the binning buffer between the calls (the dependency does the same through a
 * resize callback into torch's allocator).
 *
 * Conventions
 *  - plain C: pointers + sizes, no torch / C++ types.  All tensor pointers are DEVICE pointers to
 *    contiguous float32 / int32 arrays unless a parameter says "host".
 *  - every function launches on `stream` (a hipStream_t passed as void*) and returns 0 on success,
 *    non-zero otherwise; c3d_last_error() then returns a message (thread-local).
 *  - "optional" tensors are passed as NULL (exactly one of shs / colors_precomp and one of
 *    (scales,rotations) / cov3D_precomp must be non-NULL, as the dependency's Python wrapper demands).
 *  - buffers whose size the library decides are opaque byte buffers owned by the caller; ask the
 *    *_bytes() functions for their size.  They must be kept until backward has run.
 *  - matrices use the storage the reference passes: viewmatrix = world_view_transform = w2c^T and
 *    projmatrix = full_proj_transform, both row-major 4x4 (shared_utils/camera_utils.py:205-213).
 */
#ifndef C3D_GS_H
#define C3D_GS_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#ifndef C3D_STREAM_T
#define C3D_STREAM_T
typedef void* c3d_stream_t; /* hipStream_t */
#endif

/* mirrors GaussianRasterizationSettings (main_3DGS_renderer.py:849-862) */
typedef struct c3d_gs_settings {
    int32_t image_height;
    int32_t image_width;
    float tanfovx;
    float tanfovy;
    float scale_modifier;
    int32_t sh_degree;
    int32_t prefiltered;
    int32_t debug;
    const float* bg;         /* device [3]  */
    const float* viewmatrix; /* device [16] */
    const float* projmatrix; /* device [16] */
    const float* campos;     /* device [3]  */
    int32_t flags;           /* C3D_GS_FLAG_*; 0 = the dependency's behaviour */
    int32_t sh_coeffs;       /* raw-parameter entry points (*_raw): SH coefficients per channel the f_dc / f_rest pair STORES: 16 (or 0: the same), 9, 4 or 1, i.e.
                              * f_rest is [N, sh_coeffs - 1, 3] (NULL for 1) -- a PLY of any degree 0..3 (mesh_processer/mesh_utils.py:346-350; LGM writes degree 0).
                              * sh_degree is the ACTIVE degree, (sh_degree + 1)^2 <= sh_coeffs.  Ignored by the entry points that take M explicitly. */
} c3d_gs_settings;

/* flags.  C3D_GS_FLAG_EXACT_DSCALE -- dL/dscale convention of the backward entry points that receive these settings.  Clear (default): as the
 * dependency's backward returns it -- Sigma = R diag(m s)^2 R^T is differentiated w.r.t. (m s) and handed back as dL/dscale, i.e. WITHOUT the
 * scale_modifier factor m (identical to the exact derivative at m = 1, the only value the reference trains with; main_3DGS_renderer.py:830
 * exposes `scaling_modifier` for inference).  Set: the exact derivative (x m).  Per call, carried by the settings: the library keeps no
 * process-wide switches ("no global state besides per-device contexts"). */
#define C3D_GS_FLAG_EXACT_DSCALE 1
/* C3D_GS_FLAG_FORWARD_ONLY (ABI 600) -- forward entry points of the drop-in path (c3d_gs_forward_render, c3d_gs_forward_nosync, c3d_gs_forward_raw_nosync): no backward call will
 * follow this render (an inference caller: torch.no_grad(), the orbit nodes, LGM / TGS / TRELLIS-style consumers).  The compositing launch then records no pair activity, runs
 * no record-base scan and stores no per-pixel backward state (final_T, n_contrib: 8 of the 28 bytes it writes per pixel) -- same images, same radii, bit for bit.  A backward
 * entry point that receives settings with this flag returns an error instead of reading state that was never written.  The Python boundary sets it whenever the call is not
 * differentiated (grad mode off, or no input requires a gradient). */
#define C3D_GS_FLAG_FORWARD_ONLY 2
/* C3D_GS_FLAG_KEEP_RECORD_BASES (ABI 600) -- c3d_gs_forward_render on a geometry buffer whose view has been composited once already (c3d_gs_forward_nosync reported through its
 * count words that the view needs more pairs than its buffers held: the caller allocates a binning buffer for the exact count and renders the second half again): the
 * record-base scan of the first compositing launch is kept (its results do not depend on the pair buffers, its single-pass state is spent), and the device copy of the pair
 * count is set to num_rendered. */
#define C3D_GS_FLAG_KEEP_RECORD_BASES 4

const char* c3d_last_error(void);
int c3d_version(void);

/* sizes (bytes) of the three opaque state buffers (geometry / binning / image) */
size_t c3d_gs_geom_bytes(int32_t N);
size_t c3d_gs_binning_bytes(int64_t num_rendered, int32_t image_height, int32_t image_width);
size_t c3d_gs_image_bytes(int32_t image_height, int32_t image_width);
size_t c3d_gs_backward_scratch_bytes(int32_t N, int64_t num_rendered);

/* Forward, part 1 (A1 + depth ordering + offsets): projects N Gaussians (SH with M coefficients per
 * channel, layout [N,M,3]), writes radii[N] (int32, 0 = culled) and the geometry buffer, and returns
 * the number of (tile, splat) pairs through the HOST pointer num_rendered (the call synchronises the
 * stream for that one value). */
int c3d_gs_forward_project(const c3d_gs_settings* st, int32_t N, int32_t M, const float* means3D, const float* shs,
                           const float* colors_precomp, const float* opacities, const float* scales,
                           const float* rotations, const float* cov3D_precomp, int32_t* radii, void* geom_buffer,
                           int64_t* num_rendered /* host */, c3d_stream_t stream);

/* Forward, part 2 (A3-A6): bins, orders and composites.  out_color[3,H,W], out_depth[1,H,W],
 * out_alpha[1,H,W].  The compositing launch also records which (quadrant, splat) pairs blended and carries the record-base scan of the backward pass in its first
 * workgroups (ABI 500: both feed c3d_gs_backward only, neither is on the forward chain's critical path; C3D_GS_FLAG_FORWARD_ONLY leaves both out).  The scan walks the
 * geometry of all N Gaussians: geom_buffer and radii must be non-NULL whenever N > 0, whatever num_rendered is.  A timed-out hand-over of that scan (a device fault) raises
 * C3D_ERR_LOOKBACK in the geometry buffer's own error word, which only the NEXT c3d_gs_forward_project on that buffer would read: callers that need the fault reported for this
 * render take c3d_gs_forward_nosync, whose status words carry it. */
int c3d_gs_forward_render(const c3d_gs_settings* st, int32_t N, int32_t M, const int32_t* radii, void* geom_buffer,
                          int64_t num_rendered, void* binning_buffer, void* image_buffer, float* out_color,
                          float* out_depth, float* out_alpha, c3d_stream_t stream);

/* Forward in ONE call without host synchronisation (ABI 500; hint-sized launches and count words since ABI 600): parts 1 + 2 with the pair count left on the device, so the
 * rasterizer call of main_3DGS_renderer.py:927-936 (and of LGM core/gs.py:27-80, TRELLIS gaussian_render.py:62-130) no longer stalls the GPU once per view while the host
 * reads one number.  The reference never returns an incomplete image (its binning buffer is sized from the exact count); this call gives its caller the means to guarantee the same:
 *   pair_capacity   what the BUFFERS hold: binning_buffer = c3d_gs_binning_bytes(pair_capacity, H, W) bytes; a backward call on this state passes num_rendered = pair_capacity
 *                   and scratch of c3d_gs_backward_scratch_bytes(N, pair_capacity).
 *   first_capacity  what the LAUNCHES are sized for (0 or >= pair_capacity: for pair_capacity): a hint, not a limit.  A view with more pairs than the hint is served by
 *                   workgroups that loop -- every count up to pair_capacity is rendered exactly, bit for bit the synchronous path's result, and a count within the hint costs
 *                   nothing extra (no second launch, no gate).  Buffers are cheap (65 bytes per pair of capacity, forward and backward together, on a 288 GB device); idle
 *                   workgroups of oversized launches and clears sized for a capacity are not: learn first_capacity from the counts seen, set pair_capacity well above.
 *   count_host      (optional) 8 bytes of PINNED host memory the device can address (hipHostMalloc / hipHostRegister; torch's pin_memory()): the kernel that finds the pair
 *                   count stores {bits of status[0] known at that point (0 and 2), count} there itself, ONE 64-bit store, while emission, tile sort and compositing of the
 *                   call are still to run.  A caller that presets the second word to 0xFFFFFFFF (never a count) and waits for it to change (c3d_gs_wait_count) knows before it
 *                   hands the image on whether the view fitted -- and if not, renders the second half again at the exact count (c3d_gs_forward_render with
 *                   C3D_GS_FLAG_KEEP_RECORD_BASES): always exact, with the GPU never idle behind the host.  The Python boundary does exactly that.
 * status (DEVICE, two words; the call clears them): [0] bit 0 = the view needed more pairs than pair_capacity -- without count_host (a caller that cannot know
 * before it hands the image on) out_color / out_depth / out_alpha are then filled with NaN, never an image that merely looks plausible; radii stay valid; bit 1 = a bounded inter-workgroup wait timed out (device fault); bit 2 = the count exceeded first_capacity
 * (results exact; raise the hint).  [1] = the pair count.  status_host (optional, pinned host memory, two words): the call ends with an asynchronous copy of the two words
 * there, in stream order behind its last kernel (a caller that never waits looks at them one call late: the fault bit can be raised by any kernel of the chain).
 * N > 0 and a non-empty image only. */
int c3d_gs_forward_nosync(const c3d_gs_settings* st, int32_t N, int32_t M, const float* means3D, const float* shs, const float* colors_precomp,
                          const float* opacities, const float* scales, const float* rotations, const float* cov3D_precomp, int32_t* radii,
                          void* geom_buffer, int64_t pair_capacity, int64_t first_capacity, void* binning_buffer, void* image_buffer, float* out_color,
                          float* out_depth, float* out_alpha, uint32_t* status /* device [2] */, uint32_t* status_host /* pinned host [2] or NULL */,
                          uint32_t* count_host /* pinned, device-mapped host [2] or NULL */, c3d_stream_t stream);
/* host side of count_host: returns once its second word differs from `sentinel` (0) -- *bits = first word, *count = second --, or after timeout_us microseconds (-4; < 0: never).
 * Spins on the calling thread, touches no HIP state. */
int c3d_gs_wait_count(const uint32_t* count_host, uint32_t sentinel, int64_t timeout_us, uint32_t* bits /* host, may be NULL */, uint32_t* count /* host, may be NULL */);

/* Backward (A7 + A8).  Pixel gradients dL_dcolor[3,H,W], dL_ddepth[1,H,W] (may be NULL),
 * dL_dalpha[1,H,W] (may be NULL).  Outputs (all written in full by the library, no pre-zeroing needed):
 * dL_dmeans2D[N,3] dL_dcolors[N,3] dL_dopacity[N,1] dL_dmeans3D[N,3] dL_dcov3D[N,6] dL_dsh[N,M,3]
 * dL_dscales[N,3] dL_drotations[N,4].  num_rendered: the pair count the state buffers were sized for (the exact count of c3d_gs_forward_project, or the
 * pair_capacity of c3d_gs_forward_nosync); no record beyond it is read or written.  scratch: c3d_gs_backward_scratch_bytes(N, num_rendered) bytes (one 48-byte gradient
 * record per (tile, splat) pair + one "record written" byte per pair; the backward pass uses no atomics and is bit-reproducible). */
int c3d_gs_backward(const c3d_gs_settings* st, int32_t N, int32_t M, const float* means3D, const float* shs,
                    const float* colors_precomp, const float* scales, const float* rotations,
                    const float* cov3D_precomp, const int32_t* radii, const void* geom_buffer, int64_t num_rendered,
                    const void* binning_buffer, const void* image_buffer, const float* dL_dcolor,
                    const float* dL_ddepth, const float* dL_dalpha, float* dL_dmeans2D, float* dL_dcolors,
                    float* dL_dopacity, float* dL_dmeans3D, float* dL_dcov3D, float* dL_dsh, float* dL_dscales,
                    float* dL_drotations, void* scratch, c3d_stream_t stream);

/* ---- fused-activation variants (extension) -------------------------------------------------------------
 * The reference applies exp / sigmoid / normalize and concatenates f_dc with f_rest in separate torch ops on every render
 * (GaussianModel accessors, main_3DGS_renderer.py:294-321, called from render() :866-868) and back-propagates through them.
 * These two entry points take the RAW parameters of GaussianModel (f_dc [N,1,3], f_rest [N,K-1,3], K = settings.sh_coeffs: 16 by default) and fold
 * those activations and their backward passes into the projection kernels; outputs and state buffers are exactly those of the
 * plain entry points, c3d_gs_forward_render is shared.  `accumulate` != 0 adds the parameter gradients into the given buffers
 * (loops over views) instead of overwriting them; dL_dmeans2D is always overwritten. */
int c3d_gs_forward_project_raw(const c3d_gs_settings* st, int32_t N, const float* means3D, const float* f_dc, const float* f_rest,
                               const float* opacity_raw, const float* scaling_raw, const float* rotation_raw, int32_t* radii,
                               void* geom_buffer, int64_t* num_rendered /* host */, c3d_stream_t stream);
/* c3d_gs_forward_nosync for the raw parameters */
int c3d_gs_forward_raw_nosync(const c3d_gs_settings* st, int32_t N, const float* means3D, const float* f_dc, const float* f_rest,
                              const float* opacity_raw, const float* scaling_raw, const float* rotation_raw, int32_t* radii, void* geom_buffer,
                              int64_t pair_capacity, int64_t first_capacity, void* binning_buffer, void* image_buffer, float* out_color, float* out_depth,
                              float* out_alpha, uint32_t* status /* device [2] */, uint32_t* status_host /* pinned host [2] or NULL */,
                              uint32_t* count_host /* pinned, device-mapped host [2] or NULL */, c3d_stream_t stream);
int c3d_gs_backward_raw(const c3d_gs_settings* st, int32_t N, const float* means3D, const float* f_dc, const float* f_rest,
                        const float* scaling_raw, const float* rotation_raw, const int32_t* radii, const void* geom_buffer,
                        int64_t num_rendered, const void* binning_buffer, const void* image_buffer, const float* dL_dcolor,
                        const float* dL_ddepth, const float* dL_dalpha, float* dL_dmeans2D, float* dL_dmeans3D, float* dL_df_dc,
                        float* dL_df_rest, float* dL_dopacity_raw, float* dL_dscaling_raw, float* dL_drotation_raw, void* scratch,
                        int32_t accumulate, c3d_stream_t stream);

/* ---- fused multi-view training step (extension) ----------------------------------------------------
 * One call = for each of V views: project -> bin -> composite -> pixel loss -> backward, the parameter gradients ADDED into the
 * caller's buffers (zero them before the first call of a step).  Replaces the per-view Python loop of
 * GaussianSplatting3D.training (main_3DGS.py:158-207): the whole loss of main_3DGS.py:184-192 (L1, alpha MSE, MS-SSIM) is inside.
 * Nothing in the call synchronises with the host: the data-dependent number of (tile, splat) pairs stays on the device and every
 * launch is sized for `pair_capacity`.  status[0] becomes non-zero if a view needed more pairs than that (results are then
 * invalid: enlarge and redo the step; bit 0 = pair overflow, bit 1 = a bounded inter-workgroup wait of the binning stage timed
 * out, which indicates a device fault), status[1] holds the largest pair count seen.  loss_out (device float) accumulates the value
 *   scale * sum_v [ w_l1 mean|clamp(C_v,0,1) - Ct_v| + w_l2 mean(clamp(C_v,0,1) - Ct_v)^2 + w_alpha_mse mean(A_v - At_v)^2
 *                   + w_ssim (1 - MS_SSIM(Ct_v, clamp(C_v,0,1))) ]      (w_ssim != 0: include/c3d_loss.h, image sides > 160; masked like the other colour terms).
 * target_color / target_alpha / color_mask: HOST arrays of V device pointers ([3,H,W] / [1,H,W] / [1,H,W]); target_alpha and color_mask may be
 * NULL.  With color_mask the colour terms compare (C * mask) with (Ct * mask), the reference's masked loss (main_3DGS.py:169-186). */
typedef struct c3d_gs_loss { float w_l1; float w_l2; float w_alpha_mse; float scale; float w_ssim; } c3d_gs_loss;
/* The workspace holds one slice per view (state of every view stays alive until the single per-Gaussian backward pass at the end), followed -- when
 * loss->w_ssim != 0 -- by one MS-SSIM share per view; c3d_gs_step_workspace_bytes counts both.
 * lanes (1..8) = the number of view GROUPS the V views are split into (ABI 400): G = ceil(V / lanes) views, at most 16, go through every stage of the chain
 * together -- ONE launch per stage with the view as a grid dimension (projection with the parameters read once for the group, one clear of all state blocks, scan,
 * depth sort, scan, emit, tile sort, ranges, compositing forward, pixel loss + compositing backward, loss sums) -- and the groups follow each other on the
 * caller's `stream`.  The library creates no streams or events of its own; lanes = 1 (all views of a step in one group) is what bench.py and the trainer use.
 * After the last group ONE pass walks the Gaussians, sums each view's (tile, splat) gradient records in a fixed order and writes every parameter gradient
 * once: results are bit-reproducible and independent of `lanes`.
 * accumulate bit 0: add to the contents of the gradient buffers; clear: overwrite them (no zero-fill needed).
 * accumulate bit 1 (value 2): stop after the per-view passes -- the per-(tile, splat) records of all views are then in the workspace and the caller
 * runs the per-Gaussian pass itself with c3d_gs_step_param_backward_range, one Gaussian range after the other, e.g. to start the gradient
 * exchange of a range (multi-GPU) while the next range is still being computed.  The ranges together must cover [0, N) once; each
 * starts at a multiple of 4.  Results are bit-identical to the unchunked call. */
/* status_host (optional, ABI 600): 8 bytes of PINNED host memory the device can address.  The launch that adds up the step's loss -- ordered behind every kernel that can raise
 * a status bit, in front of the per-Gaussian pass -- stores the two status words there itself (one 64-bit store; without loss_out: a copy at the end of the call): a training loop of
 * small scenes, whose iteration is three dozen launches of 4-80 us, learns how the step went without a copy launch in the stream and before the step's last kernels have run
 * (preset the second word to 0xFFFFFFFF and wait for it to change: c3d_gs_wait_count). */
size_t c3d_gs_step_workspace_bytes(int32_t N, int32_t image_height, int32_t image_width, int64_t pair_capacity, int32_t views);
int c3d_gs_train_views_raw(const c3d_gs_settings* views /* host [V] */, int32_t V, int32_t N, const float* means3D, const float* f_dc,
                           const float* f_rest, const float* opacity_raw, const float* scaling_raw, const float* rotation_raw,
                           const float* const* target_color, const float* const* target_alpha, const float* const* color_mask,
                           const c3d_gs_loss* loss,
                           float* dL_dmeans3D, float* dL_df_dc, float* dL_df_rest, float* dL_dopacity_raw, float* dL_dscaling_raw,
                           float* dL_drotation_raw, float* loss_out, int64_t pair_capacity, int32_t lanes, int32_t accumulate, void* workspace,
                           uint32_t* status /* device [2] */, uint32_t* status_host /* pinned, device-mapped host [2] or NULL */, c3d_stream_t stream);

int c3d_gs_step_param_backward_range(const c3d_gs_settings* views /* host [V] */, int32_t V, int32_t N, const float* means3D, const float* f_dc,
                                     const float* f_rest, const float* scaling_raw, const float* rotation_raw, float* dL_dmeans3D, float* dL_df_dc,
                                     float* dL_df_rest, float* dL_dopacity_raw, float* dL_dscaling_raw, float* dL_drotation_raw, int64_t pair_capacity,
                                     int32_t accumulate, void* workspace, int32_t first, int32_t count, c3d_stream_t stream);

/* Forward only, V views of the same cloud in one call (orbit rendering of a trained model: the per-camera loop of the reference's
 * orbit-renderer node over GaussianSplattingRenderer.render, main_3DGS_renderer.py:927-936), raw parameters, no host synchronisation, everything on the
 * caller's `stream`.  HOST arrays of V device pointers: out_color [3,H,W], out_depth [1,H,W], out_alpha [1,H,W]; out_radii (array or entries may be NULL) [N] int32.
 * The workspace holds FORWARD-ONLY slices (two fifths of a training slice: no gradient records / loss buffers; c3d_gs_render_workspace_bytes(N, H, W,
 * pair_capacity, slices)), and workspace_bytes says how many: the views go through the chain in groups of min(ceil(V / lanes), slices, 16) views -- one launch
 * per stage and group, the parameters read once per group -- and every group reuses the same slices (stream order keeps that safe).  So the slice count, not
 * `lanes`, bounds the group width: for a 64-camera orbit pass lanes = 4 and 16 slices (4.5 GB at the BASELINE size); one slice renders view by view; less than
 * one slice is an error.  status as for c3d_gs_train_views_raw: on overflow the images of the affected views are incomplete. */
size_t c3d_gs_render_workspace_bytes(int32_t N, int32_t image_height, int32_t image_width, int64_t pair_capacity, int32_t slices);
int c3d_gs_render_views_raw(const c3d_gs_settings* views /* host [V] */, int32_t V, int32_t N, const float* means3D, const float* f_dc,
                            const float* f_rest, const float* opacity_raw, const float* scaling_raw, const float* rotation_raw,
                            float* const* out_color, float* const* out_depth, float* const* out_alpha, int32_t* const* out_radii,
                            int64_t pair_capacity, int32_t lanes, void* workspace, int64_t workspace_bytes, uint32_t* status /* device [2] */,
                            c3d_stream_t stream);

/* ---- the same step split at the image (round 2): any loss torch can differentiate --------------------------------------------
 * The reference's default loss adds 0.2 * (1 - MS-SSIM) to the L1 / alpha-MSE terms and draws a white or black background per view
 * (main_3DGS.py:184-192, camera_utils.py:246-249; node defaults nodes.py:1177,1181).  Any loss torch can differentiate (a perceptual term, a different MS-SSIM) stays a torch op; so that the
 * rasterizer side of such a step is still ONE sync-free call per direction, the fused step is also offered in two halves:
 *   c3d_gs_forward_views_raw   = c3d_gs_render_views_raw, but view v keeps its state in workspace slice v
 *                                (c3d_gs_step_workspace_bytes(N, H, W, pair_capacity, V)); every view has its own settings, incl. bg;
 *                                groups of ceil(V / lanes) <= 16 views per launch as in c3d_gs_train_views_raw
 *   c3d_gs_backward_views_raw  = the backward half of c3d_gs_train_views_raw for caller-supplied image gradients: HOST arrays of V
 *                                device pointers dL_dcolor [3,H,W] (w.r.t. the UNclamped colour output), dL_ddepth [1,H,W] and
 *                                dL_dalpha [1,H,W] (either array, or entries of it, may be NULL = zero).  Must follow a
 *                                c3d_gs_forward_views_raw call with the same views / N / pair_capacity / workspace whose status was clean, with
 *                                no other call on those workspace slices in between (it may be repeated: every call clears the "record written" bytes itself).
 * c3d_gs_step_read_view serves both.  out_depth (array or entries) may be NULL here. */
int c3d_gs_forward_views_raw(const c3d_gs_settings* views /* host [V] */, int32_t V, int32_t N, const float* means3D, const float* f_dc,
                             const float* f_rest, const float* opacity_raw, const float* scaling_raw, const float* rotation_raw,
                             float* const* out_color, float* const* out_depth, float* const* out_alpha, int32_t* const* out_radii,
                             int64_t pair_capacity, int32_t lanes, void* workspace, uint32_t* status /* device [2] */, c3d_stream_t stream);
int c3d_gs_backward_views_raw(const c3d_gs_settings* views /* host [V] */, int32_t V, int32_t N, const float* means3D, const float* f_dc,
                              const float* f_rest, const float* scaling_raw, const float* rotation_raw, const float* const* dL_dcolor,
                              const float* const* dL_ddepth, const float* const* dL_dalpha, float* dL_dmeans3D, float* dL_df_dc,
                              float* dL_df_rest, float* dL_dopacity_raw, float* dL_dscaling_raw, float* dL_drotation_raw,
                              int64_t pair_capacity, int32_t lanes, int32_t accumulate, void* workspace, c3d_stream_t stream);

/* Per-view by-products of the last c3d_gs_train_views_raw / c3d_gs_backward_views_raw call, copied out of its workspace (same N / H / W / pair_capacity): radii [N] (int32)
 * and the screen-space positional gradient dL/dmeans2D [N,3] of view `view` -- the densification statistics of the reference's trainer
 * (main_3DGS.py:210-213, main_3DGS_renderer.py:767-769).  Either output may be NULL. */
int c3d_gs_step_read_view(int32_t N, int32_t image_height, int32_t image_width, int64_t pair_capacity, const void* workspace, int32_t view,
                          int32_t* radii_out, float* dL_dmeans2D_out, c3d_stream_t stream);

/* The densification statistics of the reference's trainer (main_3DGS.py:210-213 -> GaussianModel.add_densification_stats, main_3DGS_renderer.py:767-769, and the
 * max_radii2D update) accumulated straight from view `view` of the last c3d_gs_train_views_raw / c3d_gs_backward_views_raw call (same N / H / W / pair_capacity / workspace):
 * for every Gaussian the view saw (radius > 0):  grad_accum[i] += ||dL/dmeans2D[i, 0:2]||,  denom[i] += 1,  max_radii[i] = max(max_radii[i], radius) (max_radii may be
 * NULL).  grad_accum, denom: [N,1] float32; max_radii: [N] float32.  One launch instead of c3d_gs_step_read_view's two copies plus the torch ops on them. */
int c3d_gs_step_accumulate_densify_stats(int32_t N, int32_t image_height, int32_t image_width, int64_t pair_capacity, const void* workspace, int32_t view,
                                         float* grad_accum, float* denom, float* max_radii, c3d_stream_t stream);

/* mark_visible: present[N] (uint8) = view-space z > 0.2 */
int c3d_gs_mark_visible(int32_t N, const float* means3D, const float* viewmatrix, const float* projmatrix,
                        uint8_t* present, c3d_stream_t stream);

/* introspection used by the parity tests: copies of internal state, all DEVICE pointers, any may be NULL.
 * point_list[num_rendered] (Gaussian id per sorted pair), ranges[tiles*2], xy[N*2], depths[N],
 * conic_opacity[N*4], rgb[N*3], tiles_touched[N]. */
int c3d_gs_debug_state(int32_t N, int32_t image_height, int32_t image_width, const void* geom_buffer,
                       int64_t num_rendered, const void* binning_buffer, uint32_t* point_list, uint32_t* ranges,
                       float* xy, float* depths, float* conic_opacity, float* rgb, uint32_t* tiles_touched,
                       c3d_stream_t stream);

/* The ONLY process-wide state of the library (tests/test_abi.py lists it): the debug / measurement hooks below -- c3d_prof_enable + c3d_prof_select (which kernel
 * groups get a pair of HIP events around them) and c3d_test_sort_phases (phase stamps of the radix sort).  They change what is MEASURED, never what is computed or
 * which kernels run; off by default, off costs one predictable branch per launch.  A host that embeds the library next to other users of it should leave them alone or
 * serialise its measurements: the switch is not per stream or per context.
 * Optional per-kernel timing: HIP events recorded on the launch stream around each kernel group.
 * c3d_prof_enable(1) resets and starts, c3d_prof_read(slot, &ms, &n) synchronises the recorded events and returns the
 * accumulated milliseconds / launches of a slot; slot names via c3d_prof_name (e.g. "gs_composite_bwd"). */
int c3d_prof_enable(int on);
/* time only the slots whose bit is set in `mask` (all bits = everything, the default).  Each timed launch costs two event records on its
 * stream, which perturbs a multi-stream schedule by a few percent when every launch of a step is timed. */
int c3d_prof_select(unsigned long long mask);
int c3d_prof_slots(void);
const char* c3d_prof_name(int slot);
int c3d_prof_read(int slot, double* total_ms, long long* launches);

/* primitives exported for unit tests of the binning machinery (device pointers) */
int c3d_test_scan_u32(const uint32_t* in, uint32_t* out, int64_t n, int32_t exclusive, c3d_stream_t stream);
int c3d_test_sort_pairs_u32(uint32_t* keys, uint32_t* vals, int64_t n, int32_t end_bit, c3d_stream_t stream);
/* the same with vals = the element indices on entry (not read): keys sorted in place, vals = the permutation.  Up to 16384 keys: one launch of one workgroup (round 6) */
int c3d_test_sort_iota_u32(uint32_t* keys, uint32_t* vals, int64_t n, int32_t end_bit, c3d_stream_t stream);
/* the record-base scan of the backward pass (it normally rides inside the recording forward compositing launch) as a launch of its own: out[n] = exclusive prefix of
 * in[n]; einfo[n][4] = {0, rect[i][0], rect[i][1], out[i]} where in[i] != 0 (other entries untouched); rect [n][2] */
int c3d_test_scan_wave(const uint32_t* in, const uint32_t* rect, uint32_t* out, uint32_t* einfo, int64_t n, c3d_stream_t stream);
/* profiling hook of the radix sort: stamps (device, [passes][tiles][8] uint64, or NULL = off) receive wall_clock64 ticks per phase */
int c3d_test_sort_phases(uint64_t* stamps);

#ifdef __cplusplus
}
#endif
#endif /* C3D_GS_H */
