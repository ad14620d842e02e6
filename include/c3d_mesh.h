/*
 * c3d_mesh.h -- C-ABI of the MI355X-native differentiable triangle-mesh ops (libc3d_hip.so).
 *
 * Drop-in boundary: what a native replacement of `nvdiffrast.torch` must provide for the calls the reference makes at
 *   /root/reference/MVs_Algorithms/DiffRastMesh/diff_mesh_renderer.py:46     dr.RasterizeCudaContext()
 *   :97    dr.rasterize(glctx, v_clip[1,V,4], f[T,3], (h, w))            -> c3d_mesh_rasterize_fwd / _bwd
 *   :101   dr.antialias(alpha[1,H,W,1], rast, v_clip, f)                  -> c3d_mesh_antialias_fwd / _bwd
 *   :104   dr.interpolate(vt[1,V,2], rast, ft, rast_db, diff_attrs='all') -> c3d_mesh_interpolate_fwd / _bwd
 *   :105   dr.texture(raw_albedo[1,Ht,Wt,3], texc, uv_da=texc_db, filter_mode='linear') -> c3d_mesh_texture_fwd / _bwd
 *   :110,131 dr.interpolate(depth / normals ...)   :138 dr.antialias(albedo[1,H,W,3], ...)
 * (also Gen_3D_Modules users of the same ops, and MVs_Algorithms/FlexiCubes/flexicubes_renderer.py:46-66).
 *
 * Conventions (identical to the dependency's): clip-space positions [B,V,4]; triangles int32 [T,3]; images NHWC fp32;
 * pixel (x,y) centre at NDC ((x+.5)2/W-1, (y+.5)2/H-1), row 0 at NDC y=-1; rast = (u, v, z/w, triangle_id+1) with 0 =
 * background; rast_db = (du/dX, du/dY, dv/dX, dv/dY).  All pointers are DEVICE pointers to contiguous arrays; every call is
 * asynchronous on `stream` and returns 0 or an error code (message via c3d_last_error()).
 * Gradient outputs documented as "accumulated" are zeroed by the library before the kernel runs (callers pass
 * uninitialised memory).  Gradients w.r.t. rast_db / out_da are not propagated (no consumer on the reference's path).
 */
#ifndef C3D_MESH_H
#define C3D_MESH_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif
#ifndef C3D_STREAM_T
#define C3D_STREAM_T
typedef void* c3d_stream_t; /* hipStream_t */
#endif

/* rasterize: scratch = per-pixel 64-bit depth|id buffer + large-triangle queue (the "context" the Python object owns) */
size_t c3d_mesh_raster_scratch_bytes(int32_t B, int32_t H, int32_t W, int32_t T);
int c3d_mesh_rasterize_fwd(const float* pos, const int32_t* tri, int32_t B, int32_t V, int32_t T, int32_t H, int32_t W,
                           void* scratch, float* rast, float* rast_db, c3d_stream_t stream);
/* Depth peeling (the dependency's DepthPeeler.rasterize_next_layer; reference use: Gen_3D_Modules/InstantMesh/models/geometry/render/
 * neural_render.py:103-106): like c3d_mesh_rasterize_fwd, but a fragment is kept only where the previous layer had a surface and only
 * if its z/w is strictly greater than that surface's.  prev_scratch = the scratch buffer the previous layer was rasterized with
 * (untouched since); NULL = first layer = c3d_mesh_rasterize_fwd.  prev_scratch and scratch must be different buffers. */
int c3d_mesh_rasterize_peel_fwd(const float* pos, const int32_t* tri, int32_t B, int32_t V, int32_t T, int32_t H, int32_t W,
                                const void* prev_scratch, void* scratch, float* rast, float* rast_db, c3d_stream_t stream);
/* dy = dL/drast [B,H,W,4] (only u,v channels are differentiable), ddb = dL/drast_db [B,H,W,4] (the dependency's grad_db path; ABI 301); either may be NULL;
 * dpos [B,V,4] written in full */
int c3d_mesh_rasterize_bwd(const float* pos, const int32_t* tri, const float* rast, const float* dy, const float* ddb, int32_t B, int32_t V,
                           int32_t T, int32_t H, int32_t W, float* dpos, c3d_stream_t stream);

/* Atomic-free, bit-reproducible rasterize backward.  topology = vertex -> (triangle, corner) adjacency of `tri`
 * (c3d_mesh_build_vertex_topology; valid while `tri` is unchanged, like the antialias edge hash); scratch holds one 16-byte
 * record per (triangle, corner) and a queue for large triangles.  A lane per triangle re-walks its bounding box, keeps the
 * pixels it owns and sums its corners' gradients in registers; a lane per vertex then sums its corners' records in a fixed
 * order.  Same result as c3d_mesh_rasterize_bwd up to summation order; dpos is written in full (no zero-fill needed). */
size_t c3d_mesh_vertex_topology_bytes(int32_t V, int32_t T);
int c3d_mesh_build_vertex_topology(const int32_t* tri, int32_t V, int32_t T, void* topology, c3d_stream_t stream);
size_t c3d_mesh_rasterize_bwd_scratch_bytes(int32_t B, int32_t T);
int c3d_mesh_rasterize_bwd_gather(const float* pos, const int32_t* tri, const float* rast, const float* dy, const float* ddb, int32_t B, int32_t V,
                                  int32_t T, int32_t H, int32_t W, const void* topology, void* scratch, float* dpos,
                                  c3d_stream_t stream);

/* interpolate: attr [Ba,V,A], Ba in {1,B}; diff = nd attribute indices for which pixel differentials are produced
 * (out_da [B,H,W,2*nd], pairs (d/dX, d/dY)); nd = 0 -> rast_db/diff/out_da may be NULL */
int c3d_mesh_interpolate_fwd(const float* attr, int32_t Ba, const float* rast, const int32_t* tri, const float* rast_db,
                             const int32_t* diff, int32_t nd, int32_t B, int32_t V, int32_t A, int32_t H, int32_t W,
                             float* out, float* out_da, c3d_stream_t stream);
/* dattr [Ba,V,A] accumulated, or NULL when the attribute gradient is not wanted (constant attributes such as texture coordinates: no vertex
 * scatter at all); drast [B,H,W,4] written in full (channels 2,3 = 0) */
int c3d_mesh_interpolate_bwd(const float* attr, int32_t Ba, const float* rast, const int32_t* tri, const float* dy,
                             int32_t B, int32_t V, int32_t A, int32_t H, int32_t W, float* dattr, float* drast,
                             c3d_stream_t stream);

/* backward of interpolate's pixel differentials out_da (what a mip-mapped texture() hands back as d uv_da): dout_da [B,H,W,2 n_diff] ->
 * dattr [Ba,V,A] ADDED to (call after c3d_mesh_interpolate_bwd; NULL = not wanted), drast_db [B,H,W,4] written in full: hand it to
 * c3d_mesh_rasterize_bwd as `ddb` (the dependency's grad_db path, ABI 301) to carry it on to the vertex positions. */
int c3d_mesh_interpolate_da_bwd(const float* attr, int32_t Ba, const float* rast, const int32_t* tri, const float* rast_db,
                                const int32_t* diff_attrs, int32_t n_diff, const float* dout_da, int32_t B, int32_t V, int32_t A,
                                int32_t H, int32_t W, float* dattr, float* drast_db, c3d_stream_t stream);

/* texture: tex [Bt,Ht,Wt,C], Bt in {1,B}; uv [B,H,W,2]; filter 0 = nearest, 1 = linear; boundary 0 = wrap, 1 = clamp */
int c3d_mesh_texture_fwd(const float* tex, int32_t Bt, const float* uv, int32_t B, int32_t H, int32_t W, int32_t Ht,
                         int32_t Wt, int32_t C, int32_t filter, int32_t boundary, float* out, c3d_stream_t stream);
/* dtex accumulated; duv [B,H,W,2] written in full */
int c3d_mesh_texture_bwd(const float* tex, int32_t Bt, const float* uv, const float* dy, int32_t B, int32_t H, int32_t W,
                         int32_t Ht, int32_t Wt, int32_t C, int32_t filter, int32_t boundary, float* dtex, float* duv,
                         c3d_stream_t stream);

/* Mip-mapped texture: the dependency's filter modes 'linear-mipmap-nearest' (filter 2) and 'linear-mipmap-linear' (filter 3), i.e. what
 * dr.texture(tex, uv, uv_da) selects under filter_mode='auto' (reference call sites: Gen_3D_Modules/LGM/nerf_marching_cubes_converter.py:232,
 * Gen_3D_Modules/TRELLIS/trellis/utils/postprocessing_utils.py:384, Gen_3D_Modules/Stable3DGen/trellis/utils/_rasterization.py:88) and
 * dr.texture_construct_mip.
 *   pyramid: level l+1 = 2x2 box average of level l (an extent that reached 1 stays 1); levels until 1x1, max_mip_level (< 0: no limit) or 16.
 *            `stack` [Bt][sum_{l>=1} h_l*w_l][C] holds levels 1..L back to back.  A pyramid that would have to halve an odd extent > 1 is an error.
 *   level  : uv_da [B,H,W,4] = (du/dX, du/dY, dv/dX, dv/dY) -> log2 of the footprint's major axis in texels, + mip_level_bias [B,H,W] (either may
 *            be NULL, not both), clamped to [0, L]; filter 3 blends floor(level) and the next level, filter 2 samples floor(level + .5).
 * c3d_mesh_mip_info is host-only: returns L (or -1), levels_hw[2*(L+1)] = (h, w) per level incl. the base (room for 34 ints), *stack_texels. */
int32_t c3d_mesh_mip_info(int32_t Ht, int32_t Wt, int32_t max_mip_level, int32_t* levels_hw, int64_t* stack_texels);
int c3d_mesh_mip_build(const float* tex, int32_t Bt, int32_t Ht, int32_t Wt, int32_t C, int32_t max_mip_level, float* stack,
                       c3d_stream_t stream);
/* dtex [Bt,Ht,Wt,C] written in full = transpose of the pyramid applied to dstack; dstack is used as scratch and modified */
int c3d_mesh_mip_build_bwd(float* dstack, int32_t Bt, int32_t Ht, int32_t Wt, int32_t C, int32_t max_mip_level, float* dtex,
                           c3d_stream_t stream);
int c3d_mesh_texture_mip_fwd(const float* tex, const float* stack, int32_t Bt, const float* uv, const float* uv_da,
                             const float* mip_level_bias, int32_t B, int32_t H, int32_t W, int32_t Ht, int32_t Wt, int32_t C,
                             int32_t filter, int32_t boundary, int32_t max_mip_level, float* out, c3d_stream_t stream);
/* dtex (level-0 taps) and dstack (levels >= 1) accumulated; duv [B,H,W,2] written in full.  d_uv_da [B,H,W,4] / d_bias [B,H,W] (each may be NULL;
 * written in full): gradients w.r.t. the pixel differentials and the level bias -- 'linear-mipmap-linear' blends two levels by the fraction of the
 * level, so d out / d level = sample(level + 1) - sample(level) where the level is not clamped; zero for 'linear-mipmap-nearest'
 * (the dependency's texture() propagates both: nvdiffrast/torch/ops.py texture, filter_mode 'linear-mipmap-linear'). */
int c3d_mesh_texture_mip_bwd(const float* tex, const float* stack, int32_t Bt, const float* uv, const float* uv_da,
                             const float* mip_level_bias, const float* dy, int32_t B, int32_t H, int32_t W, int32_t Ht, int32_t Wt,
                             int32_t C, int32_t filter, int32_t boundary, int32_t max_mip_level, float* dtex, float* dstack, float* duv,
                             float* d_uv_da, float* d_bias, c3d_stream_t stream);

/* antialias: scratch = edge hash of the topology.  A pixel pair is analysed on its NEARER triangle (smaller z/w; background counts as farthest); if that triangle has a
 * vertex at or behind the camera plane (w <= 0: a near-plane clipped triangle) the pair is left alone -- colour and gradients pass through unchanged -- and an edge
 * whose neighbour across it has such a vertex is not taken for a silhouette.  (The dependency's analysis kernel is not in the reference tree; this is the rule
 * oracle/mesh_oracle.c restates and tests/test_mesh_oracle.py, tests/test_mesh_hip.py pin: *_leaves_pairs_of_a_near_plane_clipped_triangle_alone.)  c3d_mesh_antialias_build_topology fills it from `tri` (the dependency's
 * antialias_construct_topology_hash); it stays valid for as long as `tri` is unchanged and is shared by forward and backward. */
size_t c3d_mesh_antialias_scratch_bytes(int32_t T);
int c3d_mesh_antialias_build_topology(const int32_t* tri, int32_t T, void* scratch, c3d_stream_t stream);
int c3d_mesh_antialias_fwd(const float* color, const float* rast, const float* pos, const int32_t* tri, int32_t B, int32_t V,
                           int32_t T, int32_t H, int32_t W, int32_t C, const void* scratch, float* out, c3d_stream_t stream);
/* dcolor [B,H,W,C] written in full; dpos [B,V,4] accumulated */
int c3d_mesh_antialias_bwd(const float* color, const float* rast, const float* pos, const int32_t* tri, const float* dy,
                           int32_t B, int32_t V, int32_t T, int32_t H, int32_t W, int32_t C, const void* scratch,
                           float* dcolor, float* dpos, c3d_stream_t stream);

/* Renderer glue of DiffRastRenderer.render (/root/reference/MVs_Algorithms/DiffRastMesh/diff_mesh_renderer.py:94-96 and :139-151): the
 * elementwise torch chain around the ops above, as one kernel each way.
 * transform: out[V,4] = [v, 1] . M^T for a row-major 4x4 M on the DEVICE (the reference pads v and runs two GEMMs: inverse(pose), then proj);
 * shade:     alpha_out = clamp(alpha, 0, 1); image = clamp(alpha_out * albedo + (1 - alpha_out) * bg, 0, 1)  ([P,3], [P], bg[3] on the device);
 *            backward with torch.clamp's convention (gradient passes on the closed interval); dimage / dalpha_out may be NULL (= zero). */
int c3d_mesh_transform_fwd(const float* v, const float* M, int32_t V, float* out, c3d_stream_t stream);
int c3d_mesh_transform_bwd(const float* M, const float* dout, int32_t V, float* dv, c3d_stream_t stream);
int c3d_mesh_shade_fwd(const float* albedo, const float* alpha, const float* bg, int64_t P, float* image, float* alpha_out, c3d_stream_t stream);
int c3d_mesh_shade_bwd(const float* albedo, const float* alpha, const float* bg, int64_t P, const float* dimage, const float* dalpha_out,
                       float* dalbedo, float* dalpha, c3d_stream_t stream);


/* ---- one view of DiffRastRenderer.render as ONE call each way (round 2) ------------------------------------------------------
 * The op sequence of /root/reference/MVs_Algorithms/DiffRastMesh/diff_mesh_renderer.py:94-151 for ssaa = 1:
 *   v_clip = [v (+ v_offsets), 1] . clip_from_world^T;  rast = rasterize(v_clip, f);  alpha = antialias(clamp(rast.w, 0, 1));
 *   albedo = antialias(sigmoid(texture(raw_albedo, interpolate(vt, rast, ft), 'linear')));
 *   a = clamp(alpha, 0, 1);  image = clamp(a * albedo + (1 - a) * bg, 0, 1)            -> image [H,W,3], alpha [H,W,1] (= a)
 * enqueued from C without returning to the host language, camera matrix and background as plain struct fields (no upload), and the two
 * antialias calls sharing one silhouette analysis per pixel pair.  Same values as the op-by-op path (tests/test_mesh_hip.py).
 *   state   : c3d_mesh_view_state_bytes(V, T, H, W) bytes written by _fwd and read by _bwd (v_clip, rast, uv, albedo, ..., one bit per triangle: owns a pixel); the caller
 *             may read rast ([H,W,4] floats at byte offset align256(16 V)) for the depth / normal outputs the reference produces on demand
 *   aa_topology / vertex_topology : c3d_mesh_antialias_build_topology / c3d_mesh_build_vertex_topology of `f`
 *   _bwd    : dimage [H,W,3], dalpha [H,W,1] (either may be NULL) -> d_raw_albedo [Ht,Wt,3] WRITTEN IN FULL, d_v [V,3] written in full
 *             (NULL: geometry not trained; vertex_topology may then be NULL).  scratch: c3d_mesh_view_bwd_scratch_bytes(V, T, H, W, Ht, Wt).
 *             Round 3 (ABI 301): no float atomics anywhere on this path -- the antialias blends, their colour and position gradients are gathers over a
 *             pixel's four pairs, texel gradients are added as 64-bit integers (two planes of Ht Wt 3 words in `scratch`): same bits every run. */
typedef struct c3d_mesh_view {
    int32_t V, T, Vt, H, W, Ht, Wt;
    float clip_from_world[16];      /* row-major 4x4 */
    float bg[3];
} c3d_mesh_view;
size_t c3d_mesh_view_state_bytes(int32_t V, int32_t T, int32_t H, int32_t W);
size_t c3d_mesh_view_bwd_scratch_bytes(int32_t V, int32_t T, int32_t H, int32_t W, int32_t Ht, int32_t Wt);
int c3d_mesh_view_fwd(const c3d_mesh_view* d, const float* v, const float* v_offsets, const int32_t* f, const float* vt, const int32_t* ft,
                      const float* raw_albedo, const void* aa_topology, void* raster_scratch, void* state, float* image, float* alpha,
                      c3d_stream_t stream);
int c3d_mesh_view_bwd(const c3d_mesh_view* d, const float* v, const float* v_offsets, const int32_t* f, const float* vt, const int32_t* ft,
                      const float* raw_albedo, const void* aa_topology, const void* vertex_topology, void* scratch, const void* state,
                      const float* dimage, const float* dalpha, float* d_raw_albedo, float* d_v, c3d_stream_t stream);

/* ---- fused multi-view training step (extension) ----------------------------------------------------------------------------------------
 * One call = what DiffMesh.training_step does for the views of a step (reference: MVs_Algorithms/DiffRastMesh/diff_mesh.py:98-125): for each view
 * c3d_mesh_view_fwd -> image loss -> backward, the gradients summed over the views.  ABI 400: the views go through every stage TOGETHER (groups of
 * up to 16 views, one launch per stage with the view as a grid dimension, everything on `stream`; nothing synchronises with the host).  `lanes`
 * (1..8; rounds 2-3 dealt the views onto that many library-owned streams) is still checked but no longer changes anything.  Loss, per view v of n:
 *   scale * [ w_mse * mean_{c,p} ((image_v - target_v) m_v)^2  +  w_ssim * (1 - MS_SSIM(target_v m_v, image_v m_v)) ]        (include/c3d_loss.h; sides > 160)
 * with scale = 1 / n this is the reference's batch loss (1 - lambda) F.mse_loss(imgs, refs) + lambda (1 - ms_ssim(refs, imgs)), lambda = w_ssim.
 * target_chw / mask: HOST arrays of n DEVICE pointers, [3,H,W] / [1,H,W] (mask or its entries may be NULL = 1).  loss_out (device float) accumulates
 * the value.  d_raw_albedo [Ht,Wt,3], d_v_offsets [V,3] (NULL: geometry not trained): overwritten, or added to when accumulate != 0.  The texture
 * gradient of all views accumulates in one pair of 64-bit integer planes, vertex gradients and loss terms are summed over the views in a fixed
 * order: the same bits every run.
 * workspace: c3d_mesh_step_workspace_bytes(V, T, H, W, Ht, Wt, n_views, lanes).  At most 64 views per call. */
typedef struct c3d_mesh_step_loss { float w_mse; float w_ssim; float scale; } c3d_mesh_step_loss;
size_t c3d_mesh_step_workspace_bytes(int32_t V, int32_t T, int32_t H, int32_t W, int32_t Ht, int32_t Wt, int32_t n_views, int32_t lanes);
int c3d_mesh_train_views(const c3d_mesh_view* views /* host [n_views] */, int32_t n_views, const float* v, const float* v_offsets, const int32_t* f,
                         const float* vt, const int32_t* ft, const float* raw_albedo, const void* aa_topology, const void* vertex_topology,
                         const float* const* target_chw, const float* const* mask, const c3d_mesh_step_loss* loss, float* d_raw_albedo,
                         float* d_v_offsets, float* loss_out, int32_t accumulate, int32_t lanes, void* workspace, c3d_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* C3D_MESH_H */
