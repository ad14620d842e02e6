// gs_api.hip -- extern "C" entry points declared in include/c3d_gs.h.
#include "../../include/c3d_gs.h"
#include "../../include/c3d_loss.h"
#include "gs_internal.h"
#include <mutex>
#include <stdarg.h>
#include <stdio.h>

static thread_local char g_err[1024] = "";
void c3d_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

// msssim.hip: out_word += va + vb * mean MS-SSIM(x, y_eff), dL_dy (+)= grad_scale * d mean / dy
int ms_value_grad(const float* x, const float* y, const float* mask, int clamp_y, int B, int C, int H, int W, float grad_scale, int accumulate, float* dL_dy,
                  float va, float vb, float* ms_out, void* workspace, hipStream_t s, int store_value = 0);
static int tile_sort_bits(int tiles) {
    int bits = 0;
    while ((1ll << bits) < (long long)tiles) bits++;
    return bits < 1 ? 1 : bits;
}
static int sort_result_index(int end_bit) {   // buffer index after ceil(end_bit/8) (>=1) ping-pong passes
    int passes = (end_bit + 7) / 8;
    if (passes < 1) passes = 1;
    return passes & 1;
}

static int make_params(const c3d_gs_settings* st, int N, int M, GsParams& p) {
    if (!st) { c3d_set_error("c3d_gs: settings is NULL"); return -1; }
    if (N < 0 || M < 0 || st->image_height < 0 || st->image_width < 0) { c3d_set_error("c3d_gs: negative size"); return -1; }
    if (st->sh_degree < 0 || st->sh_degree > 3) { c3d_set_error("c3d_gs: sh_degree %d not in [0,3]", st->sh_degree); return -1; }
    if (!st->bg || !st->viewmatrix || !st->projmatrix || !st->campos) { c3d_set_error("c3d_gs: settings device pointers must be non-NULL"); return -1; }
    p.N = N; p.M = M; p.deg = st->sh_degree; p.W = st->image_width; p.H = st->image_height;
    p.gx = (p.W + C3D_TILE_X - 1) / C3D_TILE_X; p.gy = (p.H + C3D_TILE_Y - 1) / C3D_TILE_Y;
    p.tanfovx = st->tanfovx; p.tanfovy = st->tanfovy;
    p.focal_x = p.W / (2.0f * st->tanfovx); p.focal_y = p.H / (2.0f * st->tanfovy);
    p.scale_modifier = st->scale_modifier;
    p.dscale_mod = (st->flags & C3D_GS_FLAG_EXACT_DSCALE) ? st->scale_modifier : 1.0f;
    p.bg = st->bg; p.view = st->viewmatrix; p.proj = st->projmatrix; p.campos = st->campos;
    return 0;
}
static int check_inputs(int N, int M, int deg, const float* means3D, const float* shs, const float* colors_precomp,
                        const float* scales, const float* rotations, const float* cov3D_precomp) {
    if (N == 0) return 0;
    if (!means3D) { c3d_set_error("c3d_gs: means3D is NULL"); return -1; }
    if ((shs == nullptr) == (colors_precomp == nullptr)) { c3d_set_error("Please provide excatly one of either SHs or precomputed colors!"); return -1; }
    if (((scales == nullptr || rotations == nullptr) && cov3D_precomp == nullptr) || ((scales != nullptr || rotations != nullptr) && cov3D_precomp != nullptr)) {
        c3d_set_error("Please provide exactly one of either scale/rotation pair or precomputed 3D covariance!");
        return -1;
    }
    if (shs && (deg + 1) * (deg + 1) > M) { c3d_set_error("c3d_gs: sh_degree %d needs %d coefficients, tensor has %d", deg, (deg + 1) * (deg + 1), M); return -1; }
    return 0;
}

// A2-A4' shared by every forward entry point: record bases (scan in Gaussian-id order), depth sort, emit offsets (scan in depth-rank
// order, gather folded in).  Three single-pass primitives, their state cleared by ONE memset; 7 launches where round 1 issued 21.
// cap / status: pair capacity and status words of the sync-free paths (the pair count then stays on the device in g.meta[0]).
// C3D_BIN_LOCAL = 0 (default): global depth sort of the Gaussians, second scan in depth-rank order, rank-ordered emission, tile sort, ranges.
// 1 (round 3, built and measured): ONE scan (record bases = emission offsets in Gaussian-id order), emission in id order, tile sort, ranges, then the depth order
// per tile (c3d_segment_sort_u32) -- the same lists bit for bit, five latency-bound launches fewer per view.  It loses: ordering AFTER duplication sorts 4.0 M
// (tile, splat) pairs instead of 1.0 M Gaussians, and the ballot ranking costs ~1 VALU instruction per element and pass -- throughput-bound work (0.10 ms per view)
// in exchange for latency-bound work (0.13 + 0.025 ms) that the view lanes were hiding anyway: 5.94 vs 5.88 ms per step (profiles/r03/r03g_*).
static bool gs_bin_local() {
    static int v = -1;
    if (v < 0) { const char* e = getenv("C3D_BIN_LOCAL"); v = e ? atoi(e) != 0 : 0; }
    return v != 0;
}
static int binning_front(GsGeom& g, int N, uint32_t cap, uint32_t* status, hipStream_t s) {
    int rc, res = 0;
    C3D_CHECK(hipMemsetAsync(g.meta, 0, g.zero_bytes, s));
    uint32_t* err = status ? status : (uint32_t*)g.meta + 2;      // a timed-out look-back (bounded spins) surfaces as C3D_ERR_LOOKBACK
    if (gs_bin_local()) {
        C3dProfScope ps(C3D_P_SCAN, s);
        return c3d_scan_u32_einfo(g.tiles, g.rbase, (size_t)N, g.tmp_scan_a, s, false, err, g.rect, g.einfo, (uint32_t*)g.meta, status, cap);
    }
    { C3dProfScope ps(C3D_P_SCAN, s);
      if ((rc = c3d_scan_u32_einfo(g.tiles, g.rbase, (size_t)N, g.tmp_scan_a, s, false, err, g.rect, g.einfo))) return rc; }
    { C3dProfScope ps(C3D_P_DEPTH_SORT, s);
      if ((rc = c3d_sort_pairs_u32(g.key[0], g.key[1], g.order[0], g.order[1], true, (size_t)N, 32, g.tmp_sort, &res, s, nullptr, false, err))) return rc; }
    if (res != sort_result_index(32)) { c3d_set_error("internal: depth sort buffer parity"); return -2; }
    { C3dProfScope ps(C3D_P_SCAN, s);
      if ((rc = c3d_scan_gather_u32(g.tiles, g.order[res], g.offsets, (size_t)N, false, g.tmp_scan_b, s, false, (uint32_t*)g.meta, status, cap, err))) return rc; }
    return 0;
}
// tile sort + per-tile ranges; D = pair count on the host, or the capacity when d_dev (device count) is given
static int binning_back(const GsParams& p, GsGeom& g, GsBinning& b, const int* radii, long long D, uint32_t cap, const uint32_t* d_dev, uint32_t* status,
                        hipStream_t s, int* res_out) {
    const int tiles = p.gx * p.gy;
    int rc, res = 0;
    C3D_CHECK(hipMemsetAsync(b.ranges, 0, b.zero_bytes, s));
    *res_out = 0;
    if (D <= 0) return 0;
    uint32_t* err = status ? status : (uint32_t*)g.meta + 2;
    { C3dProfScope ps(C3D_P_EMIT, s);
      if ((rc = gs_launch_emit(p, g, gs_bin_local() ? -1 : sort_result_index(32), radii, b, s, cap))) return rc; }
    { C3dProfScope ps(C3D_P_TILE_SORT, s);
      if ((rc = c3d_sort_pairs_u32(b.tkey[0], b.tkey[1], b.tval[0], b.tval[1], false, (size_t)D, tile_sort_bits(tiles), b.tmp, &res, s, d_dev, false, err))) return rc; }
    if (res != sort_result_index(tile_sort_bits(tiles))) { c3d_set_error("internal: tile sort buffer parity"); return -2; }
    { C3dProfScope ps(C3D_P_RANGES, s);
      if ((rc = gs_launch_ranges(b, res, D, tiles, s, d_dev))) return rc; }
    if (gs_bin_local()) {      // depth order inside every tile's list (keys: the depth bits k_preprocess left per Gaussian); the idle value buffer is the scratch of long lists
        C3dProfScope ps(C3D_P_DEPTH_SORT, s);
        if ((rc = c3d_segment_sort_u32(b.ranges, tiles, g.key[0], b.tval[res], b.tval[res ^ 1], s))) return rc;
    }
    *res_out = res;
    return 0;
}

// drop-in path: the single D2H read of the pair count (the wheel has the same synchronisation) also brings back the error word
static int project_tail(GsGeom& g, int N, int64_t* num_rendered, hipStream_t s) {
    int rc;
    if ((rc = binning_front(g, N, 0xFFFFFFFFu, nullptr, s))) return rc;
    uint32_t host[4] = {0, 0, 0, 0};
    C3D_CHECK(hipMemcpyAsync(host, g.meta, sizeof(host), hipMemcpyDeviceToHost, s));
    C3D_CHECK(hipStreamSynchronize(s));
    if (host[2] & C3D_ERR_LOOKBACK) { c3d_set_error("c3d_gs: a chained-scan look-back timed out in the binning stage (device fault or a wedged workgroup)"); return -3; }
    *num_rendered = (int64_t)host[0];
    return 0;
}

extern "C" {

const char* c3d_last_error(void) { return g_err; }
int c3d_version(void) { return 301; }

size_t c3d_gs_geom_bytes(int32_t N) { GsGeom g; gs_carve_geom(nullptr, N, g); return g.bytes; }
size_t c3d_gs_binning_bytes(int64_t D, int32_t H, int32_t W) {
    GsBinning b;
    gs_carve_binning(nullptr, D, ((W + C3D_TILE_X - 1) / C3D_TILE_X) * ((H + C3D_TILE_Y - 1) / C3D_TILE_Y), b);
    return b.bytes;
}
size_t c3d_gs_image_bytes(int32_t H, int32_t W) { GsImage im; gs_carve_image(nullptr, W, H, im); return im.bytes; }
static size_t pairgrad_bytes(long long D) { return c3d_align(sizeof(float) * GS_PAIR_FLOATS * (size_t)(D > 0 ? D : 1)); }
// backward scratch = one 48-byte gradient record per (tile, splat) pair, then one "record written" byte per pair
size_t c3d_gs_backward_scratch_bytes(int32_t N, int64_t D) { (void)N; return pairgrad_bytes(D) + c3d_align((size_t)(D > 0 ? D : 1)); }

int c3d_gs_forward_project(const c3d_gs_settings* st, int32_t N, int32_t M, const float* means3D, const float* shs,
                           const float* colors_precomp, const float* opacities, const float* scales,
                           const float* rotations, const float* cov3D_precomp, int32_t* radii, void* geom_buffer,
                           int64_t* num_rendered, c3d_stream_t stream) {
    hipStream_t s = (hipStream_t)stream;
    GsParams p;
    if (make_params(st, N, M, p)) return -1;
    if (check_inputs(N, M, p.deg, means3D, shs, colors_precomp, scales, rotations, cov3D_precomp)) return -1;
    if (!num_rendered) { c3d_set_error("c3d_gs_forward_project: num_rendered (host) is NULL"); return -1; }
    *num_rendered = 0;
    if (N == 0) return 0;
    if (!geom_buffer || !radii || !opacities) { c3d_set_error("c3d_gs_forward_project: NULL buffer"); return -1; }
    GsGeom g;
    gs_carve_geom((char*)geom_buffer, N, g);
    int rc;
    { C3dProfScope ps(C3D_P_PREPROCESS, s);
    if ((rc = gs_launch_preprocess(p, means3D, shs, colors_precomp, opacities, scales, rotations, cov3D_precomp, g, radii, s))) return rc; }
    return project_tail(g, N, num_rendered, s);
}

int c3d_gs_forward_project_raw(const c3d_gs_settings* st, int32_t N, const float* means3D, const float* f_dc, const float* f_rest,
                               const float* opacity_raw, const float* scaling_raw, const float* rotation_raw, int32_t* radii,
                               void* geom_buffer, int64_t* num_rendered, c3d_stream_t stream) {
    hipStream_t s = (hipStream_t)stream;
    GsParams p;
    if (make_params(st, N, 16, p)) return -1;
    if (!num_rendered) { c3d_set_error("c3d_gs_forward_project_raw: num_rendered (host) is NULL"); return -1; }
    *num_rendered = 0;
    if (N == 0) return 0;
    if (!means3D || !f_dc || !f_rest || !opacity_raw || !scaling_raw || !rotation_raw || !radii || !geom_buffer) { c3d_set_error("c3d_gs_forward_project_raw: NULL pointer"); return -1; }
    if ((uintptr_t)f_rest % 16 || (uintptr_t)rotation_raw % 16) { c3d_set_error("c3d_gs_forward_project_raw: f_rest / rotation must be 16-byte aligned"); return -1; }
    GsGeom g;
    gs_carve_geom((char*)geom_buffer, N, g);
    int rc;
    { C3dProfScope ps(C3D_P_PREPROCESS, s);
    if ((rc = gs_launch_preprocess_raw(p, means3D, f_dc, f_rest, opacity_raw, scaling_raw, rotation_raw, g, radii, s))) return rc; }
    return project_tail(g, N, num_rendered, s);
}

int c3d_gs_forward_render(const c3d_gs_settings* st, int32_t N, int32_t M, const int32_t* radii, void* geom_buffer,
                          int64_t num_rendered, void* binning_buffer, void* image_buffer, float* out_color,
                          float* out_depth, float* out_alpha, c3d_stream_t stream) {
    hipStream_t s = (hipStream_t)stream;
    GsParams p;
    if (make_params(st, N, M, p)) return -1;
    const int tiles = p.gx * p.gy;
    if (tiles == 0) return 0;
    if (!binning_buffer || !image_buffer || !out_color || !out_depth || !out_alpha) { c3d_set_error("c3d_gs_forward_render: NULL buffer"); return -1; }
    GsGeom g;
    gs_carve_geom((char*)geom_buffer, N, g);
    GsBinning b;
    gs_carve_binning((char*)binning_buffer, num_rendered, tiles, b);
    GsImage im;
    gs_carve_image((char*)image_buffer, p.W, p.H, im);
    int rc, res = 0;
    if (num_rendered > 0 && (!geom_buffer || !radii)) { c3d_set_error("c3d_gs_forward_render: NULL geometry"); return -1; }
    if ((rc = binning_back(p, g, b, radii, num_rendered, 0xFFFFFFFFu, nullptr, nullptr, s, &res))) return rc;
    C3dProfScope ps(C3D_P_COMPOSITE_FWD, s);
    return gs_launch_composite_fwd(p, g, b, res, im, out_color, out_depth, out_alpha, true, s);   // a backward call may follow: record the blended (quadrant, splat) pairs
}

int c3d_gs_backward(const c3d_gs_settings* st, int32_t N, int32_t M, const float* means3D, const float* shs,
                    const float* colors_precomp, const float* scales, const float* rotations,
                    const float* cov3D_precomp, const int32_t* radii, const void* geom_buffer, int64_t num_rendered,
                    const void* binning_buffer, const void* image_buffer, const float* dL_dcolor,
                    const float* dL_ddepth, const float* dL_dalpha, float* dL_dmeans2D, float* dL_dcolors,
                    float* dL_dopacity, float* dL_dmeans3D, float* dL_dcov3D, float* dL_dsh, float* dL_dscales,
                    float* dL_drotations, void* scratch, c3d_stream_t stream) {
    hipStream_t s = (hipStream_t)stream;
    GsParams p;
    if (make_params(st, N, M, p)) return -1;
    if (check_inputs(N, M, p.deg, means3D, shs, colors_precomp, scales, rotations, cov3D_precomp)) return -1;
    if (N == 0) return 0;
    if (!dL_dcolor || !dL_dmeans2D || !dL_dcolors || !dL_dopacity || !dL_dmeans3D || !scratch || !radii || !geom_buffer) {
        c3d_set_error("c3d_gs_backward: NULL buffer");
        return -1;
    }
    if (!colors_precomp && !dL_dsh) { c3d_set_error("c3d_gs_backward: dL_dsh is NULL"); return -1; }
    if (!cov3D_precomp && (!dL_dscales || !dL_drotations)) { c3d_set_error("c3d_gs_backward: dL_dscales/dL_drotations is NULL"); return -1; }
    const int tiles = p.gx * p.gy;
    GsGeom g;
    gs_carve_geom((char*)geom_buffer, N, g);
    GsBinning b;
    gs_carve_binning((char*)binning_buffer, num_rendered, tiles, b);
    GsImage im;
    gs_carve_image((char*)image_buffer, p.W, p.H, im);
    float* pairgrad = (float*)scratch;   // [num_rendered][GS_PAIR_FLOATS]
    uint8_t* pvalid = (uint8_t*)scratch + pairgrad_bytes(num_rendered);
    int rc;
    if (num_rendered > 0 && tiles > 0) {
        if (!binning_buffer || !image_buffer) { c3d_set_error("c3d_gs_backward: NULL state buffer"); return -1; }
        const int res = sort_result_index(tile_sort_bits(tiles));
        C3dProfScope ps(C3D_P_COMPOSITE_BWD, s);
        if ((rc = gs_launch_composite_bwd(p, g, b, res, im, dL_dcolor, dL_ddepth, dL_dalpha, pairgrad, pvalid, num_rendered, s))) return rc;
    }
    C3dProfScope ps(C3D_P_PREPROCESS_BWD, s);
    return gs_launch_preprocess_bwd(p, g, radii, means3D, shs, colors_precomp, scales, rotations, cov3D_precomp, pairgrad, pvalid, dL_dmeans2D,
                                    dL_dcolors, dL_dopacity, dL_dmeans3D, dL_dcov3D, dL_dsh, dL_dscales, dL_drotations, s);
}

int c3d_gs_backward_raw(const c3d_gs_settings* st, int32_t N, const float* means3D, const float* f_dc, const float* f_rest,
                        const float* scaling_raw, const float* rotation_raw, const int32_t* radii, const void* geom_buffer, int64_t num_rendered,
                        const void* binning_buffer, const void* image_buffer, const float* dL_dcolor, const float* dL_ddepth, const float* dL_dalpha,
                        float* dL_dmeans2D, float* dL_dmeans3D, float* dL_df_dc, float* dL_df_rest, float* dL_dopacity_raw, float* dL_dscaling_raw,
                        float* dL_drotation_raw, void* scratch, int32_t accumulate, c3d_stream_t stream) {
    hipStream_t s = (hipStream_t)stream;
    GsParams p;
    if (make_params(st, N, 16, p)) return -1;
    if (N == 0) return 0;
    if (!means3D || !f_dc || !f_rest || !scaling_raw || !rotation_raw || !radii || !geom_buffer || !dL_dcolor || !dL_dmeans2D || !dL_dmeans3D ||
        !dL_df_dc || !dL_df_rest || !dL_dopacity_raw || !dL_dscaling_raw || !dL_drotation_raw || !scratch) { c3d_set_error("c3d_gs_backward_raw: NULL pointer"); return -1; }
    if ((uintptr_t)f_rest % 16 || (uintptr_t)dL_df_rest % 16 || (uintptr_t)rotation_raw % 16 || (uintptr_t)dL_drotation_raw % 16) { c3d_set_error("c3d_gs_backward_raw: f_rest / rotation tensors must be 16-byte aligned"); return -1; }
    const int tiles = p.gx * p.gy;
    GsGeom g;
    gs_carve_geom((char*)geom_buffer, N, g);
    GsBinning b;
    gs_carve_binning((char*)binning_buffer, num_rendered, tiles, b);
    GsImage im;
    gs_carve_image((char*)image_buffer, p.W, p.H, im);
    float* pairgrad = (float*)scratch;
    uint8_t* pvalid = (uint8_t*)scratch + pairgrad_bytes(num_rendered);
    int rc;
    if (num_rendered > 0 && tiles > 0) {
        if (!binning_buffer || !image_buffer) { c3d_set_error("c3d_gs_backward_raw: NULL state buffer"); return -1; }
        const int res = sort_result_index(tile_sort_bits(tiles));
        C3dProfScope ps(C3D_P_COMPOSITE_BWD, s);
        if ((rc = gs_launch_composite_bwd(p, g, b, res, im, dL_dcolor, dL_ddepth, dL_dalpha, pairgrad, pvalid, num_rendered, s))) return rc;
    }
    C3dProfScope ps(C3D_P_PREPROCESS_BWD, s);
    return gs_launch_preprocess_bwd_raw(p, g, radii, means3D, f_dc, f_rest, scaling_raw, rotation_raw, pairgrad, pvalid, dL_dmeans2D, dL_dopacity_raw,
                                        dL_dmeans3D, dL_df_dc, dL_df_rest, dL_dscaling_raw, dL_drotation_raw, accumulate != 0, s);
}

// ---- fused multi-view paths (no host synchronisation inside) ----------------------------------------------------------------
struct StepWs {
    char* geom; char* binning; char* image; int* radii; float* color; float* depth; float* alpha; float* dcolor; float* dalpha; float* pairgrad; uint8_t* pvalid;
    float* dmeans2D; float* gcol; char* ms_ws; float* tile_loss;
    size_t bytes;
};
// fwd_only (c3d_gs_render_views_raw: no backward pass will read the slice): the gradient records, their valid bytes, the loss buffers and the MS-SSIM workspace --
// three quarters of a training slice at the BASELINE size -- are left out
static void carve_step(char* base, int N, int H, int W, long long cap, StepWs& w, bool fwd_only = false) {
    size_t off = 0;
    auto take = [&](size_t b) { char* p = base ? base + off : nullptr; off += c3d_align(b); return p; };
    const size_t P = (size_t)H * W, n = (size_t)(N > 0 ? N : 1);
    GsGeom g; gs_carve_geom(nullptr, N, g);
    GsBinning b; gs_carve_binning(nullptr, cap, ((W + C3D_TILE_X - 1) / C3D_TILE_X) * ((H + C3D_TILE_Y - 1) / C3D_TILE_Y), b);
    GsImage im; gs_carve_image(nullptr, W, H, im);
    w.geom = take(g.bytes); w.binning = take(b.bytes); w.image = take(im.bytes);
    w.radii = (int*)take(4 * n);
    w.color = (float*)take(12 * P); w.depth = (float*)take(4 * P); w.alpha = (float*)take(4 * P);
    w.dcolor = w.dalpha = w.pairgrad = w.dmeans2D = w.gcol = w.tile_loss = nullptr; w.pvalid = nullptr; w.ms_ws = nullptr;
    if (!fwd_only) {
        w.dcolor = (float*)take(12 * P); w.dalpha = (float*)take(4 * P);
        w.pairgrad = (float*)take(sizeof(float) * GS_PAIR_FLOATS * (size_t)(cap > 0 ? cap : 1));
        w.pvalid = (uint8_t*)take((size_t)(cap > 0 ? cap : 1));
        w.dmeans2D = (float*)take(12 * n);
        w.gcol = (float*)take(12 * n);
        w.ms_ws = take(c3d_msssim_workspace_bytes(1, 3, H, W));       // MS-SSIM term of the pixel loss (w_ssim != 0)
        w.tile_loss = (float*)take(4 * (size_t)(((W + C3D_TILE_X - 1) / C3D_TILE_X) * ((H + C3D_TILE_Y - 1) / C3D_TILE_Y) + 2));   // per-tile partial sums of the pixel loss + one slot for the view's MS-SSIM term + one for the view's sum
    }
    w.bytes = off;
}

// ---- view lanes: the V views of a call are dealt round-robin onto `lanes` HIP streams (lane 0 = the caller's stream) so that the latency-bound
// sort / scan chains of one view run underneath the VALU-bound compositing kernels of another.  Every view owns a workspace slice; after the
// join ONE pass over the Gaussians turns all views' pair records into the parameter gradients.
namespace {
struct LanePool {
    bool init = false; hipStream_t st[C3D_MAX_LANES - 1]; hipEvent_t fork, join[C3D_MAX_LANES - 1];
    // forward-only rendering in groups of L views: a projection stream running two groups ahead of the lanes (render_views_grouped)
    hipStream_t pre; hipEvent_t pre_done[2], lane_done[2][C3D_MAX_LANES];
    hipEvent_t bin_done[16];      // binning chains running ahead of the compositing lanes (train_views): view v -> bin_done[v % 16]
    hipEvent_t comp_done[16];     // compositing token (train_views): view v's compositing kernels are done -> comp_done[v % 16]
    std::mutex busy;              // held from a call's fork to its join: two host threads (or two step objects) on one device take turns instead of re-recording each other's events
};
LanePool g_lanes[16];
std::mutex g_lane_mu;
int lane_pool(LanePool** out) {
    int dev = 0;
    C3D_CHECK(hipGetDevice(&dev));
    if (dev < 0 || dev >= 16) { c3d_set_error("c3d: device ordinal %d out of range", dev); return -1; }
    std::lock_guard<std::mutex> lk(g_lane_mu);
    LanePool& lp = g_lanes[dev];
    if (!lp.init) {
        C3D_CHECK(hipEventCreateWithFlags(&lp.fork, hipEventDisableTiming));
        for (int i = 0; i < C3D_MAX_LANES - 1; i++) {
            C3D_CHECK(hipStreamCreateWithFlags(&lp.st[i], hipStreamNonBlocking));
            C3D_CHECK(hipEventCreateWithFlags(&lp.join[i], hipEventDisableTiming));
        }
        C3D_CHECK(hipStreamCreateWithFlags(&lp.pre, hipStreamNonBlocking));
        for (int i = 0; i < 2; i++) {
            C3D_CHECK(hipEventCreateWithFlags(&lp.pre_done[i], hipEventDisableTiming));
            for (int l = 0; l < C3D_MAX_LANES; l++) C3D_CHECK(hipEventCreateWithFlags(&lp.lane_done[i][l], hipEventDisableTiming));
        }
        for (int i = 0; i < 16; i++) C3D_CHECK(hipEventCreateWithFlags(&lp.bin_done[i], hipEventDisableTiming));
        for (int i = 0; i < 16; i++) C3D_CHECK(hipEventCreateWithFlags(&lp.comp_done[i], hipEventDisableTiming));
        lp.init = true;
    }
    *out = &lp;
    return 0;
}
struct Lanes {
    int L = 1; LanePool* lp = nullptr; hipStream_t s0 = nullptr; hipStream_t ls[C3D_MAX_LANES];
    bool locked = false;
    int fork(hipStream_t caller, int lanes, int V) {
        s0 = caller; L = lanes < V ? lanes : V; ls[0] = s0;
        if (lane_pool(&lp)) return -1;
        lp->busy.lock(); locked = true;           // the pool's streams and events are this call's until join()
        if (L > 1) {
            if (hipEventRecord(lp->fork, s0) != hipSuccess) { unlock(); c3d_set_error("lane fork failed"); return -1; }
            for (int l = 1; l < L; l++) {
                ls[l] = lp->st[l - 1];
                if (hipStreamWaitEvent(ls[l], lp->fork, 0) != hipSuccess) { unlock(); c3d_set_error("lane fork failed"); return -1; }
            }
        }
        return 0;
    }
    void unlock() { if (locked && lp) { lp->busy.unlock(); locked = false; } }
    int need_pool() { return lp ? 0 : lane_pool(&lp); }
    // always executed, so the caller's stream never runs ahead of work queued on the lanes (also after an error)
    int join(const char* who) {
        int rc = 0;
        for (int l = 1; l < L; l++) {
            if (hipEventRecord(lp->join[l - 1], ls[l]) != hipSuccess || hipStreamWaitEvent(s0, lp->join[l - 1], 0) != hipSuccess) {
                (void)hipDeviceSynchronize();
                c3d_set_error("%s: lane join failed", who); rc = -1;
            }
        }
        unlock();
        return rc;
    }
};
}  // namespace

// the same lanes for the other multi-view step of the library (mesh.hip)
extern "C++" {
int c3d_lanes_fork(hipStream_t caller, int lanes, int n_views, hipStream_t* streams, int* L) {
    Lanes ln;
    if (ln.fork(caller, lanes, n_views)) return -1;
    for (int l = 0; l < ln.L; l++) streams[l] = ln.ls[l];
    *L = ln.L;
    return 0;                                      // the pool stays locked: c3d_lanes_join (same thread) releases it
}
int c3d_lanes_join(hipStream_t caller, const hipStream_t* streams, int L, const char* who) {
    Lanes ln;
    ln.s0 = caller; ln.L = L;
    for (int l = 0; l < L; l++) ln.ls[l] = streams[l];
    if (ln.need_pool()) return -1;
    ln.locked = true;                              // taken by the matching c3d_lanes_fork
    return ln.join(who);
}
}  // extern "C++"

// forward of one view of the fused paths (A1-A6), everything on stream s, no host synchronisation: the pair count stays on the device
// (g.meta[0]) and every launch that depends on it is sized for the pair capacity.
// `projected`: A1 of this view has already run (step_preprocess_all)
// A2-A5 of one view (the latency-bound part of its chain: scans, two radix sorts, emit, ranges), sync-free
static int step_view_binning(const GsParams& p, GsGeom& g, GsBinning& b, int* radii, uint32_t cap, uint32_t* status, hipStream_t s, int* res_out) {
    int rc;
    if ((rc = binning_front(g, p.N, cap, status, s))) return rc;
    return binning_back(p, g, b, radii, (long long)cap, cap, (const uint32_t*)g.meta, status, s, res_out);
}
static int step_view_forward(const GsParams& p, const float* means3D, const float* f_dc, const float* f_rest, const float* opacity_raw, const float* scaling_raw,
                             const float* rotation_raw, GsGeom& g, GsBinning& b, GsImage& im, int* radii, uint32_t cap, uint32_t* status, float* color, float* depth,
                             float* alpha, bool record_activity, hipStream_t s, int* res_out, bool projected = false, bool binned = false) {
    int rc, res = binned ? *res_out : 0;
    if (!projected) {
        C3dProfScope ps(C3D_P_PREPROCESS, s);
        if ((rc = gs_launch_preprocess_raw(p, means3D, f_dc, f_rest, opacity_raw, scaling_raw, rotation_raw, g, radii, s))) return rc;
    }
    if (!binned && (rc = step_view_binning(p, g, b, radii, cap, status, s, &res))) return rc;
    { C3dProfScope ps(C3D_P_COMPOSITE_FWD, s);
      if ((rc = gs_launch_composite_fwd(p, g, b, res, im, color, depth, alpha, record_activity, s))) return rc; }
    *res_out = res;
    return 0;
}
// Binning ahead (round 3, measured and left OFF): with all views projected up front, the views' binning chains -- 15 launches of 10-30 us each that cannot
// fill the machine -- can run on the pool's streams the compositing lanes do not use, every chain from the start of the step, instead of at the head of each
// view on its lane (under four lanes in lockstep the chains of four views coincide twice per 8-view step).  On the MI355X box: 6.50 ms per step against 5.89 ms
// with the chains on the lanes (profiles/r03/r03c_bench_binahead_on.json / _off.json): eight streams share four hardware queues, and a lane's compositing
// kernel then waits behind another stream's chain in its queue.  C3D_BIN_AHEAD=1 enables it (worth re-measuring with GPU_MAX_HW_QUEUES=8).
static bool bin_ahead() {
    static int v = -1;
    if (v < 0) { const char* e = getenv("C3D_BIN_AHEAD"); v = e ? atoi(e) != 0 : 0; }
    return v != 0;
}

// A1 of every view of a step whose views keep their own workspace slice: the parameters are streamed once for up to GS_MAX_BWD_VIEWS views
// (k_preprocess_views) instead of once per view.  Runs on the caller's stream BEFORE the lanes fork.  C3D_PRE_MULTIVIEW=0 keeps the per-view launches.
static bool pre_multiview() {
    static int v = -1;
    if (v < 0) { const char* e = getenv("C3D_PRE_MULTIVIEW"); v = e ? atoi(e) != 0 : 1; }
    return v != 0;
}
// Staggered projection (round 3, C3D_PRE_SPLIT=1; measured, OFF by default): only the first `lanes` views are projected before the fork; the rest is projected on
// lane 0's stream right behind view 0's binning chain, i.e. while ALL lanes sit in their first, latency-bound binning chains; the lanes wait for it (one event)
// before their second view.  The serial head of the step shrinks from a V-view projection to an L-view one -- and the step does not: 5.946 / 6.002 ms against
// 5.945 / 5.924 ms on the same box (profiles/r03/r03i_*): the projection kernel fills every CU (128-thread workgroups, 25 KB of LDS each) and the other lanes'
// binning chains then queue behind it for slots, i.e. what the head saves the first binning round loses.
static bool pre_split() {
    static int v = -1;
    if (v < 0) { const char* e = getenv("C3D_PRE_SPLIT"); v = e ? atoi(e) != 0 : 0; }
    return v != 0;
}
// Compositing tokens (round 3).  Hypothesis: left alone the lanes run in lockstep -- they finish their binning chains together, their compositing kernels share the
// machine and finish together, and all of them sit in the next latency-bound binning chain together.  With k tokens view v's compositing kernels wait for those of
// view v - k.  Measured (profiles/r03/r03k_*, ms per 8-view step, two runs each): unconstrained 5.878 / 5.894, k = 1 (a strict chain of compositing kernels)
// 6.306 / 6.311, k = 2 5.843 / 5.838; training step 8.027 / 9.259 / 8.001.  The strict chain LOSES 0.42 ms: kernels of one chain leave a 15-50 us hole at
// every boundary (memset + dependent dispatch + the tail of the grid) that only a second, independent chain fills -- the step's 0.7 ms above the sum of its
// throughput-bound kernels is those ~20 boundaries, not lockstep.  Two tokens keep two chains in flight and win a little.  C3D_COMP_TOKENS=k, 0 = unconstrained.
static int comp_tokens() {
    static int v = -1;
    if (v < 0) { const char* e = getenv("C3D_COMP_TOKENS"); v = e ? atoi(e) : 2; if (v < 0 || v > 8) v = 0; }
    return v;
}
static int step_preprocess_all(const c3d_gs_settings* views, int V, int N, size_t slice_bytes, void* workspace, long long pair_capacity, const float* means3D,
                               const float* f_dc, const float* f_rest, const float* opacity_raw, const float* scaling_raw, const float* rotation_raw, hipStream_t s0,
                               int first = 0) {
    for (int v0 = first; v0 < V; v0 += GS_MAX_BWD_VIEWS) {
        const int nv = (V - v0) < GS_MAX_BWD_VIEWS ? (V - v0) : GS_MAX_BWD_VIEWS;
        GsParams ps[GS_MAX_BWD_VIEWS];
        GsGeom gs[GS_MAX_BWD_VIEWS];
        int* radii[GS_MAX_BWD_VIEWS];
        for (int i = 0; i < nv; i++) {
            if (make_params(&views[v0 + i], N, 16, ps[i])) return -1;
            StepWs w; carve_step((char*)workspace + (size_t)(v0 + i) * slice_bytes, N, ps[i].H, ps[i].W, pair_capacity, w);
            gs_carve_geom(w.geom, N, gs[i]);
            radii[i] = w.radii;
        }
        C3dProfScope sc(C3D_P_PREPROCESS, s0);
        int rc;
        if ((rc = gs_launch_preprocess_views(ps, nv, gs, radii, means3D, f_dc, f_rest, opacity_raw, scaling_raw, rotation_raw, s0))) return rc;
    }
    return 0;
}

// per-Gaussian chain rule over all views of a step, every gradient written once (chunks of GS_MAX_BWD_VIEWS views); stream s0, after the join
static int step_a8_all_views(const c3d_gs_settings* views, int V, int N, size_t slice_bytes, void* workspace, long long pair_capacity, const float* means3D,
                             const float* f_dc, const float* f_rest, const float* scaling_raw, const float* rotation_raw, float* dL_dmeans3D, float* dL_df_dc,
                             float* dL_df_rest, float* dL_dopacity_raw, float* dL_dscaling_raw, float* dL_drotation_raw, bool accumulate, hipStream_t s0,
                             int first = 0, int count = -1) {
    const uint32_t cap = (uint32_t)pair_capacity;
    GsParams p_first{};
    for (int v0 = 0; v0 < V; v0 += GS_MAX_BWD_VIEWS) {
        GsBwdViews bv;
        bv.V = (V - v0) < GS_MAX_BWD_VIEWS ? (V - v0) : GS_MAX_BWD_VIEWS;
        for (int i = 0; i < bv.V; i++) {
            GsParams p;
            if (make_params(&views[v0 + i], N, 16, p)) return -1;
            if (v0 + i == 0) p_first = p;
            StepWs w; carve_step((char*)workspace + (size_t)(v0 + i) * slice_bytes, N, p.H, p.W, pair_capacity, w);
            GsGeom g; gs_carve_geom(w.geom, N, g);
            GsBwdView& o = bv.v[i];
            o.view = p.view; o.proj = p.proj; o.campos = p.campos; o.radii = w.radii; o.rec0 = g.rec0; o.rec1 = g.rec1; o.tiles = g.tiles; o.rbase = g.rbase;
            o.clamped = g.clamped; o.pairgrad = (const float4*)w.pairgrad; o.pvalid = w.pvalid; o.dmean2D = w.dmeans2D; o.gcol = w.gcol;
            o.tanfovx = p.tanfovx; o.tanfovy = p.tanfovy; o.focal_x = p.focal_x; o.focal_y = p.focal_y;
        }
        int rc;
        C3dProfScope ps(C3D_P_PREPROCESS_BWD, s0);
        if ((rc = gs_launch_preprocess_bwd_views(p_first, bv, means3D, f_dc, f_rest, scaling_raw, rotation_raw, dL_dopacity_raw, dL_dmeans3D, dL_df_dc, dL_df_rest,
                                                 dL_dscaling_raw, dL_drotation_raw, accumulate || v0 > 0, s0, cap, first, count))) return rc;
    }
    return 0;
}

static int check_step_args(const char* who, const c3d_gs_settings* views, int V, int64_t pair_capacity, int lanes, const void* workspace) {
    if (!views || !workspace) { c3d_set_error("%s: NULL pointer", who); return -1; }
    if (pair_capacity <= 0 || pair_capacity > (int64_t)0x3FFFFFF0ll) { c3d_set_error("%s: pair_capacity out of range (1 .. 2^30 - 16)", who); return -1; }
    if (lanes < 1 || lanes > C3D_MAX_LANES) { c3d_set_error("%s: lanes must be in [1, %d]", who, C3D_MAX_LANES); return -1; }
    for (int v = 1; v < V; v++)
        if (views[v].image_width != views[0].image_width || views[v].image_height != views[0].image_height || views[v].sh_degree != views[0].sh_degree ||
            views[v].scale_modifier != views[0].scale_modifier) { c3d_set_error("%s: all views must share resolution, sh_degree and scale_modifier", who); return -1; }
    return 0;
}

size_t c3d_gs_step_workspace_bytes(int32_t N, int32_t H, int32_t W, int64_t pair_capacity, int32_t views) {
    StepWs w; carve_step(nullptr, N, H, W, pair_capacity, w);
    return (size_t)(views > 0 ? views : 1) * w.bytes;
}
size_t c3d_gs_render_workspace_bytes(int32_t N, int32_t H, int32_t W, int64_t pair_capacity, int32_t slices) {
    StepWs w; carve_step(nullptr, N, H, W, pair_capacity, w, true);
    return (size_t)(slices > 0 ? slices : 1) * w.bytes;
}

int c3d_gs_train_views_raw(const c3d_gs_settings* views, int32_t V, int32_t N, const float* means3D, const float* f_dc, const float* f_rest,
                           const float* opacity_raw, const float* scaling_raw, const float* rotation_raw, const float* const* target_color,
                           const float* const* target_alpha, const float* const* color_mask, const c3d_gs_loss* loss, float* dL_dmeans3D, float* dL_df_dc, float* dL_df_rest,
                           float* dL_dopacity_raw, float* dL_dscaling_raw, float* dL_drotation_raw, float* loss_out, int64_t pair_capacity, int32_t lanes,
                           int32_t accumulate, void* workspace, uint32_t* status, c3d_stream_t stream) {
    hipStream_t s0 = (hipStream_t)stream;
    if (V <= 0 || N <= 0) return 0;
    if (!loss || !target_color || !status) { c3d_set_error("c3d_gs_train_views_raw: NULL pointer"); return -1; }
    if (check_step_args("c3d_gs_train_views_raw", views, V, pair_capacity, lanes, workspace)) return -1;
    if (!means3D || !f_dc || !f_rest || !opacity_raw || !scaling_raw || !rotation_raw || !dL_dmeans3D || !dL_df_dc || !dL_df_rest || !dL_dopacity_raw ||
        !dL_dscaling_raw || !dL_drotation_raw) { c3d_set_error("c3d_gs_train_views_raw: NULL parameter / gradient pointer"); return -1; }
    if ((uintptr_t)f_rest % 16 || (uintptr_t)dL_df_rest % 16 || (uintptr_t)rotation_raw % 16 || (uintptr_t)dL_drotation_raw % 16) { c3d_set_error("c3d_gs_train_views_raw: f_rest / rotation tensors must be 16-byte aligned"); return -1; }
    const uint32_t cap = (uint32_t)pair_capacity;
    StepWs w0; carve_step(nullptr, N, views[0].image_height, views[0].image_width, pair_capacity, w0);
    const bool projected = pre_multiview();
    static int fuse_loss = -1;
    if (fuse_loss < 0) { const char* e = getenv("C3D_FUSE_LOSS"); fuse_loss = e ? atoi(e) != 0 : 1; }
    const int L_ = lanes < V ? lanes : V;
    const bool split = projected && pre_split() && L_ > 1 && V > L_ && !bin_ahead();
    if (projected && step_preprocess_all(views, split ? L_ : V, N, w0.bytes, workspace, pair_capacity, means3D, f_dc, f_rest, opacity_raw, scaling_raw, rotation_raw, s0)) return -1;
    Lanes ln;
    if (ln.fork(s0, lanes, V)) return -1;
    int rc_all = 0;
    const int n_bin = C3D_MAX_LANES - ln.L;                       // pool streams the lanes leave free
    const bool ahead = projected && bin_ahead() && ln.L > 1 && n_bin > 0 && V > 1;
    if (ahead)
        for (int i = 0; i < n_bin && i < V; i++)
            if (hipStreamWaitEvent(ln.lp->st[ln.L - 1 + i], ln.lp->fork, 0) != hipSuccess) { c3d_set_error("c3d_gs_train_views_raw: fork failed"); rc_all = -1; }
    for (int v = 0; v < V && !rc_all; v++) {
        hipStream_t s = ln.ls[v % ln.L];
        GsParams p;
        if (make_params(&views[v], N, 16, p)) { rc_all = -1; break; }
        if (!target_color[v]) { c3d_set_error("c3d_gs_train_views_raw: target_color[%d] is NULL", v); rc_all = -1; break; }
        const int tiles = p.gx * p.gy;
        StepWs w; carve_step((char*)workspace + (size_t)v * w0.bytes, N, p.H, p.W, pair_capacity, w);
        GsGeom g; gs_carve_geom(w.geom, N, g);
        GsBinning b; gs_carve_binning(w.binning, pair_capacity, tiles, b);
        GsImage im; gs_carve_image(w.image, p.W, p.H, im);
        int rc = 0, res = 0;
        do {
            if (ahead) {      // this view's binning chain on a free stream; its lane waits for it and goes straight to the compositing
                hipStream_t sb = ln.lp->st[ln.L - 1 + (v % n_bin)];
                if ((rc = step_view_binning(p, g, b, w.radii, cap, status, sb, &res))) break;
                if (hipEventRecord(ln.lp->bin_done[v % 16], sb) != hipSuccess || hipStreamWaitEvent(s, ln.lp->bin_done[v % 16], 0) != hipSuccess) { c3d_set_error("c3d_gs_train_views_raw: event failed"); rc = -1; break; }
            }
            bool binned = ahead;
            if (split && v == 0) {      // view 0's binning, then the projection of views L.. on this lane (lane 0); the other lanes wait for it before their second view
                if ((rc = step_view_binning(p, g, b, w.radii, cap, status, s, &res))) break;
                if ((rc = step_preprocess_all(views, V, N, w0.bytes, workspace, pair_capacity, means3D, f_dc, f_rest, opacity_raw, scaling_raw, rotation_raw, s, ln.L))) break;
                if (hipEventRecord(ln.lp->pre_done[0], s) != hipSuccess) { c3d_set_error("c3d_gs_train_views_raw: event failed"); rc = -1; break; }
                binned = true;
            } else if (split && v >= ln.L && v < 2 * ln.L && (v % ln.L) != 0) {
                if (hipStreamWaitEvent(s, ln.lp->pre_done[0], 0) != hipSuccess) { c3d_set_error("c3d_gs_train_views_raw: event failed"); rc = -1; break; }
            }
            // the "record written" bytes of the backward pass are cleared here, inside / ahead of the binning chain, not between the two compositing kernels
            if (hipMemsetAsync(w.pvalid, 0, (size_t)cap, s) != hipSuccess) { c3d_set_error("c3d_gs_train_views_raw: memset failed"); rc = -1; break; }
            const int tok = (projected && ln.L > 1) ? comp_tokens() : 0;
            if (tok && v >= tok) {      // binning first (unconstrained), then wait for the token: the compositing kernels of view v - tok are done
                if (!binned) { if ((rc = step_view_binning(p, g, b, w.radii, cap, status, s, &res))) break; binned = true; }
                if (hipStreamWaitEvent(s, ln.lp->comp_done[(v - tok) % 16], 0) != hipSuccess) { c3d_set_error("c3d_gs_train_views_raw: event failed"); rc = -1; break; }
            }
            if ((rc = step_view_forward(p, means3D, f_dc, f_rest, opacity_raw, scaling_raw, rotation_raw, g, b, im, w.radii, cap, status, w.color, w.depth, w.alpha, true, s, &res, projected, binned))) break;
            // pixel loss and its gradient.  Default: inside the backward compositing kernel (GsPixelLoss); C3D_FUSE_LOSS=0 keeps the separate launch.
            const float* tal = target_alpha ? target_alpha[v] : nullptr;
            const float* cmk = color_mask ? color_mask[v] : nullptr;
            const bool ssim = loss->w_ssim != 0.f;
            if (!fuse_loss) {
                C3dProfScope ps(C3D_P_OTHER, s);
                if ((rc = gs_launch_loss_grad(w.color, w.alpha, target_color[v], tal, cmk, (long long)p.W * p.H, loss->w_l1, loss->w_l2, loss->w_alpha_mse, loss->scale,
                                              w.dcolor, w.dalpha, loss_out, s))) break;
            }
            // + scale * w_ssim * (1 - MS-SSIM(target * mask, clamp(C) * mask)) of this view (the batch mean of the reference, main_3DGS.py:192, is the mean of
            // the per-image values): value into loss_out, gradient into (fused loss) / added to (separate launch) dL/dcolor -- ~20 launches on this view's lane
            if (ssim) {
                C3dProfScope ps(C3D_P_OTHER, s);
                const float ws_ = loss->scale * loss->w_ssim;
                // fused loss: the value goes into this view's own slot behind its tile partials (k_sum_tile_loss adds the views in order); otherwise to the shared word
                const bool slot = fuse_loss && loss_out;
                if ((rc = ms_value_grad(target_color[v], w.color, cmk, 1, 1, 3, p.H, p.W, -ws_, fuse_loss ? 0 : 1, w.dcolor, ws_, -ws_, slot ? w.tile_loss + tiles : loss_out, w.ms_ws, s, slot ? 1 : 0))) break;
            }
            // backward down to the per-(tile, splat) records of this view
            { C3dProfScope ps(C3D_P_COMPOSITE_BWD, s);
              if (fuse_loss) {
                  const GsPixelLoss pl{w.color, w.alpha, target_color[v], tal, cmk, loss->w_l1, loss->w_l2, loss->w_alpha_mse, loss->scale, loss_out ? w.tile_loss : nullptr};
                  rc = gs_launch_composite_bwd(p, g, b, res, im, ssim ? w.dcolor : nullptr, nullptr, nullptr, w.pairgrad, w.pvalid, (long long)cap, s, cap, &pl, true);
              } else {
                  rc = gs_launch_composite_bwd(p, g, b, res, im, w.dcolor, nullptr, w.dalpha, w.pairgrad, w.pvalid, (long long)cap, s, cap, nullptr, true);
              }
              if (rc) break;
              if (projected && ln.L > 1 && comp_tokens() && hipEventRecord(ln.lp->comp_done[v % 16], s) != hipSuccess) { c3d_set_error("c3d_gs_train_views_raw: event failed"); rc = -1; break; }
              if (fuse_loss && loss_out && (rc = gs_launch_sum_view_loss(w.tile_loss, tiles + (ssim ? 1 : 0), w.tile_loss + tiles + 1, s))) break; }   // this view's loss value, on its lane
        } while (0);
        rc_all = rc;
    }
    if (ahead)      // the binning streams too end in the caller's stream (every chain already has a consumer on a lane; this keeps error paths ordered as well)
        for (int i = 0; i < n_bin && i < V; i++)
            if (hipEventRecord(ln.lp->join[ln.L - 1 + i], ln.lp->st[ln.L - 1 + i]) != hipSuccess || hipStreamWaitEvent(s0, ln.lp->join[ln.L - 1 + i], 0) != hipSuccess) {
                (void)hipDeviceSynchronize();
                if (!rc_all) { c3d_set_error("c3d_gs_train_views_raw: binning stream join failed"); rc_all = -1; }
            }
    if (ln.join("c3d_gs_train_views_raw") && !rc_all) rc_all = -1;
    if (rc_all) return rc_all;
    if (fuse_loss && loss_out) {   // the views' per-tile partial sums of the pixel loss -> loss_out, in a fixed order
        StepWs wf; carve_step((char*)workspace, N, views[0].image_height, views[0].image_width, pair_capacity, wf);
        const int tiles = ((views[0].image_width + C3D_TILE_X - 1) / C3D_TILE_X) * ((views[0].image_height + C3D_TILE_Y - 1) / C3D_TILE_Y);
        if (gs_launch_sum_tile_loss(wf.tile_loss + tiles + 1, w0.bytes, V, loss_out, s0)) return -1;
    }
    if (accumulate & 2) return 0;   // the caller runs the per-Gaussian pass itself, range by range (c3d_gs_step_param_backward_range)
    return step_a8_all_views(views, V, N, w0.bytes, workspace, pair_capacity, means3D, f_dc, f_rest, scaling_raw, rotation_raw, dL_dmeans3D, dL_df_dc, dL_df_rest,
                             dL_dopacity_raw, dL_dscaling_raw, dL_drotation_raw, (accumulate & 1) != 0, s0);
}

int c3d_gs_step_param_backward_range(const c3d_gs_settings* views, int32_t V, int32_t N, const float* means3D, const float* f_dc, const float* f_rest,
                                     const float* scaling_raw, const float* rotation_raw, float* dL_dmeans3D, float* dL_df_dc, float* dL_df_rest, float* dL_dopacity_raw,
                                     float* dL_dscaling_raw, float* dL_drotation_raw, int64_t pair_capacity, int32_t accumulate, void* workspace, int32_t first,
                                     int32_t count, c3d_stream_t stream) {
    if (V <= 0 || N <= 0 || count == 0) return 0;
    if (check_step_args("c3d_gs_step_param_backward_range", views, V, pair_capacity, 1, workspace)) return -1;
    if (!means3D || !f_dc || !f_rest || !scaling_raw || !rotation_raw || !dL_dmeans3D || !dL_df_dc || !dL_df_rest || !dL_dopacity_raw || !dL_dscaling_raw || !dL_drotation_raw) {
        c3d_set_error("c3d_gs_step_param_backward_range: NULL parameter / gradient pointer"); return -1; }
    if ((uintptr_t)f_rest % 16 || (uintptr_t)dL_df_rest % 16 || (uintptr_t)rotation_raw % 16 || (uintptr_t)dL_drotation_raw % 16) { c3d_set_error("c3d_gs_step_param_backward_range: f_rest / rotation tensors must be 16-byte aligned"); return -1; }
    if (first < 0 || count < 0 || (first & 3) || (long long)first + count > N) { c3d_set_error("c3d_gs_step_param_backward_range: range [%d, %d + %d) must lie inside [0, N) and start at a multiple of 4", first, first, count); return -1; }
    StepWs w0; carve_step(nullptr, N, views[0].image_height, views[0].image_width, pair_capacity, w0);
    return step_a8_all_views(views, V, N, w0.bytes, workspace, pair_capacity, means3D, f_dc, f_rest, scaling_raw, rotation_raw, dL_dmeans3D, dL_df_dc, dL_df_rest,
                             dL_dopacity_raw, dL_dscaling_raw, dL_drotation_raw, accumulate != 0, (hipStream_t)stream, first, count);
}

// forward of V views; keep_state: view v uses workspace slice v (what c3d_gs_backward_views_raw reads), otherwise a lane reuses its slice view after view
static int views_forward(const char* who, const c3d_gs_settings* views, int32_t V, int32_t N, const float* means3D, const float* f_dc, const float* f_rest,
                         const float* opacity_raw, const float* scaling_raw, const float* rotation_raw, float* const* out_color, float* const* out_depth,
                         float* const* out_alpha, int32_t* const* out_radii, int64_t pair_capacity, int32_t lanes, void* workspace, uint32_t* status,
                         hipStream_t s0, bool keep_state, bool two_slice_sets = false) {
    if (V <= 0 || N <= 0) return 0;
    if (!out_color || !out_alpha || !status) { c3d_set_error("%s: NULL pointer", who); return -1; }
    if (check_step_args(who, views, V, pair_capacity, lanes, workspace)) return -1;
    if (!means3D || !f_dc || !f_rest || !opacity_raw || !scaling_raw || !rotation_raw) { c3d_set_error("%s: NULL parameter pointer", who); return -1; }
    if ((uintptr_t)f_rest % 16 || (uintptr_t)rotation_raw % 16) { c3d_set_error("%s: f_rest / rotation tensors must be 16-byte aligned", who); return -1; }
    const uint32_t cap = (uint32_t)pair_capacity;
    const bool fwd_only = !keep_state;       // c3d_gs_render_views_raw: slices without the backward pass's buffers (c3d_gs_render_workspace_bytes)
    StepWs w0; carve_step(nullptr, N, views[0].image_height, views[0].image_width, pair_capacity, w0, fwd_only);
    const bool projected = pre_multiview();
    if (projected && keep_state && step_preprocess_all(views, V, N, w0.bytes, workspace, pair_capacity, means3D, f_dc, f_rest, opacity_raw, scaling_raw, rotation_raw, s0)) return -1;
    Lanes ln;
    if (ln.fork(s0, lanes, V)) return -1;
    const int L = ln.L;
    int rc_all = 0;
    // Forward only (no state kept): a lane reuses workspace slices view after view.  The views go in groups of L (one per lane); group k lives in
    // slice set k % 2, and a projection stream runs k_preprocess_views for whole groups two ahead of the lanes: parameters are streamed once per
    // L views, and group k + 2 is projected as soon as every lane has finished its view of group k.
    const bool grouped = projected && !keep_state && two_slice_sets;      // forward only, and the workspace has the second slice set the projection stream fills ahead
    hipStream_t sp = nullptr;
    auto project_group = [&](int k) -> int {
        const int v0 = k * L, nv = (V - v0) < L ? (V - v0) : L;
        GsParams ps[C3D_MAX_LANES]; GsGeom gs[C3D_MAX_LANES]; int* rd[C3D_MAX_LANES];
        for (int i = 0; i < nv; i++) {
            if (make_params(&views[v0 + i], N, 16, ps[i])) return -1;
            StepWs w; carve_step((char*)workspace + (size_t)((k & 1) * L + i) * w0.bytes, N, ps[i].H, ps[i].W, pair_capacity, w, fwd_only);
            gs_carve_geom(w.geom, N, gs[i]);
            rd[i] = (out_radii && out_radii[v0 + i]) ? out_radii[v0 + i] : w.radii;
        }
        { C3dProfScope sc(C3D_P_PREPROCESS, sp);
          if (gs_launch_preprocess_views(ps, nv, gs, rd, means3D, f_dc, f_rest, opacity_raw, scaling_raw, rotation_raw, sp)) return -1; }
        C3D_CHECK(hipEventRecord(ln.lp->pre_done[k & 1], sp));
        return 0;
    };
    const int groups = (V + L - 1) / L;
    if (grouped) {
        if (ln.need_pool()) rc_all = -1;
        else {
            sp = ln.lp->pre;
            if (hipEventRecord(ln.lp->fork, s0) != hipSuccess || hipStreamWaitEvent(sp, ln.lp->fork, 0) != hipSuccess) { c3d_set_error("%s: fork failed", who); rc_all = -1; }
            for (int k = 0; k < 2 && k < groups && !rc_all; k++) rc_all = project_group(k);
        }
    }
    for (int v = 0; v < V && !rc_all; v++) {
        const int lane = v % L, k = v / L;
        hipStream_t s = ln.ls[lane];
        GsParams p;
        if (make_params(&views[v], N, 16, p)) { rc_all = -1; break; }
        if (!out_color[v] || !out_alpha[v]) { c3d_set_error("%s: output %d is NULL", who, v); rc_all = -1; break; }
        const int slice = keep_state ? v : (grouped ? (k & 1) * L + lane : lane);
        StepWs w; carve_step((char*)workspace + (size_t)slice * w0.bytes, N, p.H, p.W, pair_capacity, w, fwd_only);
        GsGeom g; gs_carve_geom(w.geom, N, g);
        GsBinning b; gs_carve_binning(w.binning, pair_capacity, p.gx * p.gy, b);
        GsImage im; gs_carve_image(w.image, p.W, p.H, im);
        int res = 0;
        int* radii = (!keep_state && out_radii && out_radii[v]) ? out_radii[v] : w.radii;      // kept state: the backward pass reads the slice's copy
        float* depth = (out_depth && out_depth[v]) ? out_depth[v] : w.depth;
        if (grouped && hipStreamWaitEvent(s, ln.lp->pre_done[k & 1], 0) != hipSuccess) { c3d_set_error("%s: wait failed", who); rc_all = -1; break; }
        rc_all = step_view_forward(p, means3D, f_dc, f_rest, opacity_raw, scaling_raw, rotation_raw, g, b, im, radii, cap, status, out_color[v], depth, out_alpha[v], keep_state, s, &res, keep_state ? projected : grouped);
        if (!rc_all && keep_state && out_radii && out_radii[v] &&
            hipMemcpyAsync(out_radii[v], w.radii, sizeof(int) * (size_t)N, hipMemcpyDeviceToDevice, s) != hipSuccess) {
            c3d_set_error("%s: radii copy failed", who); rc_all = -1;       // no early return: the lanes must still be joined
        }
        if (grouped && !rc_all) {
            if (hipEventRecord(ln.lp->lane_done[k & 1][lane], s) != hipSuccess) { c3d_set_error("%s: record failed", who); rc_all = -1; break; }
            const bool last_of_group = (lane == L - 1) || (v == V - 1);
            if (last_of_group && k + 2 < groups) {      // slice set k % 2 is free once every lane is through group k: project group k + 2 into it
                for (int l = 0; l <= lane && !rc_all; l++)
                    if (hipStreamWaitEvent(sp, ln.lp->lane_done[k & 1][l], 0) != hipSuccess) { c3d_set_error("%s: wait failed", who); rc_all = -1; }
                if (!rc_all) rc_all = project_group(k + 2);
            }
        }
    }
    if (grouped && sp) {   // nothing may be left running on the projection stream when the caller's stream continues (error paths included)
        if (hipEventRecord(ln.lp->pre_done[0], sp) != hipSuccess || hipStreamWaitEvent(s0, ln.lp->pre_done[0], 0) != hipSuccess) { (void)hipDeviceSynchronize(); if (!rc_all) { c3d_set_error("%s: join failed", who); rc_all = -1; } }
    }
    if (ln.join(who) && !rc_all) rc_all = -1;
    return rc_all;
}

int c3d_gs_render_views_raw(const c3d_gs_settings* views, int32_t V, int32_t N, const float* means3D, const float* f_dc, const float* f_rest,
                             const float* opacity_raw, const float* scaling_raw, const float* rotation_raw, float* const* out_color, float* const* out_depth,
                             float* const* out_alpha, int32_t* const* out_radii, int64_t pair_capacity, int32_t lanes, void* workspace, int64_t workspace_bytes,
                             uint32_t* status, c3d_stream_t stream) {
    if (V > 0 && N > 0 && !out_depth) { c3d_set_error("c3d_gs_render_views_raw: NULL pointer"); return -1; }
    if (V > 0 && N > 0 && views && lanes >= 1 && pair_capacity > 0) {      // the slice count decides the schedule: say what the buffer holds instead of trusting a convention
        const size_t one = c3d_gs_render_workspace_bytes(N, views[0].image_height, views[0].image_width, pair_capacity, 1);
        const int L = lanes < V ? lanes : V;
        if (workspace_bytes < (int64_t)(one * (size_t)L)) { c3d_set_error("c3d_gs_render_views_raw: workspace of %lld bytes holds fewer than %d slices of %zu bytes", (long long)workspace_bytes, L, one); return -1; }
        const bool two_sets = workspace_bytes >= (int64_t)(one * (size_t)(2 * L));
        return views_forward("c3d_gs_render_views_raw", views, V, N, means3D, f_dc, f_rest, opacity_raw, scaling_raw, rotation_raw, out_color, out_depth, out_alpha, out_radii,
                             pair_capacity, lanes, workspace, status, (hipStream_t)stream, false, two_sets);
    }
    return views_forward("c3d_gs_render_views_raw", views, V, N, means3D, f_dc, f_rest, opacity_raw, scaling_raw, rotation_raw, out_color, out_depth, out_alpha, out_radii,
                         pair_capacity, lanes, workspace, status, (hipStream_t)stream, false, false);
}

int c3d_gs_forward_views_raw(const c3d_gs_settings* views, int32_t V, int32_t N, const float* means3D, const float* f_dc, const float* f_rest,
                              const float* opacity_raw, const float* scaling_raw, const float* rotation_raw, float* const* out_color, float* const* out_depth,
                              float* const* out_alpha, int32_t* const* out_radii, int64_t pair_capacity, int32_t lanes, void* workspace, uint32_t* status,
                              c3d_stream_t stream) {
    return views_forward("c3d_gs_forward_views_raw", views, V, N, means3D, f_dc, f_rest, opacity_raw, scaling_raw, rotation_raw, out_color, out_depth, out_alpha, out_radii,
                         pair_capacity, lanes, workspace, status, (hipStream_t)stream, true);
}

int c3d_gs_backward_views_raw(const c3d_gs_settings* views, int32_t V, int32_t N, const float* means3D, const float* f_dc, const float* f_rest,
                               const float* scaling_raw, const float* rotation_raw, const float* const* dL_dcolor, const float* const* dL_ddepth,
                               const float* const* dL_dalpha, float* dL_dmeans3D, float* dL_df_dc, float* dL_df_rest, float* dL_dopacity_raw, float* dL_dscaling_raw,
                               float* dL_drotation_raw, int64_t pair_capacity, int32_t lanes, int32_t accumulate, void* workspace, c3d_stream_t stream) {
    hipStream_t s0 = (hipStream_t)stream;
    if (V <= 0 || N <= 0) return 0;
    if (!dL_dcolor) { c3d_set_error("c3d_gs_backward_views_raw: NULL pointer"); return -1; }
    if (check_step_args("c3d_gs_backward_views_raw", views, V, pair_capacity, lanes, workspace)) return -1;
    if (!means3D || !f_dc || !f_rest || !scaling_raw || !rotation_raw || !dL_dmeans3D || !dL_df_dc || !dL_df_rest || !dL_dopacity_raw || !dL_dscaling_raw ||
        !dL_drotation_raw) { c3d_set_error("c3d_gs_backward_views_raw: NULL parameter / gradient pointer"); return -1; }
    if ((uintptr_t)f_rest % 16 || (uintptr_t)dL_df_rest % 16 || (uintptr_t)rotation_raw % 16 || (uintptr_t)dL_drotation_raw % 16) { c3d_set_error("c3d_gs_backward_views_raw: f_rest / rotation tensors must be 16-byte aligned"); return -1; }
    const uint32_t cap = (uint32_t)pair_capacity;
    StepWs w0; carve_step(nullptr, N, views[0].image_height, views[0].image_width, pair_capacity, w0);
    Lanes ln;
    if (ln.fork(s0, lanes, V)) return -1;
    int rc_all = 0;
    for (int v = 0; v < V && !rc_all; v++) {
        hipStream_t s = ln.ls[v % ln.L];
        GsParams p;
        if (make_params(&views[v], N, 16, p)) { rc_all = -1; break; }
        if (!dL_dcolor[v]) { c3d_set_error("c3d_gs_backward_views_raw: dL_dcolor[%d] is NULL", v); rc_all = -1; break; }
        const int tiles = p.gx * p.gy;
        StepWs w; carve_step((char*)workspace + (size_t)v * w0.bytes, N, p.H, p.W, pair_capacity, w);
        GsGeom g; gs_carve_geom(w.geom, N, g);
        GsBinning b; gs_carve_binning(w.binning, pair_capacity, tiles, b);
        GsImage im; gs_carve_image(w.image, p.W, p.H, im);
        C3dProfScope ps(C3D_P_COMPOSITE_BWD, s);
        rc_all = gs_launch_composite_bwd(p, g, b, sort_result_index(tile_sort_bits(tiles)), im, dL_dcolor[v], dL_ddepth ? dL_ddepth[v] : nullptr,
                                         dL_dalpha ? dL_dalpha[v] : nullptr, w.pairgrad, w.pvalid, (long long)cap, s, cap);
    }
    if (ln.join("c3d_gs_backward_views_raw") && !rc_all) rc_all = -1;
    if (rc_all) return rc_all;
    return step_a8_all_views(views, V, N, w0.bytes, workspace, pair_capacity, means3D, f_dc, f_rest, scaling_raw, rotation_raw, dL_dmeans3D, dL_df_dc, dL_df_rest,
                             dL_dopacity_raw, dL_dscaling_raw, dL_drotation_raw, accumulate != 0, s0);
}

int c3d_gs_step_read_view(int32_t N, int32_t H, int32_t W, int64_t pair_capacity, const void* workspace, int32_t view, int32_t* radii_out,
                          float* dL_dmeans2D_out, c3d_stream_t stream) {
    if (N <= 0) return 0;
    if (!workspace || view < 0) { c3d_set_error("c3d_gs_step_read_view: bad argument"); return -1; }
    StepWs w0; carve_step(nullptr, N, H, W, pair_capacity, w0);
    StepWs w; carve_step((char*)workspace + (size_t)view * w0.bytes, N, H, W, pair_capacity, w);
    hipStream_t s = (hipStream_t)stream;
    if (radii_out) C3D_CHECK(hipMemcpyAsync(radii_out, w.radii, sizeof(int) * (size_t)N, hipMemcpyDeviceToDevice, s));
    if (dL_dmeans2D_out) C3D_CHECK(hipMemcpyAsync(dL_dmeans2D_out, w.dmeans2D, sizeof(float) * 3 * (size_t)N, hipMemcpyDeviceToDevice, s));
    return 0;
}

int c3d_gs_mark_visible(int32_t N, const float* means3D, const float* viewmatrix, const float* projmatrix, uint8_t* present, c3d_stream_t stream) {
    if (N > 0 && (!means3D || !viewmatrix || !present)) { c3d_set_error("c3d_gs_mark_visible: NULL pointer"); return -1; }
    return gs_launch_mark_visible(N, means3D, viewmatrix, projmatrix, present, (hipStream_t)stream);
}

// ---- test / introspection hooks ----
__global__ void k_debug_unpack(int N, GsGeom g, float* xy, float* depths, float* conic_opacity, float* rgb, uint32_t* tiles) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    const bool vis = g.key[0] != nullptr;   // records of culled Gaussians are never written; callers mask with radii
    const float4 a0 = vis ? g.rec0[GS_REC(i)] : make_float4(0, 0, 0, 0), a1 = vis ? g.rec1[GS_REC(i)] : make_float4(0, 0, 0, 0);
    const float4 a2 = vis ? g.rec2[GS_REC(i)] : make_float4(0, 0, 0, 0);
    if (xy) { xy[2 * i] = a0.x; xy[2 * i + 1] = a0.y; }
    if (depths) depths[i] = a2.y;
    if (conic_opacity) { conic_opacity[4 * i] = a0.z; conic_opacity[4 * i + 1] = a0.w; conic_opacity[4 * i + 2] = a1.x; conic_opacity[4 * i + 3] = a1.y; }
    if (rgb) { rgb[3 * i] = a1.z; rgb[3 * i + 1] = a1.w; rgb[3 * i + 2] = a2.x; }
    if (tiles) tiles[i] = g.tiles[i];
}
int c3d_gs_debug_state(int32_t N, int32_t H, int32_t W, const void* geom_buffer, int64_t D, const void* binning_buffer,
                       uint32_t* point_list, uint32_t* ranges, float* xy, float* depths, float* conic_opacity, float* rgb,
                       uint32_t* tiles_touched, c3d_stream_t stream) {
    hipStream_t s = (hipStream_t)stream;
    const int tiles = ((W + C3D_TILE_X - 1) / C3D_TILE_X) * ((H + C3D_TILE_Y - 1) / C3D_TILE_Y);
    if (N > 0 && geom_buffer) {
        GsGeom g;
        gs_carve_geom((char*)geom_buffer, N, g);
        hipLaunchKernelGGL(k_debug_unpack, dim3(c3d_cdiv(N, 256)), dim3(256), 0, s, N, g, xy, depths, conic_opacity, rgb, tiles_touched);
        C3D_LAUNCH_CHECK();
    }
    if (binning_buffer && tiles > 0) {
        GsBinning b;
        gs_carve_binning((char*)binning_buffer, D, tiles, b);
        const int res = sort_result_index(tile_sort_bits(tiles));
        if (point_list && D > 0) C3D_CHECK(hipMemcpyAsync(point_list, b.tval[res], sizeof(uint32_t) * (size_t)D, hipMemcpyDeviceToDevice, s));
        if (ranges) C3D_CHECK(hipMemcpyAsync(ranges, b.ranges, sizeof(uint2) * (size_t)tiles, hipMemcpyDeviceToDevice, s));
    }
    return 0;
}

int c3d_test_scan_u32(const uint32_t* in, uint32_t* out, int64_t n, int32_t exclusive, c3d_stream_t stream) {
    hipStream_t s = (hipStream_t)stream;
    if (n <= 0) return 0;
    void* tmp = nullptr;
    C3D_CHECK(hipMalloc(&tmp, c3d_scan_tmp_bytes((size_t)n)));
    int rc = c3d_scan_u32(in, out, (size_t)n, exclusive != 0, tmp, s);
    (void)hipStreamSynchronize(s);
    (void)hipFree(tmp);
    return rc;
}
int c3d_test_sort_phases(uint64_t* stamps) { return c3d_sort_set_debug((unsigned long long*)stamps); }
int c3d_test_segment_sort_u32(const uint32_t* ranges, int32_t nseg, const uint32_t* key_table, uint32_t* vals, int64_t n, c3d_stream_t stream) {
    hipStream_t s = (hipStream_t)stream;
    if (n <= 0 || nseg <= 0) return 0;
    if (!ranges || !key_table || !vals) { c3d_set_error("c3d_test_segment_sort_u32: NULL pointer"); return -1; }
    uint32_t* v1 = nullptr;
    C3D_CHECK(hipMalloc(&v1, 4 * (size_t)n));
    int rc = c3d_segment_sort_u32((const uint2*)ranges, nseg, key_table, vals, v1, s);
    (void)hipStreamSynchronize(s);
    (void)hipFree(v1);
    return rc;
}
int c3d_test_sort_pairs_u32(uint32_t* keys, uint32_t* vals, int64_t n, int32_t end_bit, c3d_stream_t stream) {
    hipStream_t s = (hipStream_t)stream;
    if (n <= 0) return 0;
    if (!keys || !vals) { c3d_set_error("c3d_test_sort_pairs_u32: NULL pointer"); return -1; }
    uint32_t *k1 = nullptr, *v1 = nullptr;
    void* tmp = nullptr;
    C3D_CHECK(hipMalloc(&k1, 4 * (size_t)n));
    C3D_CHECK(hipMalloc(&v1, 4 * (size_t)n));
    C3D_CHECK(hipMalloc(&tmp, c3d_sort_tmp_bytes((size_t)n)));
    int res = 0;
    int rc = c3d_sort_pairs_u32(keys, k1, vals, v1, false, (size_t)n, end_bit, tmp, &res, s);
    if (!rc && res == 1) {
        (void)hipMemcpyAsync(keys, k1, 4 * (size_t)n, hipMemcpyDeviceToDevice, s);
        (void)hipMemcpyAsync(vals, v1, 4 * (size_t)n, hipMemcpyDeviceToDevice, s);
    }
    (void)hipStreamSynchronize(s);
    (void)hipFree(k1); (void)hipFree(v1); (void)hipFree(tmp);
    return rc;
}

}  // extern "C"
