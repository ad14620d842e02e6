// gs_api.hip -- extern "C" entry points declared in include/c3d_gs.h.
#include "../../include/c3d_gs.h"
#include "../../include/c3d_loss.h"
#include "gs_internal.h"
#include <stdarg.h>
#include <stdio.h>
#include <sched.h>
#include <time.h>

static thread_local char g_err[1024] = "";
void c3d_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

// msssim.hip: out_word += va + vb * mean MS-SSIM(x, y_eff), dL_dy (+)= grad_scale * d mean / dy
int ms_value_grad_images(const float* const* x, const float* const* y, const float* const* mask, float* const* dy, int clamp_y, int B, int C, int H, int W, float grad_scale,
                         int accumulate, float va, float vb, float* out0, size_t out_stride, void* workspace, hipStream_t s);
// record indices are bounded by the pair count the state buffers were sized for: the exact count of c3d_gs_forward_project, or the capacity of a sync-free forward
// (whose real count stays on the device and may -- reported through its status words -- have exceeded it)
static inline uint32_t pair_cap(int64_t num_rendered) { return num_rendered >= 0xFFFFFFFFll ? 0xFFFFFFFFu : (uint32_t)(num_rendered > 0 ? num_rendered : 0); }
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
    p.rect4 = (p.gx <= 255 && p.gy <= 255) ? 1 : 0;
    return 0;
}
// SH coefficients per channel the raw parameters store (c3d_gs_settings::sh_coeffs; 0 = 16); -1 + message if it is not one of 1, 4, 9, 16
static int raw_coeffs(const c3d_gs_settings* st) {
    if (!st) { c3d_set_error("c3d_gs: settings is NULL"); return -1; }
    const int k = st->sh_coeffs ? st->sh_coeffs : 16;
    if (k != 1 && k != 4 && k != 9 && k != 16) { c3d_set_error("c3d_gs: settings.sh_coeffs = %d (raw parameters store 1, 4, 9 or 16 SH coefficients per channel)", st->sh_coeffs); return -1; }
    if ((st->sh_degree + 1) * (st->sh_degree + 1) > k) { c3d_set_error("c3d_gs: sh_degree %d needs %d coefficients, the raw parameters store %d", st->sh_degree, (st->sh_degree + 1) * (st->sh_degree + 1), k); return -1; }
    return k;
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
// order, gather folded in).  Three single-pass primitives whose state is cleared by ONE memset (one view) or by the caller's c3d_zero_views
// (`cleared`, V views).  V / vs: the chain of V views in one launch per stage (g = view 0's state, view v's lies v * vs bytes behind).
// cap / status: pair capacity and status words of the sync-free paths (the pair count then stays on the device in g.meta[0]).
// (the record-base scan of the backward pass -- rbase, einfo -- is not on this chain any more: it rides in the recording forward compositing launch, scan_wave.h)
// hint / early: c3d_gs_forward_nosync -- the pair count its launches are sized for, and where the scan's tail leaves {bits, count} for the host (c3d_scan_rect_gather)
static int binning_front(const GsParams& p, GsGeom& g, int N, uint32_t cap, uint32_t* status, hipStream_t s, int V = 1, size_t vs = 0, bool cleared = false, uint32_t hint = 0,
                         unsigned long long* early = nullptr) {
    int rc, res = 0;
    if (!cleared) {
        if (V != 1) { c3d_set_error("internal: a multi-view binning chain needs its state cleared by the caller"); return -2; }
        C3D_CHECK(hipMemsetAsync(g.meta, 0, g.zero_bytes, s));
    }
    uint32_t* err = status ? status : (uint32_t*)g.meta + 2;      // a timed-out look-back (bounded spins) surfaces as C3D_ERR_LOOKBACK
    { C3dProfScope ps(C3D_P_DEPTH_SORT, s);
      if ((rc = c3d_sort_pairs_u32(g.key[0], g.key[1], g.order[0], g.order[1], true, (size_t)N, 32, g.tmp_sort, &res, s, nullptr, false, err, V, vs))) return rc; }
    if (res != sort_result_index(32)) { c3d_set_error("internal: depth sort buffer parity"); return -2; }
    { C3dProfScope ps(C3D_P_SCAN, s);
      if ((rc = c3d_scan_rect_gather(g.rect, g.order[res], g.offsets, g.rsort, (size_t)N, g.tmp_scan_b, s, false, (uint32_t*)g.meta, status, cap, err, V, vs, hint, early, p.rect4 != 0))) return rc; }
    return 0;
}
// emit + tile sort + per-tile ranges; D = pair count on the host, or the capacity when d_dev (device count, per view) is given
// D_hint (0 = none; one view, count on the device): the launches are sized for D_hint pairs although the buffers hold D -- workgroups loop when the count exceeds the hint,
// and the state is cleared for the count that is really there (c3d_sort_zero_state_counted) instead of a memset sized for D.
static int binning_back(const GsParams& p, GsGeom& g, GsBinning& b, long long D, uint32_t cap, const uint32_t* d_dev, uint32_t* status,
                        hipStream_t s, int* res_out, int V = 1, size_t vs = 0, bool cleared = false, long long D_hint = 0) {
    const int tiles = p.gx * p.gy;
    int rc, res = 0;
    const bool hinted = D_hint > 0 && D_hint < D && V == 1 && d_dev;
    if (hinted) {
        C3dProfScope ps(C3D_P_EMIT, s);      // (the clear belongs to the emission's group: no timing scope -- two event records -- of its own)
        if ((rc = c3d_sort_zero_state_counted(b.tmp, (size_t)D, tile_sort_bits(tiles), d_dev, (size_t)D_hint, b.ranges, (size_t)((char*)b.tmp - (char*)b.ranges), s))) return rc;
    } else if (!cleared) {
        if (V != 1) { c3d_set_error("internal: a multi-view binning chain needs its state cleared by the caller"); return -2; }
        C3D_CHECK(hipMemsetAsync(b.ranges, 0, b.zero_bytes, s));
    }
    *res_out = 0;
    if (D <= 0) return 0;
    uint32_t* err = status ? status : (uint32_t*)g.meta + 2;
    { C3dProfScope ps(C3D_P_EMIT, s);
      if ((rc = gs_launch_emit(p, g, sort_result_index(32), b, s, cap, V, vs, (tile_sort_bits(tiles) + 7) / 8))) return rc; }      // (also counts the digits of the keys it writes)
    { C3dProfScope ps(C3D_P_TILE_SORT, s);
      if ((rc = c3d_sort_pairs_u32(b.tkey[0], b.tkey[1], b.tval[0], b.tval[1], false, (size_t)D, tile_sort_bits(tiles), b.tmp, &res, s, d_dev, false, err, V, vs, true,
                                   hinted ? (size_t)D_hint : 0, b.ranges))) return rc; }      // (the last pass leaves the per-tile ranges: A5)
    if (res != sort_result_index(tile_sort_bits(tiles))) { c3d_set_error("internal: tile sort buffer parity"); return -2; }
    *res_out = res;
    return 0;
}

// drop-in path: the single D2H read of the pair count (the wheel has the same synchronisation) also brings back the error word
static int project_tail(const GsParams& p, GsGeom& g, int N, int64_t* num_rendered, hipStream_t s) {
    int rc;
    if ((rc = binning_front(p, g, N, 0xFFFFFFFFu, nullptr, s))) return rc;
    uint32_t host[4] = {0, 0, 0, 0};
    C3D_CHECK(hipMemcpyAsync(host, g.meta, sizeof(host), hipMemcpyDeviceToHost, s));
    C3D_CHECK(hipStreamSynchronize(s));
    if (host[2] & C3D_ERR_LOOKBACK) { c3d_set_error("c3d_gs: a chained-scan look-back timed out in the binning stage (device fault or a wedged workgroup)"); return -3; }
    *num_rendered = (int64_t)host[0];
    return 0;
}

extern "C" {

const char* c3d_last_error(void) { return g_err; }
int c3d_version(void) { return 600; }

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
    return project_tail(p, g, N, num_rendered, s);
}

int c3d_gs_forward_project_raw(const c3d_gs_settings* st, int32_t N, const float* means3D, const float* f_dc, const float* f_rest,
                               const float* opacity_raw, const float* scaling_raw, const float* rotation_raw, int32_t* radii,
                               void* geom_buffer, int64_t* num_rendered, c3d_stream_t stream) {
    hipStream_t s = (hipStream_t)stream;
    GsParams p;
    const int K = raw_coeffs(st);
    if (K < 0 || make_params(st, N, K, p)) return -1;
    if (!num_rendered) { c3d_set_error("c3d_gs_forward_project_raw: num_rendered (host) is NULL"); return -1; }
    *num_rendered = 0;
    if (N == 0) return 0;
    if (!means3D || !f_dc || (!f_rest && K > 1) || !opacity_raw || !scaling_raw || !rotation_raw || !radii || !geom_buffer) { c3d_set_error("c3d_gs_forward_project_raw: NULL pointer"); return -1; }
    if ((K > 1 && (uintptr_t)f_rest % 16) || (uintptr_t)rotation_raw % 16) { c3d_set_error("c3d_gs_forward_project_raw: f_rest / rotation must be 16-byte aligned"); return -1; }
    GsGeom g;
    gs_carve_geom((char*)geom_buffer, N, g);
    int rc;
    { C3dProfScope ps(C3D_P_PREPROCESS, s);
    if ((rc = gs_launch_preprocess_raw(p, means3D, f_dc, f_rest, opacity_raw, scaling_raw, rotation_raw, g, radii, s))) return rc; }
    return project_tail(p, g, N, num_rendered, s);
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
    const bool record = !(st->flags & C3D_GS_FLAG_FORWARD_ONLY);      // a backward call may follow: record the blended (quadrant, splat) pairs, run the record-base scan, keep final_T / n_contrib
    if ((num_rendered > 0 || (record && N > 0)) && (!geom_buffer || !radii)) { c3d_set_error("c3d_gs_forward_render: NULL geometry"); return -1; }      // (the record-base scan walks the geometry of all N Gaussians)
    if ((st->flags & C3D_GS_FLAG_KEEP_RECORD_BASES) && geom_buffer)      // a second rendering at the exact count: the device copy of the count (the backward pass clears its "record written" bytes by it) still holds the first attempt's capacity
        C3D_CHECK(hipMemsetD32Async((hipDeviceptr_t)g.meta, (int)(uint32_t)pair_cap(num_rendered), 1, s));
    if ((rc = binning_back(p, g, b, num_rendered, 0xFFFFFFFFu, nullptr, nullptr, s, &res))) return rc;
    C3dProfScope ps(C3D_P_COMPOSITE_FWD, s);
    GsFwdViews vp{};
    vp.bg[0] = p.bg; vp.color[0] = out_color; vp.depth[0] = out_depth; vp.alpha[0] = out_alpha;
    return gs_launch_composite_fwd(p, g, b, res, im, vp, 1, 0, record, s, nullptr, !(st->flags & C3D_GS_FLAG_KEEP_RECORD_BASES));
}

// A2-A6 of ONE view without the host.  The pair count stays on the device (g.meta[0]); the BUFFERS hold pair_capacity pairs, the LAUNCHES are sized for first_capacity (a hint
// learnt from earlier counts): a count beyond the hint is served by workgroups that loop (k_onesweep STAY == 2, the counted clear), so every count the buffers hold
// is rendered exactly, at no cost when the hint was right.  Only a view that needs more than pair_capacity is lost: C3D_ST_OVERFLOW, and NaN planes.
static int forward_tail_nosync(const c3d_gs_settings* st, const GsParams& p, GsGeom& g, int N, int64_t pair_capacity, int64_t first_capacity, void* binning_buffer, void* image_buffer,
                               float* out_color, float* out_depth, float* out_alpha, uint32_t* status, uint32_t* status_host, uint32_t* count_host, hipStream_t s) {
    const int tiles = p.gx * p.gy;
    const uint32_t cap = (uint32_t)pair_capacity;
    const long long hint = (first_capacity > 0 && first_capacity < pair_capacity) ? (long long)first_capacity : 0;
    const bool record = !(st->flags & C3D_GS_FLAG_FORWARD_ONLY);
    GsBinning b;
    gs_carve_binning((char*)binning_buffer, pair_capacity, tiles, b);
    GsImage im;
    gs_carve_image((char*)image_buffer, p.W, p.H, im);
    unsigned long long* early = nullptr;
    if (count_host) {      // pinned host memory the device can address: the scan's tail stores the count there itself (no copy in the stream between the scan and the emission)
        void* dp = nullptr;
        if (((uintptr_t)count_host & 7) || hipHostGetDevicePointer(&dp, count_host, 0) != hipSuccess || !dp) {
            (void)hipGetLastError();
            c3d_set_error("c3d_gs_forward_nosync: count_host must be 8-byte aligned pinned host memory mapped into the device's address space (hipHostMalloc / hipHostRegister)");
            return -1;
        }
        early = (unsigned long long*)dp;
    }
    int rc, res = 0;
    C3D_CHECK(hipMemsetAsync(status, 0, 2 * sizeof(uint32_t), s));
    if ((rc = binning_front(p, g, N, cap, status, s, 1, 0, false, (uint32_t)hint, early))) return rc;
    if ((rc = binning_back(p, g, b, pair_capacity, cap, (const uint32_t*)g.meta, status, s, &res, 1, 0, false, hint))) return rc;
    { C3dProfScope ps(C3D_P_COMPOSITE_FWD, s);
      GsFwdViews vp{};
      vp.bg[0] = p.bg; vp.color[0] = out_color; vp.depth[0] = out_depth; vp.alpha[0] = out_alpha;
      if ((rc = gs_launch_composite_fwd(p, g, b, res, im, vp, 1, 0, record, s, status))) return rc;
      // a caller that waits for the count (count_host) deals with a view beyond its buffers itself; one that does not must never see an image that merely looks plausible
      if (!count_host && (rc = gs_launch_poison_on_overflow(status, out_color, out_depth, out_alpha, p.W, p.H, s))) return rc; }
    // the copy rides behind the last launch in stream order and is nobody's critical path (the fault bit of a bounded wait can be raised by any kernel of the chain)
    if (status_host) C3D_CHECK(hipMemcpyAsync(status_host, status, 2 * sizeof(uint32_t), hipMemcpyDeviceToHost, s));
    return 0;
}
static int check_nosync_args(const char* who, int64_t pair_capacity, int64_t first_capacity, const void* radii, const void* geom_buffer, const void* binning_buffer, const void* image_buffer,
                             const float* out_color, const float* out_depth, const float* out_alpha, const uint32_t* status) {
    if (pair_capacity <= 0 || pair_capacity > (int64_t)0x3FFFFFF0ll) { c3d_set_error("%s: pair_capacity out of range (1 .. 2^30 - 16)", who); return -1; }
    if (first_capacity < 0) { c3d_set_error("%s: first_capacity is negative (0 or >= pair_capacity: launches sized for pair_capacity)", who); return -1; }
    if (!radii || !geom_buffer || !binning_buffer || !image_buffer || !out_color || !out_depth || !out_alpha || !status) { c3d_set_error("%s: NULL buffer", who); return -1; }
    return 0;
}

int c3d_gs_forward_nosync(const c3d_gs_settings* st, int32_t N, int32_t M, const float* means3D, const float* shs, const float* colors_precomp, const float* opacities,
                          const float* scales, const float* rotations, const float* cov3D_precomp, int32_t* radii, void* geom_buffer, int64_t pair_capacity,
                          int64_t first_capacity, void* binning_buffer, void* image_buffer, float* out_color, float* out_depth, float* out_alpha, uint32_t* status,
                          uint32_t* status_host, uint32_t* count_host, c3d_stream_t stream) {
    hipStream_t s = (hipStream_t)stream;
    GsParams p;
    if (make_params(st, N, M, p)) return -1;
    if (check_inputs(N, M, p.deg, means3D, shs, colors_precomp, scales, rotations, cov3D_precomp)) return -1;
    if (N == 0 || p.gx * p.gy == 0) { c3d_set_error("c3d_gs_forward_nosync: empty cloud / image (take c3d_gs_forward_project + c3d_gs_forward_render)"); return -1; }
    if (check_nosync_args("c3d_gs_forward_nosync", pair_capacity, first_capacity, radii, geom_buffer, binning_buffer, image_buffer, out_color, out_depth, out_alpha, status)) return -1;
    if (!opacities) { c3d_set_error("c3d_gs_forward_nosync: opacities is NULL"); return -1; }
    GsGeom g;
    gs_carve_geom((char*)geom_buffer, N, g);
    int rc;
    { C3dProfScope ps(C3D_P_PREPROCESS, s);
    if ((rc = gs_launch_preprocess(p, means3D, shs, colors_precomp, opacities, scales, rotations, cov3D_precomp, g, radii, s))) return rc; }
    return forward_tail_nosync(st, p, g, N, pair_capacity, first_capacity, binning_buffer, image_buffer, out_color, out_depth, out_alpha, status, status_host, count_host, s);
}

int c3d_gs_forward_raw_nosync(const c3d_gs_settings* st, int32_t N, const float* means3D, const float* f_dc, const float* f_rest, const float* opacity_raw,
                              const float* scaling_raw, const float* rotation_raw, int32_t* radii, void* geom_buffer, int64_t pair_capacity, int64_t first_capacity,
                              void* binning_buffer, void* image_buffer, float* out_color, float* out_depth, float* out_alpha, uint32_t* status, uint32_t* status_host,
                              uint32_t* count_host, c3d_stream_t stream) {
    hipStream_t s = (hipStream_t)stream;
    GsParams p;
    const int K = raw_coeffs(st);
    if (K < 0 || make_params(st, N, K, p)) return -1;
    if (N == 0 || p.gx * p.gy == 0) { c3d_set_error("c3d_gs_forward_raw_nosync: empty cloud / image (take c3d_gs_forward_project_raw + c3d_gs_forward_render)"); return -1; }
    if (check_nosync_args("c3d_gs_forward_raw_nosync", pair_capacity, first_capacity, radii, geom_buffer, binning_buffer, image_buffer, out_color, out_depth, out_alpha, status)) return -1;
    if (!means3D || !f_dc || (!f_rest && K > 1) || !opacity_raw || !scaling_raw || !rotation_raw) { c3d_set_error("c3d_gs_forward_raw_nosync: NULL pointer"); return -1; }
    if ((K > 1 && (uintptr_t)f_rest % 16) || (uintptr_t)rotation_raw % 16) { c3d_set_error("c3d_gs_forward_raw_nosync: f_rest / rotation must be 16-byte aligned"); return -1; }
    GsGeom g;
    gs_carve_geom((char*)geom_buffer, N, g);
    int rc;
    { C3dProfScope ps(C3D_P_PREPROCESS, s);
    if ((rc = gs_launch_preprocess_raw(p, means3D, f_dc, f_rest, opacity_raw, scaling_raw, rotation_raw, g, radii, s))) return rc; }
    return forward_tail_nosync(st, p, g, N, pair_capacity, first_capacity, binning_buffer, image_buffer, out_color, out_depth, out_alpha, status, status_host, count_host, s);
}

// the host's side of count_host: spin (the calling thread only; ctypes has released the GIL) until the scan of the call has stored its words
int c3d_gs_wait_count(const uint32_t* count_host, uint32_t sentinel, int64_t timeout_us, uint32_t* bits, uint32_t* count) {
    if (!count_host) { c3d_set_error("c3d_gs_wait_count: NULL pointer"); return -1; }
    const volatile unsigned long long* w = (const volatile unsigned long long*)count_host;
    struct timespec t0, t1;
    clock_gettime(CLOCK_MONOTONIC, &t0);
    for (unsigned spins = 0;; spins++) {
        const unsigned long long v = __atomic_load_n(w, __ATOMIC_ACQUIRE);
        if ((uint32_t)(v >> 32) != sentinel) {
            if (bits) *bits = (uint32_t)v;
            if (count) *count = (uint32_t)(v >> 32);
            return 0;
        }
        if ((spins & 63) == 63) {
            clock_gettime(CLOCK_MONOTONIC, &t1);
            const long long us = (t1.tv_sec - t0.tv_sec) * 1000000ll + (t1.tv_nsec - t0.tv_nsec) / 1000;
            if (timeout_us >= 0 && us > timeout_us) { c3d_set_error("c3d_gs_wait_count: timed out"); return -4; }
            // the word arrives within a view's worth of GPU work (<= ~1 ms on the per-view loops): spin, yielding -- a 20 us nap after 50 us of waiting, tried first, came back
            // 50-70 us late and cost the inference loop 8 % (profiles/r06/r06m_*).  Only a host that is a whole queue of kernels ahead (> 2 ms) stops burning its core on the poll
            if (us > 2000) { const struct timespec nap = {0, 50000}; nanosleep(&nap, nullptr); } else sched_yield();
        }
    }
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
    if (st->flags & C3D_GS_FLAG_FORWARD_ONLY) { c3d_set_error("c3d_gs_backward: these settings carry C3D_GS_FLAG_FORWARD_ONLY -- a forward pass rendered with them kept no state for a backward pass"); return -1; }
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
        if ((rc = c3d_zero_count(pvalid, (const uint32_t*)g.meta, pair_cap(num_rendered), s))) return rc;      // the pairs the forward pass worked on (<= num_rendered: the buffers may hold a capacity)
        GsBwdPix px{};
        px.bg[0] = p.bg; px.dcolor[0] = dL_dcolor; px.ddepth[0] = dL_ddepth; px.dalpha[0] = dL_dalpha;
        if ((rc = gs_launch_composite_bwd(p, g, b, res, im, px, dL_ddepth != nullptr, pairgrad, pvalid, s, pair_cap(num_rendered)))) return rc;
    }
    C3dProfScope ps(C3D_P_PREPROCESS_BWD, s);
    return gs_launch_preprocess_bwd(p, g, radii, means3D, shs, colors_precomp, scales, rotations, cov3D_precomp, pairgrad, pvalid, dL_dmeans2D,
                                    dL_dcolors, dL_dopacity, dL_dmeans3D, dL_dcov3D, dL_dsh, dL_dscales, dL_drotations, s, pair_cap(num_rendered));
}

int c3d_gs_backward_raw(const c3d_gs_settings* st, int32_t N, const float* means3D, const float* f_dc, const float* f_rest,
                        const float* scaling_raw, const float* rotation_raw, const int32_t* radii, const void* geom_buffer, int64_t num_rendered,
                        const void* binning_buffer, const void* image_buffer, const float* dL_dcolor, const float* dL_ddepth, const float* dL_dalpha,
                        float* dL_dmeans2D, float* dL_dmeans3D, float* dL_df_dc, float* dL_df_rest, float* dL_dopacity_raw, float* dL_dscaling_raw,
                        float* dL_drotation_raw, void* scratch, int32_t accumulate, c3d_stream_t stream) {
    hipStream_t s = (hipStream_t)stream;
    GsParams p;
    const int K = raw_coeffs(st);
    if (K < 0 || make_params(st, N, K, p)) return -1;
    if (st->flags & C3D_GS_FLAG_FORWARD_ONLY) { c3d_set_error("c3d_gs_backward_raw: these settings carry C3D_GS_FLAG_FORWARD_ONLY -- a forward pass rendered with them kept no state for a backward pass"); return -1; }
    if (N == 0) return 0;
    if (!means3D || !f_dc || (!f_rest && K > 1) || !scaling_raw || !rotation_raw || !radii || !geom_buffer || !dL_dcolor || !dL_dmeans2D || !dL_dmeans3D ||
        !dL_df_dc || (!dL_df_rest && K > 1) || !dL_dopacity_raw || !dL_dscaling_raw || !dL_drotation_raw || !scratch) { c3d_set_error("c3d_gs_backward_raw: NULL pointer"); return -1; }
    if ((K > 1 && ((uintptr_t)f_rest % 16 || (uintptr_t)dL_df_rest % 16)) || (uintptr_t)rotation_raw % 16 || (uintptr_t)dL_drotation_raw % 16) { c3d_set_error("c3d_gs_backward_raw: f_rest / rotation tensors must be 16-byte aligned"); return -1; }
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
        if ((rc = c3d_zero_count(pvalid, (const uint32_t*)g.meta, pair_cap(num_rendered), s))) return rc;      // the pairs the forward pass worked on (<= num_rendered: the buffers may hold a capacity)
        GsBwdPix px{};
        px.bg[0] = p.bg; px.dcolor[0] = dL_dcolor; px.ddepth[0] = dL_ddepth; px.dalpha[0] = dL_dalpha;
        if ((rc = gs_launch_composite_bwd(p, g, b, res, im, px, dL_ddepth != nullptr, pairgrad, pvalid, s, pair_cap(num_rendered)))) return rc;
    }
    C3dProfScope ps(C3D_P_PREPROCESS_BWD, s);
    return gs_launch_preprocess_bwd_raw(p, g, radii, means3D, f_dc, f_rest, scaling_raw, rotation_raw, pairgrad, pvalid, dL_dmeans2D, dL_dopacity_raw,
                                        dL_dmeans3D, dL_df_dc, dL_df_rest, dL_dscaling_raw, dL_drotation_raw, accumulate != 0, s, pair_cap(num_rendered));
}

// ---- fused multi-view paths (no host synchronisation inside) ----------------------------------------------------------------
struct StepWs {
    char* geom; char* binning; char* image; int* radii; float* color; float* depth; float* alpha; float* dcolor; float* pairgrad; uint8_t* pvalid;
    float* dmeans2D; float* gcol; float* tile_loss;
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
    w.dcolor = w.pairgrad = w.dmeans2D = w.gcol = w.tile_loss = nullptr; w.pvalid = nullptr;
    if (!fwd_only) {
        w.dcolor = (float*)take(12 * P);
        w.pairgrad = (float*)take(sizeof(float) * GS_PAIR_FLOATS * (size_t)(cap > 0 ? cap : 1));
        w.pvalid = (uint8_t*)take((size_t)(cap > 0 ? cap : 1));
        w.dmeans2D = (float*)take(12 * n);
        w.gcol = (float*)take(12 * n);
        w.tile_loss = (float*)take(4 * (size_t)(((W + C3D_TILE_X - 1) / C3D_TILE_X) * ((H + C3D_TILE_Y - 1) / C3D_TILE_Y) + 2));   // per-tile partial sums of the pixel loss + one slot for the view's MS-SSIM term + one for the view's sum
    }
    w.bytes = off;
}

// ---- view groups (round 4) ---------------------------------------------------------------------------------------------------------------------
// G consecutive views of a call (G <= GS_MAX_GROUP) whose state lives in G consecutive workspace slices go through every stage of the chain TOGETHER:
// ONE launch per stage with blockIdx.y = view -- projection (parameters streamed once for the group), one clear of all state blocks, scan, depth sort
// (histogram + 4 onesweep passes), scan, emit, tile sort (histogram + 2 passes), ranges, compositing forward [, pixel loss + compositing backward, loss sums]:
// 15 launches forward, 17 forward + backward, whatever G is.  Rounds 1-3 ran one such chain PER VIEW on a pool of streams ("view lanes") to hide the chain's
// latency-bound kernels (a 1 M-key radix pass is 16 MB of traffic and ~17 us of fixed latencies) under other views' compositing: ~170 launches per 8-view
// step and a 15-50 us hole at every kernel boundary of every chain.  A stage over 8 views is bandwidth-sized work instead.
// `lanes` survives as the number of GROUPS in flight: lanes = 1 puts all views of a step in one group on the caller's stream.
struct ViewGroup {
    int G = 0, N = 0, tiles = 0; size_t vs = 0; char* slice0 = nullptr; uint32_t cap = 0;
    GsParams p[GS_MAX_GROUP]; StepWs w[GS_MAX_GROUP]; GsGeom g0; GsBinning b0; GsImage im0;
};
static int group_setup(ViewGroup& q, const c3d_gs_settings* views, int G, int N, char* slice0, size_t vs, long long pair_capacity, bool fwd_only) {
    if (G < 1 || G > GS_MAX_GROUP) { c3d_set_error("internal: group of %d views", G); return -2; }
    q.G = G; q.N = N; q.vs = G > 1 ? vs : 0; q.slice0 = slice0; q.cap = (uint32_t)pair_capacity;
    for (int i = 0; i < G; i++) {
        const int K = raw_coeffs(&views[i]);
        if (K < 0 || make_params(&views[i], N, K, q.p[i])) return -1;
        carve_step(slice0 + (size_t)i * vs, N, q.p[i].H, q.p[i].W, pair_capacity, q.w[i], fwd_only);
    }
    q.tiles = q.p[0].gx * q.p[0].gy;
    gs_carve_geom(q.w[0].geom, N, q.g0);
    gs_carve_binning(q.w[0].binning, pair_capacity, q.tiles, q.b0);
    gs_carve_image(q.w[0].image, q.p[0].W, q.p[0].H, q.im0);
    return 0;
}
// A1 of the group: parameters read once, 93 B of projected state written per Gaussian and view.  radii_out[i] (optional): where view i's radii go instead of its slice.
static int group_project(ViewGroup& q, const float* means3D, const float* f_dc, const float* f_rest, const float* opacity_raw, const float* scaling_raw,
                         const float* rotation_raw, int32_t* const* radii_out, hipStream_t s) {
    GsGeom gs[GS_MAX_GROUP];
    int* rd[GS_MAX_GROUP];
    for (int i = 0; i < q.G; i++) {
        gs_carve_geom(q.w[i].geom, q.N, gs[i]);
        rd[i] = (radii_out && radii_out[i]) ? radii_out[i] : q.w[i].radii;
    }
    C3dProfScope sc(C3D_P_PREPROCESS, s);
    return gs_launch_preprocess_views(q.p, q.G, gs, rd, means3D, f_dc, f_rest, opacity_raw, scaling_raw, rotation_raw, s);
}
// A2-A5 of the group, sync-free: ONE clear of every view's state blocks (+ the backward pass's "record written" bytes), then one launch per stage
static int group_bin(ViewGroup& q, uint32_t* status, bool clear_pvalid, hipStream_t s, int* res_out) {
    int rc;
    {
        C3dProfScope ps(C3D_P_OTHER, s);
        const size_t off[3] = {(size_t)(q.w[0].geom - q.slice0) + q.g0.zero_off, (size_t)(q.w[0].binning - q.slice0) + q.b0.zero_off,
                               clear_pvalid ? (size_t)((char*)q.w[0].pvalid - q.slice0) : 0};
        const size_t bytes[3] = {q.g0.zero_bytes, q.b0.zero_bytes, clear_pvalid ? (((size_t)q.cap + 15) & ~(size_t)15) : 0};
        if ((rc = c3d_zero_views(q.slice0, q.vs, q.G, off, bytes, clear_pvalid ? 3 : 2, s))) return rc;
    }
    if ((rc = binning_front(q.p[0], q.g0, q.N, q.cap, status, s, q.G, q.vs, true))) return rc;
    return binning_back(q.p[0], q.g0, q.b0, (long long)q.cap, q.cap, (const uint32_t*)q.g0.meta, status, s, res_out, q.G, q.vs, true);
}
// A6 of the group.  out_*[i] (arrays or entries may be NULL): the caller's planes of view i; otherwise the slice's own.
// a view without a caller's depth plane gets none (nothing reads a slice's own: c3d_gs_step_read_view hands out radii and dL/dmeans2D); a group in which no view wants depth
// runs the kernel instance without the accumulator -- the fused training step, whose loss reads image and alpha only
static int group_composite_fwd(ViewGroup& q, int res, float* const* out_color, float* const* out_depth, float* const* out_alpha, bool record, hipStream_t s, uint32_t* status) {
    GsFwdViews vp{};
    for (int i = 0; i < q.G; i++) {
        vp.bg[i] = q.p[i].bg;
        vp.color[i] = (out_color && out_color[i]) ? out_color[i] : q.w[i].color;
        vp.depth[i] = (out_depth && out_depth[i]) ? out_depth[i] : nullptr;
        vp.alpha[i] = (out_alpha && out_alpha[i]) ? out_alpha[i] : q.w[i].alpha;
    }
    C3dProfScope ps(C3D_P_COMPOSITE_FWD, s);
    return gs_launch_composite_fwd(q.p[0], q.g0, q.b0, res, q.im0, vp, q.G, q.vs, record, s, status);
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
            const int K = raw_coeffs(&views[v0 + i]);
            if (K < 0 || make_params(&views[v0 + i], N, K, p)) return -1;
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
            (views[v].sh_coeffs ? views[v].sh_coeffs : 16) != (views[0].sh_coeffs ? views[0].sh_coeffs : 16) || views[v].scale_modifier != views[0].scale_modifier) { c3d_set_error("%s: all views must share resolution, sh_degree, sh_coeffs and scale_modifier", who); return -1; }
    return 0;
}
// views per group when V views are spread over `lanes` groups in flight
static int group_width(int V, int lanes) {
    int G = (V + lanes - 1) / lanes;
    return G > GS_MAX_GROUP ? GS_MAX_GROUP : (G < 1 ? 1 : G);
}

// training workspace of V views: V slices at a uniform stride, then the MS-SSIM workspace of the step's loss term (w_ssim != 0): one per-image share per view; a
// group of g views starting at view v0 uses the shares [v0, v0 + g) as ONE batched workspace (c3d_msssim_workspace_bytes(g, ..) <= g shares: the plan only pads less)
static size_t ms_share_bytes(int H, int W) { return (H > 160 && W > 160) ? c3d_msssim_workspace_bytes(1, 3, H, W) : 0; }
size_t c3d_gs_step_workspace_bytes(int32_t N, int32_t H, int32_t W, int64_t pair_capacity, int32_t views) {
    StepWs w; carve_step(nullptr, N, H, W, pair_capacity, w);
    return (size_t)(views > 0 ? views : 1) * (w.bytes + ms_share_bytes(H, W));
}
size_t c3d_gs_render_workspace_bytes(int32_t N, int32_t H, int32_t W, int64_t pair_capacity, int32_t slices) {
    StepWs w; carve_step(nullptr, N, H, W, pair_capacity, w, true);
    return (size_t)(slices > 0 ? slices : 1) * w.bytes;
}

int c3d_gs_train_views_raw(const c3d_gs_settings* views, int32_t V, int32_t N, const float* means3D, const float* f_dc, const float* f_rest,
                           const float* opacity_raw, const float* scaling_raw, const float* rotation_raw, const float* const* target_color,
                           const float* const* target_alpha, const float* const* color_mask, const c3d_gs_loss* loss, float* dL_dmeans3D, float* dL_df_dc, float* dL_df_rest,
                           float* dL_dopacity_raw, float* dL_dscaling_raw, float* dL_drotation_raw, float* loss_out, int64_t pair_capacity, int32_t lanes,
                           int32_t accumulate, void* workspace, uint32_t* status, uint32_t* status_host, c3d_stream_t stream) {
    hipStream_t s0 = (hipStream_t)stream;
    if (V <= 0 || N <= 0) return 0;
    unsigned long long* sh_dev = nullptr;
    if (status_host) {      // pinned host memory the device can address: the step's last loss launch stores the status words there itself (no copy in the stream)
        void* dp = nullptr;
        if (((uintptr_t)status_host & 7) || hipHostGetDevicePointer(&dp, status_host, 0) != hipSuccess || !dp) {
            (void)hipGetLastError();
            c3d_set_error("c3d_gs_train_views_raw: status_host must be 8-byte aligned pinned host memory mapped into the device's address space");
            return -1;
        }
        sh_dev = (unsigned long long*)dp;
    }
    if (!loss || !target_color || !status) { c3d_set_error("c3d_gs_train_views_raw: NULL pointer"); return -1; }
    if (check_step_args("c3d_gs_train_views_raw", views, V, pair_capacity, lanes, workspace)) return -1;
    const bool dc_only = views[0].sh_coeffs == 1;      // degree-0 storage: there is no f_rest
    if (!means3D || !f_dc || (!f_rest && !dc_only) || !opacity_raw || !scaling_raw || !rotation_raw || !dL_dmeans3D || !dL_df_dc || (!dL_df_rest && !dc_only) || !dL_dopacity_raw ||
        !dL_dscaling_raw || !dL_drotation_raw) { c3d_set_error("c3d_gs_train_views_raw: NULL parameter / gradient pointer"); return -1; }
    if ((!dc_only && ((uintptr_t)f_rest % 16 || (uintptr_t)dL_df_rest % 16)) || (uintptr_t)rotation_raw % 16 || (uintptr_t)dL_drotation_raw % 16) { c3d_set_error("c3d_gs_train_views_raw: f_rest / rotation tensors must be 16-byte aligned"); return -1; }
    for (int v = 0; v < V; v++) if (!target_color[v]) { c3d_set_error("c3d_gs_train_views_raw: target_color[%d] is NULL", v); return -1; }
    StepWs w0; carve_step(nullptr, N, views[0].image_height, views[0].image_width, pair_capacity, w0);
    const size_t vs = w0.bytes;
    const bool ssim = loss->w_ssim != 0.f;
    if (ssim && !(views[0].image_height > 160 && views[0].image_width > 160)) { c3d_set_error("c3d_gs_train_views_raw: the MS-SSIM term needs image sides > 160"); return -1; }
    const int G = group_width(V, lanes), groups = (V + G - 1) / G;
    int rc_all = 0;
    {
        // The groups follow each other on the caller's stream.  Measured and dropped (round 4, profiles/r04*): a two-stage pipeline with projection + binning of
        // group k + 1 on a high-priority stream underneath the compositing of group k -- 6.93 ms against 5.52 ms for the 8-view step (two groups of four), and no gain
        // forward-only: the chained-scan kernels of the sorts need their workgroups resident TOGETHER (a tile spins until its predecessors have published), and next to a
        // compositing grid that holds every wave slot they trickle in and wait for each other.  Rounds 1-3 hid the chains under other views' compositing at the price of a
        // hole per kernel boundary; one launch per stage for all views leaves neither the holes nor anything worth hiding.
        hipStream_t sf = s0, s = s0;
        for (int k = 0; k < groups && !rc_all; k++) {
            const int v0 = k * G, g = (V - v0) < G ? (V - v0) : G;
            ViewGroup q;
            int rc = 0, res = 0;
            do {
                if ((rc = group_setup(q, views + v0, g, N, (char*)workspace + (size_t)v0 * vs, vs, pair_capacity, false))) break;
                if ((rc = group_project(q, means3D, f_dc, f_rest, opacity_raw, scaling_raw, rotation_raw, nullptr, sf))) break;
                if ((rc = group_bin(q, status, true, sf, &res))) break;
                if ((rc = group_composite_fwd(q, res, nullptr, nullptr, nullptr, true, s, status))) break;
                // + scale * w_ssim * (1 - MS-SSIM(target * mask, clamp(C) * mask)) of every view (the batch mean of the reference, main_3DGS.py:192, is the mean of
                // the per-image values): the value goes into the view's own slot behind its tile partials, the gradient into the slice's dL/dcolor plane
                if (ssim) {      // all views of the group in one set of ~20 launches (the views' images, targets, masks and gradient planes as per-image pointer tables)
                    const float ws_ = loss->scale * loss->w_ssim;
                    const float* ys[GS_MAX_GROUP]; float* dys[GS_MAX_GROUP];
                    for (int i = 0; i < g; i++) { ys[i] = q.w[i].color; dys[i] = q.w[i].dcolor; }
                    char* ms_ws = (char*)workspace + (size_t)V * vs + (size_t)v0 * ms_share_bytes(q.p[0].H, q.p[0].W);
                    if ((rc = ms_value_grad_images(target_color + v0, ys, color_mask ? color_mask + v0 : nullptr, dys, 1, g, 3, q.p[0].H, q.p[0].W, -ws_, 0, ws_, -ws_,
                                                   loss_out ? q.w[0].tile_loss + q.tiles : nullptr, q.vs ? q.vs : vs, ms_ws, s))) break;
                }
                // pixel loss + backward down to the per-(tile, splat) records, all views of the group in one launch
                { C3dProfScope ps(C3D_P_COMPOSITE_BWD, s);
                  GsBwdPix px{};
                  for (int i = 0; i < g; i++) {
                      px.bg[i] = q.p[i].bg; px.dcolor[i] = ssim ? q.w[i].dcolor : nullptr; px.ddepth[i] = nullptr; px.dalpha[i] = nullptr;
                      px.color[i] = q.w[i].color; px.alpha[i] = q.w[i].alpha; px.tcolor[i] = target_color[v0 + i];
                      px.talpha[i] = target_alpha ? target_alpha[v0 + i] : nullptr; px.cmask[i] = color_mask ? color_mask[v0 + i] : nullptr;
                  }
                  const GsPixelLossW plw{loss->w_l1, loss->w_l2, loss->w_alpha_mse, loss->scale, loss_out ? q.w[0].tile_loss : nullptr};
                  if ((rc = gs_launch_composite_bwd(q.p[0], q.g0, q.b0, res, q.im0, px, false, q.w[0].pairgrad, q.w[0].pvalid, s, q.cap, &plw, q.G, q.vs))) break; }
                if (loss_out && groups > 1) { C3dProfScope ps(C3D_P_OTHER, s);
                                if ((rc = gs_launch_sum_view_loss(q.w[0].tile_loss, q.tiles + (ssim ? 1 : 0), q.w[0].tile_loss + q.tiles + 1, q.G, q.vs, s))) break; }
            } while (0);
            rc_all = rc;
        }
    }
    if (rc_all) return rc_all;
    if (loss_out) {   // the views' loss values -> loss_out, in view order
        StepWs wf; carve_step((char*)workspace, N, views[0].image_height, views[0].image_width, pair_capacity, wf);
        const int tiles = ((views[0].image_width + C3D_TILE_X - 1) / C3D_TILE_X) * ((views[0].image_height + C3D_TILE_Y - 1) / C3D_TILE_Y);
        C3dProfScope ps(C3D_P_OTHER, s0);
        if (groups == 1) { if (gs_launch_sum_group_loss(wf.tile_loss, tiles + (ssim ? 1 : 0), V, V > 1 ? vs : 0, loss_out, s0, status, sh_dev)) return -1; }      // one launch for both stages
        else if (gs_launch_sum_tile_loss(wf.tile_loss + tiles + 1, vs, V, loss_out, s0, status, sh_dev)) return -1;
    } else if (status_host) C3D_CHECK(hipMemcpyAsync(status_host, status, 2 * sizeof(uint32_t), hipMemcpyDeviceToHost, s0));      // (no loss launch to ride in)
    if (accumulate & 2) return 0;   // the caller runs the per-Gaussian pass itself, range by range (c3d_gs_step_param_backward_range)
    return step_a8_all_views(views, V, N, vs, workspace, pair_capacity, means3D, f_dc, f_rest, scaling_raw, rotation_raw, dL_dmeans3D, dL_df_dc, dL_df_rest,
                             dL_dopacity_raw, dL_dscaling_raw, dL_drotation_raw, (accumulate & 1) != 0, s0);
}

int c3d_gs_step_param_backward_range(const c3d_gs_settings* views, int32_t V, int32_t N, const float* means3D, const float* f_dc, const float* f_rest,
                                     const float* scaling_raw, const float* rotation_raw, float* dL_dmeans3D, float* dL_df_dc, float* dL_df_rest, float* dL_dopacity_raw,
                                     float* dL_dscaling_raw, float* dL_drotation_raw, int64_t pair_capacity, int32_t accumulate, void* workspace, int32_t first,
                                     int32_t count, c3d_stream_t stream) {
    if (V <= 0 || N <= 0 || count == 0) return 0;
    if (check_step_args("c3d_gs_step_param_backward_range", views, V, pair_capacity, 1, workspace)) return -1;
    const bool dc_only = views[0].sh_coeffs == 1;
    if (!means3D || !f_dc || (!f_rest && !dc_only) || !scaling_raw || !rotation_raw || !dL_dmeans3D || !dL_df_dc || (!dL_df_rest && !dc_only) || !dL_dopacity_raw || !dL_dscaling_raw || !dL_drotation_raw) {
        c3d_set_error("c3d_gs_step_param_backward_range: NULL parameter / gradient pointer"); return -1; }
    if ((!dc_only && ((uintptr_t)f_rest % 16 || (uintptr_t)dL_df_rest % 16)) || (uintptr_t)rotation_raw % 16 || (uintptr_t)dL_drotation_raw % 16) { c3d_set_error("c3d_gs_step_param_backward_range: f_rest / rotation tensors must be 16-byte aligned"); return -1; }
    if (first < 0 || count < 0 || (first & 3) || (long long)first + count > N) { c3d_set_error("c3d_gs_step_param_backward_range: range [%d, %d + %d) must lie inside [0, N) and start at a multiple of 4", first, first, count); return -1; }
    StepWs w0; carve_step(nullptr, N, views[0].image_height, views[0].image_width, pair_capacity, w0);
    return step_a8_all_views(views, V, N, w0.bytes, workspace, pair_capacity, means3D, f_dc, f_rest, scaling_raw, rotation_raw, dL_dmeans3D, dL_df_dc, dL_df_rest,
                             dL_dopacity_raw, dL_dscaling_raw, dL_drotation_raw, accumulate != 0, (hipStream_t)stream, first, count);
}

// forward of V views in groups.  keep_state: view v uses workspace slice v (what c3d_gs_backward_views_raw reads); otherwise the workspace holds `slices` forward-only
// slices and group k of a lane reuses that lane's slice set (stream order keeps the reuse safe).
static int views_forward(const char* who, const c3d_gs_settings* views, int32_t V, int32_t N, const float* means3D, const float* f_dc, const float* f_rest,
                         const float* opacity_raw, const float* scaling_raw, const float* rotation_raw, float* const* out_color, float* const* out_depth,
                         float* const* out_alpha, int32_t* const* out_radii, int64_t pair_capacity, int32_t lanes, void* workspace, uint32_t* status,
                         hipStream_t s0, bool keep_state, int slices) {
    if (V <= 0 || N <= 0) return 0;
    if (!out_color || !out_alpha || !status) { c3d_set_error("%s: NULL pointer", who); return -1; }
    if (check_step_args(who, views, V, pair_capacity, lanes, workspace)) return -1;
    if (!means3D || !f_dc || (!f_rest && views[0].sh_coeffs != 1) || !opacity_raw || !scaling_raw || !rotation_raw) { c3d_set_error("%s: NULL parameter pointer", who); return -1; }
    if ((views[0].sh_coeffs != 1 && (uintptr_t)f_rest % 16) || (uintptr_t)rotation_raw % 16) { c3d_set_error("%s: f_rest / rotation tensors must be 16-byte aligned", who); return -1; }
    for (int v = 0; v < V; v++) if (!out_color[v] || !out_alpha[v]) { c3d_set_error("%s: output %d is NULL", who, v); return -1; }
    const bool fwd_only = !keep_state;       // c3d_gs_render_views_raw: slices without the backward pass's buffers (c3d_gs_render_workspace_bytes)
    StepWs w0; carve_step(nullptr, N, views[0].image_height, views[0].image_width, pair_capacity, w0, fwd_only);
    const size_t vs = w0.bytes;
    // `lanes` groups, one after the other on the caller's stream (see c3d_gs_train_views_raw for the pipeline that was measured and dropped).  Forward only: every group
    // reuses the workspace's slices (stream order keeps that safe), so the slice count bounds the group width.
    int G = group_width(V, lanes);
    if (!keep_state && G > slices) G = slices < 1 ? 1 : slices;
    const int groups = (V + G - 1) / G;
    int rc_all = 0;
    hipStream_t s = s0;
    for (int k = 0; k < groups && !rc_all; k++) {
        const int v0 = k * G, g = (V - v0) < G ? (V - v0) : G;
        ViewGroup q;
        int res = 0;
        char* slice0 = (char*)workspace + (size_t)(keep_state ? v0 : 0) * vs;
        if ((rc_all = group_setup(q, views + v0, g, N, slice0, vs, pair_capacity, fwd_only))) break;
        // kept state: the backward pass reads the slice's copy of the radii; forward only: straight into the caller's buffer where there is one
        if ((rc_all = group_project(q, means3D, f_dc, f_rest, opacity_raw, scaling_raw, rotation_raw, (!keep_state && out_radii) ? out_radii + v0 : nullptr, s))) break;
        if ((rc_all = group_bin(q, status, keep_state, s, &res))) break;
        if ((rc_all = group_composite_fwd(q, res, out_color + v0, out_depth ? out_depth + v0 : nullptr, out_alpha + v0, keep_state, s, status))) break;
        for (int i = 0; i < g && keep_state && out_radii && !rc_all; i++)
            if (out_radii[v0 + i] && hipMemcpyAsync(out_radii[v0 + i], q.w[i].radii, sizeof(int) * (size_t)N, hipMemcpyDeviceToDevice, s) != hipSuccess) {
                c3d_set_error("%s: radii copy failed", who); rc_all = -1;
            }
    }
    return rc_all;
}

int c3d_gs_render_views_raw(const c3d_gs_settings* views, int32_t V, int32_t N, const float* means3D, const float* f_dc, const float* f_rest,
                             const float* opacity_raw, const float* scaling_raw, const float* rotation_raw, float* const* out_color, float* const* out_depth,
                             float* const* out_alpha, int32_t* const* out_radii, int64_t pair_capacity, int32_t lanes, void* workspace, int64_t workspace_bytes,
                             uint32_t* status, c3d_stream_t stream) {
    int slices = 1;
    if (V > 0 && N > 0 && views && lanes >= 1 && pair_capacity > 0) {      // the slice count decides the schedule: say what the buffer holds instead of trusting a convention
        const size_t one = c3d_gs_render_workspace_bytes(N, views[0].image_height, views[0].image_width, pair_capacity, 1);
        if (workspace_bytes < (int64_t)one) { c3d_set_error("c3d_gs_render_views_raw: workspace of %lld bytes holds less than one slice of %zu bytes", (long long)workspace_bytes, one); return -1; }
        const int64_t n = workspace_bytes / (int64_t)one;
        slices = n > 1024 ? 1024 : (int)n;
    }
    return views_forward("c3d_gs_render_views_raw", views, V, N, means3D, f_dc, f_rest, opacity_raw, scaling_raw, rotation_raw, out_color, out_depth, out_alpha, out_radii,
                         pair_capacity, lanes, workspace, status, (hipStream_t)stream, false, slices);
}

int c3d_gs_forward_views_raw(const c3d_gs_settings* views, int32_t V, int32_t N, const float* means3D, const float* f_dc, const float* f_rest,
                              const float* opacity_raw, const float* scaling_raw, const float* rotation_raw, float* const* out_color, float* const* out_depth,
                              float* const* out_alpha, int32_t* const* out_radii, int64_t pair_capacity, int32_t lanes, void* workspace, uint32_t* status,
                              c3d_stream_t stream) {
    return views_forward("c3d_gs_forward_views_raw", views, V, N, means3D, f_dc, f_rest, opacity_raw, scaling_raw, rotation_raw, out_color, out_depth, out_alpha, out_radii,
                         pair_capacity, lanes, workspace, status, (hipStream_t)stream, true, V);
}

int c3d_gs_backward_views_raw(const c3d_gs_settings* views, int32_t V, int32_t N, const float* means3D, const float* f_dc, const float* f_rest,
                               const float* scaling_raw, const float* rotation_raw, const float* const* dL_dcolor, const float* const* dL_ddepth,
                               const float* const* dL_dalpha, float* dL_dmeans3D, float* dL_df_dc, float* dL_df_rest, float* dL_dopacity_raw, float* dL_dscaling_raw,
                               float* dL_drotation_raw, int64_t pair_capacity, int32_t lanes, int32_t accumulate, void* workspace, c3d_stream_t stream) {
    hipStream_t s0 = (hipStream_t)stream;
    if (V <= 0 || N <= 0) return 0;
    if (!dL_dcolor) { c3d_set_error("c3d_gs_backward_views_raw: NULL pointer"); return -1; }
    if (check_step_args("c3d_gs_backward_views_raw", views, V, pair_capacity, lanes, workspace)) return -1;
    const bool dc_only = views[0].sh_coeffs == 1;
    if (!means3D || !f_dc || (!f_rest && !dc_only) || !scaling_raw || !rotation_raw || !dL_dmeans3D || !dL_df_dc || (!dL_df_rest && !dc_only) || !dL_dopacity_raw || !dL_dscaling_raw ||
        !dL_drotation_raw) { c3d_set_error("c3d_gs_backward_views_raw: NULL parameter / gradient pointer"); return -1; }
    if ((!dc_only && ((uintptr_t)f_rest % 16 || (uintptr_t)dL_df_rest % 16)) || (uintptr_t)rotation_raw % 16 || (uintptr_t)dL_drotation_raw % 16) { c3d_set_error("c3d_gs_backward_views_raw: f_rest / rotation tensors must be 16-byte aligned"); return -1; }
    for (int v = 0; v < V; v++) if (!dL_dcolor[v]) { c3d_set_error("c3d_gs_backward_views_raw: dL_dcolor[%d] is NULL", v); return -1; }
    StepWs w0; carve_step(nullptr, N, views[0].image_height, views[0].image_width, pair_capacity, w0);
    const size_t vs = w0.bytes;
    const int G = group_width(V, lanes), groups = (V + G - 1) / G;
    int rc_all = 0;
    {
        hipStream_t s = s0;      // the backward compositing kernels are VALU-bound: nothing to overlap, the groups follow each other on the caller's stream
        for (int k = 0; k < groups && !rc_all; k++) {
            const int v0 = k * G, g = (V - v0) < G ? (V - v0) : G;
            ViewGroup q;
            if ((rc_all = group_setup(q, views + v0, g, N, (char*)workspace + (size_t)v0 * vs, vs, pair_capacity, false))) break;
            {   // the compositing kernel marks the pairs it writes a record for and expects the bytes clear: a call in between on these slices (or an earlier backward
                // over different gradients) must not leave stale marks behind
                C3dProfScope pz(C3D_P_OTHER, s);
                const size_t off[1] = {(size_t)((char*)q.w[0].pvalid - q.slice0)}, bytes[1] = {((size_t)q.cap + 15) & ~(size_t)15};
                if ((rc_all = c3d_zero_views(q.slice0, q.vs, q.G, off, bytes, 1, s))) break;
            }
            GsBwdPix px{};
            bool depth = false;
            for (int i = 0; i < g; i++) {
                px.bg[i] = q.p[i].bg; px.dcolor[i] = dL_dcolor[v0 + i];
                px.ddepth[i] = dL_ddepth ? dL_ddepth[v0 + i] : nullptr; px.dalpha[i] = dL_dalpha ? dL_dalpha[v0 + i] : nullptr;
                depth = depth || px.ddepth[i];
            }
            C3dProfScope ps(C3D_P_COMPOSITE_BWD, s);
            rc_all = gs_launch_composite_bwd(q.p[0], q.g0, q.b0, sort_result_index(tile_sort_bits(q.tiles)), q.im0, px, depth, q.w[0].pairgrad, q.w[0].pvalid, s, q.cap, nullptr, q.G, q.vs);
        }
    }
    if (rc_all) return rc_all;
    return step_a8_all_views(views, V, N, vs, workspace, pair_capacity, means3D, f_dc, f_rest, scaling_raw, rotation_raw, dL_dmeans3D, dL_df_dc, dL_df_rest,
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

// the reference's add_densification_stats + max_radii2D update (main_3DGS.py:210-213, main_3DGS_renderer.py:767-769) for ONE view of the last step, read straight out of
// the step's workspace: per visible Gaussian (radius > 0)  grad_accum += |dL/dmean2D.xy|, denom += 1, max_radii = max(max_radii, radius).  One launch instead of two
// device copies and nine torch ops per training iteration of the densification window (the node's default run is bound by its launch count).
__global__ void __launch_bounds__(256) k_densify_stats(int N, const int* __restrict__ radii, const float* __restrict__ d2, float* __restrict__ grad_accum, float* __restrict__ denom,
                                                        float* __restrict__ max_radii) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    const int r = radii[i];
    if (r <= 0) return;
    const float gx = d2[3 * (size_t)i], gy = d2[3 * (size_t)i + 1];
    grad_accum[i] += sqrtf(gx * gx + gy * gy);
    denom[i] += 1.f;
    if (max_radii) max_radii[i] = fmaxf(max_radii[i], (float)r);
}
int c3d_gs_step_accumulate_densify_stats(int32_t N, int32_t H, int32_t W, int64_t pair_capacity, const void* workspace, int32_t view, float* grad_accum, float* denom,
                                         float* max_radii, c3d_stream_t stream) {
    if (N <= 0) return 0;
    if (!workspace || view < 0 || !grad_accum || !denom) { c3d_set_error("c3d_gs_step_accumulate_densify_stats: bad argument"); return -1; }
    StepWs w0; carve_step(nullptr, N, H, W, pair_capacity, w0);
    StepWs w; carve_step((char*)workspace + (size_t)view * w0.bytes, N, H, W, pair_capacity, w);
    hipStream_t s = (hipStream_t)stream;
    C3dProfScope ps(C3D_P_OTHER, s);
    hipLaunchKernelGGL(k_densify_stats, dim3(c3d_cdiv(N, 256)), dim3(256), 0, s, N, (const int*)w.radii, (const float*)w.dmeans2D, grad_accum, denom, max_radii);
    C3D_LAUNCH_CHECK();
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
__global__ void k_debug_ranges(int tiles, const uint2* __restrict__ raw, uint2* __restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < tiles) out[i] = gs_tile_range(raw[i]);
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
        if (ranges) { hipLaunchKernelGGL(k_debug_ranges, dim3(c3d_cdiv(tiles, 256)), dim3(256), 0, s, tiles, (const uint2*)b.ranges, (uint2*)ranges); C3D_LAUNCH_CHECK(); }
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
// the record-base scan that rides in the recording forward compositing launch (scan_wave.h), as a launch of its own: out[i] = exclusive prefix of in, einfo[i] = {0, rect[i], out[i]} where in[i] != 0
__global__ void __launch_bounds__(64) k_test_scan_wave(ScanWaveJob sj) { scan_wave_tile(sj, 0); }
int c3d_test_scan_wave(const uint32_t* in, const uint32_t* rect, uint32_t* out, uint32_t* einfo, int64_t n, c3d_stream_t stream) {
    hipStream_t s = (hipStream_t)stream;
    if (n <= 0) return 0;
    if (!in || !rect || !out || !einfo || n >= (1ll << 32)) { c3d_set_error("c3d_test_scan_wave: bad argument"); return -1; }
    void* tmp = nullptr;
    C3D_CHECK(hipMalloc(&tmp, scan_wave_tmp_bytes((size_t)n) + 64));
    C3D_CHECK(hipMemsetAsync(tmp, 0, scan_wave_tmp_bytes((size_t)n) + 64, s));
    uint32_t* err = (uint32_t*)((char*)tmp + scan_wave_tmp_bytes((size_t)n));
    const ScanWaveJob sj{in, out, (const uint2*)rect, (uint4*)einfo, (uint32_t*)tmp, err, (uint32_t)n, scan_wave_blocks((size_t)n), 0};
    hipLaunchKernelGGL(k_test_scan_wave, dim3(sj.blocks), dim3(64), 0, s, sj);
    uint32_t e = 0;
    hipError_t rc = hipMemcpyAsync(&e, err, 4, hipMemcpyDeviceToHost, s);
    if (rc == hipSuccess) rc = hipStreamSynchronize(s);
    (void)hipFree(tmp);
    if (rc != hipSuccess) { c3d_set_error("c3d_test_scan_wave: %s", hipGetErrorString(rc)); return (int)rc; }
    if (e) { c3d_set_error("c3d_test_scan_wave: a hand-over timed out"); return -3; }
    return 0;
}
// keys sorted in place, vals = the permutation (element indices): the form the depth sort and the topology builders use -- up to 16384 keys it is ONE launch of ONE workgroup (k_sort_small)
int c3d_test_sort_iota_u32(uint32_t* keys, uint32_t* vals, int64_t n, int32_t end_bit, c3d_stream_t stream) {
    hipStream_t s = (hipStream_t)stream;
    if (n <= 0) return 0;
    if (!keys || !vals) { c3d_set_error("c3d_test_sort_iota_u32: NULL pointer"); return -1; }
    uint32_t *k1 = nullptr, *v1 = nullptr;
    void* tmp = nullptr;
    C3D_CHECK(hipMalloc(&k1, 4 * (size_t)n));
    C3D_CHECK(hipMalloc(&v1, 4 * (size_t)n));
    C3D_CHECK(hipMalloc(&tmp, c3d_sort_tmp_bytes((size_t)n)));
    int res = 0;
    int rc = c3d_sort_pairs_u32(keys, k1, vals, v1, true, (size_t)n, end_bit, tmp, &res, s);
    if (!rc && res == 1) {
        (void)hipMemcpyAsync(keys, k1, 4 * (size_t)n, hipMemcpyDeviceToDevice, s);
        (void)hipMemcpyAsync(vals, v1, 4 * (size_t)n, hipMemcpyDeviceToDevice, s);
    }
    (void)hipStreamSynchronize(s);
    (void)hipFree(k1); (void)hipFree(v1); (void)hipFree(tmp);
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
