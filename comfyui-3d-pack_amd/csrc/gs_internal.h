// gs_internal.h -- internal types of the 3DGS rasterizer (not part of the C-ABI; see include/c3d_gs.h)
#pragma once
#include "c3d_common.h"
#include "scan_wave.h"

#define GS_PAIR_FLOATS 12   // per (tile, splat) gradient record: colour3, depth, mean2D2, conic3, opacity (+2 pad)

struct GsParams {
    int N, M, deg, W, H, gx, gy;
    float tanfovx, tanfovy, focal_x, focal_y, scale_modifier;
    float dscale_mod;    // factor on dL/dscale: 1 = as the dependency's backward (default), scale_modifier = exact derivative
    int rect4;           // the tile grid is at most 255 x 255: GsGeom::rect holds 4-byte packed rects (c3d_rect_pack)
    const float* bg;     // [3]   device
    const float* view;   // [16]  device, row-major storage of w2c^T (camera_utils.py:205)
    const float* proj;   // [16]  device, full projection, same storage (camera_utils.py:213)
    const float* campos; // [3]   device
};

// Per-Gaussian projected record ("geometry state"): ONE 64-byte line per Gaussian, three 16-B parts used
//   part 0 = (pix.x, pix.y, conic.xx, conic.xy)   part 1 = (conic.yy, opacity, r, g)   part 2 = (b, view depth, ex, ey)   part 3 = spare
// (ex, ey) = half extents, in pixels, of the axis-aligned box outside which alpha < 1/255 for certain
// (sqrt(2 ln(255 o) Sigma_xx), ... plus a safety margin): used to drop tiles / 8x8 quadrants that cannot contribute.
// Array-of-structures on purpose: the compositing kernels gather records by Gaussian id (random), and the memory system serves such
// gathers per 64-B line -- one line per splat instead of three.  rec0/rec1/rec2 point at parts 0/1/2 of record 0; index with GS_REC(i).
#define GS_REC(i) (4 * (size_t)(i))
struct GsGeom {
    float4* rec0;
    float4* rec1;
    float4* rec2;
    uint32_t* tiles;        // tiles touched per Gaussian (0 = culled)
    uint2* rect;            // the tile rect those tiles form: {x0 | y0 << 16, x1 | y1 << 16} (written with `tiles`; {0, 0} for culled Gaussians) -- or, GsParams::rect4, one packed uint32 per Gaussian
    uint32_t* key[2];       // depth-sort keys (float bits of view depth; 0xFFFFFFFF = culled)
    uint32_t* order[2];     // Gaussian ids, ping-pong; after stage 1 order[res] is rank -> id
    uint2* rsort;           // the tile rects in depth-rank order (left behind by the emit-offset scan, which gathers them; read by k_emit)
    uint32_t* offsets;      // inclusive scan of the tile counts in depth-rank order
    uint4* einfo;           // per Gaussian with tiles: {0, x0 | y0<<16, x1 | y1<<16, record base} of its tile rect -- written by the record-base scan (coalesced; the scan runs inside the recording forward compositing launch);
                            // GsParams::rect4: EIGHT bytes per Gaussian, {packed rect, record base} (scan_wave.h)
    uint32_t* rbase;        // exclusive scan of `tiles` in Gaussian-id order: where this Gaussian's backward gradient records start
    uint8_t* clamped;       // 3 bits per Gaussian: SH colour channel clamped at 0
    int* meta;              // [0] = min(num_rendered, capacity) (device copy)  [2] = error word of the binning chain (C3D_ERR_LOOKBACK)
    void* tmp;              // state of the three single-pass primitives of a view, side by side (ONE memset from `meta` clears them all):
    void* tmp_scan_a;       //   exclusive scan of `tiles` in Gaussian-id order -> rbase
    void* tmp_sort;         //   depth sort
    void* tmp_scan_b;       //   inclusive scan of the tile counts in depth-rank order -> offsets
    size_t zero_off;        // byte offset of `meta` inside the buffer
    size_t zero_bytes;      // bytes to clear, starting at `meta`
    size_t bytes;
};
static inline void gs_carve_geom(char* base, int N, GsGeom& g) {
    size_t n = (size_t)(N > 0 ? N : 1), off = 0;
    auto take = [&](size_t b) { char* p = base ? base + off : nullptr; off += c3d_align(b); return p; };
    g.rec0 = (float4*)take(64 * n);
    g.rec1 = g.rec0 ? g.rec0 + 1 : nullptr;
    g.rec2 = g.rec0 ? g.rec0 + 2 : nullptr;
    g.tiles = (uint32_t*)take(4 * n);
    g.rect = (uint2*)take(8 * n);
    g.key[0] = (uint32_t*)take(4 * n);
    g.key[1] = (uint32_t*)take(4 * n);
    g.order[0] = (uint32_t*)take(4 * n);
    g.order[1] = (uint32_t*)take(4 * n);
    g.rsort = (uint2*)take(8 * n);
    g.offsets = (uint32_t*)take(4 * n);
    g.einfo = (uint4*)take(16 * n);
    g.rbase = (uint32_t*)take(4 * n);
    g.clamped = (uint8_t*)take(n);
    const size_t meta_off = off;
    g.meta = (int*)take(64);
    g.tmp = g.tmp_scan_a = take(scan_wave_tmp_bytes(n));            // one-wave tiles: the scan rides in the recording forward compositing launch (scan_wave.h)
    g.tmp_scan_b = take(c3d_scan_tmp_bytes(n));
    g.tmp_sort = take(c3d_sort_tmp_bytes(n));                       // last: only its first c3d_sort_state_bytes need clearing
    g.zero_off = meta_off;
    g.zero_bytes = (off - c3d_sort_tmp_bytes(n)) - meta_off + c3d_sort_state_bytes(n, 32);
    g.bytes = off;
}

// Binning state (per tile-splat pair).  Pairs are emitted in depth-rank order (emit index e: the pairs of one
// Gaussian are contiguous in e, row-major over its tile rect), then stably sorted by tile id with the Gaussian
// id as payload: tval[res] is the per-tile, depth-ordered splat list.  The emit index of a (tile, Gaussian) pair
// is recomputed from GsGeom::einfo wherever it is needed.  The backward gradient records of a Gaussian sit at rbase[id] + (row-major position
// of the tile inside its rect): Gaussian-id order, so that the per-Gaussian pass streams them with consecutive lanes on consecutive records.
struct GsBinning {
    uint32_t* tkey[2];
    uint32_t* tval[2];
    uint2* ranges;   // [tiles] as the last tile-sort pass leaves them: {~start, end}, {0, 0} = an empty tile -- read through gs_tile_range
    int* meta;
    void* tmp;       // tile-sort state; `ranges`, `meta` and the state are adjacent: ONE memset of zero_bytes from `ranges` clears them
    size_t pair_stride;   // elements between the four byte planes of the pair-activity record (gs_pair_activity)
    size_t zero_off;      // byte offset of `ranges` inside the buffer
    size_t zero_bytes;
    size_t bytes;
};
static inline void gs_carve_binning(char* base, long long D, int tiles, GsBinning& b) {
    size_t d = (size_t)(D > 0 ? D : 1), off = 0;
    auto take = [&](size_t n) { char* p = base ? base + off : nullptr; off += c3d_align(n); return p; };
    b.tkey[0] = (uint32_t*)take(4 * d);
    b.tkey[1] = (uint32_t*)take(4 * d);
    b.pair_stride = c3d_align(4 * d) / 4;
    b.tval[0] = (uint32_t*)take(4 * d);
    b.tval[1] = (uint32_t*)take(4 * d);
    const size_t ranges_off = off;
    b.ranges = (uint2*)take(8 * (size_t)(tiles > 0 ? tiles : 1));
    b.meta = (int*)take(64);
    b.tmp = take(c3d_sort_tmp_bytes(d));
    int bits = 1;
    while ((1ll << bits) < (long long)tiles) bits++;
    b.zero_off = ranges_off;
    b.zero_bytes = (off - c3d_sort_tmp_bytes(d)) - ranges_off + c3d_sort_state_bytes(d, bits);
    b.bytes = off;
}

#ifdef __HIPCC__
// [start, end) of a tile in the sorted pair list from the stored words (both grown by atomicMax out of the cleared state {0, 0}, c3d_sort_pairs_u32)
__device__ __forceinline__ uint2 gs_tile_range(uint2 raw) { return raw.y ? make_uint2(~raw.x, raw.y) : make_uint2(0u, 0u); }
#endif

// Pair activity, written by the recording forward compositing and read by the backward one: FOUR byte planes of b.pair_stride bytes each, plane w
// byte i = quadrant w blended the splat at sorted list position i into at least one of its pixels.  A quadrant's wave writes its plane only for the
// list positions it walked; the backward pass reads plane w only below the deepest position a pixel of quadrant w reached (max n_contrib), which
// the wave always walked.  Lives in the key buffer the tile sort did NOT finish in (dead once the ranges are found; 4 bytes per pair).
static inline uint8_t* gs_pair_activity(const GsBinning& b, int res) { return (uint8_t*)b.tkey[1 - res]; }

struct GsImage {
    float* final_T;        // [H*W]
    uint32_t* n_contrib;   // [H*W]
    size_t bytes;
};
static inline void gs_carve_image(char* base, int W, int H, GsImage& im) {
    size_t p = (size_t)W * H, off = 0;
    if (p == 0) p = 1;
    auto take = [&](size_t n) { char* q = base ? base + off : nullptr; off += c3d_align(n); return q; };
    im.final_T = (float*)take(4 * p);
    im.n_contrib = (uint32_t*)take(4 * p);
    im.bytes = off;
}

// A8 over several views at once (fused training step): what the kernel needs of each view's state.  Passed by value (kernel argument).
#define GS_MAX_BWD_VIEWS 16
struct GsBwdView {
    const float* view; const float* proj; const float* campos;   // device matrices of the view
    const int* radii;
    const float4* rec0; const float4* rec1;
    const uint32_t* tiles; const uint32_t* rbase; const uint8_t* clamped;
    const float4* pairgrad;
    const uint8_t* pvalid;                                       // one byte per pair: 1 = the compositing backward wrote a record
    float* dmean2D;                                              // [N,3] per-view screen-space gradient (densification statistic)
    float* gcol;                                                 // [N,3] per-view dL/dcolour after the clamp mask (hand-over between the two A8 kernels)
    float tanfovx, tanfovy, focal_x, focal_y;
};
struct GsBwdViews { int V; GsBwdView v[GS_MAX_BWD_VIEWS]; };

// raw-parameter paths: launch a kernel template instance by the number of stored SH coefficients per channel (GsParams::M: 16 / 9 / 4 / 1 -> 3 (M - 1) floats of f_rest)
#define GS_BY_SH_COEFFS(M_, CALL_)                                                                                          \
    switch (M_) {                                                                                                           \
        case 16: CALL_(45); break;                                                                                          \
        case 9: CALL_(24); break;                                                                                           \
        case 4: CALL_(9); break;                                                                                            \
        case 1: CALL_(0); break;                                                                                            \
        default: c3d_set_error("c3d_gs: %d SH coefficients per channel (raw parameter paths take 1, 4, 9 or 16)", (int)(M_)); return -1; \
    }

// ---- multi-view launches (round 4) -----------------------------------------------------------------------------------------------------
// The views of a group (<= GS_MAX_GROUP) keep their state in workspace slices at a uniform stride `vs`; every stage of the chain is ONE launch with
// blockIdx.y = view, state pointers given for the group's first view.  What does not live in the slices -- the caller's output planes, loss targets,
// pixel gradients, the per-view background -- travels as a table of per-view pointers in the kernel arguments.
#define GS_MAX_GROUP GS_MAX_BWD_VIEWS
struct GsFwdViews { const float* bg[GS_MAX_GROUP]; float* color[GS_MAX_GROUP]; float* depth[GS_MAX_GROUP]; float* alpha[GS_MAX_GROUP]; };
// A7: pixel-space inputs of each view.  dcolor / ddepth / dalpha: caller-supplied gradients (NULL = zero).  The rest feeds the pixel loss of the fused training
// step (LOSS instances), evaluated in the kernel's prologue where every lane owns one pixel anyway:
//   L = scale * [ w_l1 mean|clamp(c) - t| m + w_l2 mean((clamp(c) - t) m)^2 + w_a mean(alpha - ta)^2 ],  m = cmask or 1
// its gradient is ADDED to what dcolor holds (the MS-SSIM term's gradient, when there is one).  tile_loss [tiles] (view 0's, in the slice): every tile WRITES
// its partial sum (8160 float atomics per view on one address cost the step 1.6 %); the sums are added in a fixed order afterwards: bit-reproducible.
struct GsBwdPix {
    const float* bg[GS_MAX_GROUP]; const float* dcolor[GS_MAX_GROUP]; const float* ddepth[GS_MAX_GROUP]; const float* dalpha[GS_MAX_GROUP];
    const float* color[GS_MAX_GROUP]; const float* alpha[GS_MAX_GROUP]; const float* tcolor[GS_MAX_GROUP]; const float* talpha[GS_MAX_GROUP]; const float* cmask[GS_MAX_GROUP];
};
struct GsPixelLossW { float w_l1, w_l2, w_a, scale; float* tile_loss; };

// kernels' launchers (gs_forward.hip / gs_backward.hip); V / vs: views of the launch and the byte stride between their slices (1 / 0 = one view)
int gs_launch_preprocess_bwd_views(const GsParams& p0, const GsBwdViews& views, const float* means3D, const float* f_dc, const float* f_rest,
                                   const float* scaling_raw, const float* rotation_raw, float* dL_dopacity_raw, float* dL_dmeans3D, float* dL_df_dc,
                                   float* dL_df_rest, float* dL_dscaling_raw, float* dL_drotation_raw, bool accumulate, hipStream_t s, uint32_t cap,
                                   int first = 0, int count = -1);   // Gaussians [first, first + count): a chunked pass (first % 4 == 0); default = all
int gs_launch_preprocess(const GsParams& p, const float* means3D, const float* shs, const float* colors_precomp,
                         const float* opacities, const float* scales, const float* rotations, const float* cov3D_precomp,
                         GsGeom& g, int* radii, hipStream_t s);
int gs_launch_preprocess_raw(const GsParams& p, const float* means3D, const float* f_dc, const float* f_rest, const float* opacity_raw,
                             const float* scaling_raw, const float* rotation_raw, GsGeom& g, int* radii, hipStream_t s);
int gs_launch_preprocess_views(const GsParams* views, int V, const GsGeom* geoms, int* const* radii, const float* means3D, const float* f_dc, const float* f_rest,
                               const float* opacity_raw, const float* scaling_raw, const float* rotation_raw, hipStream_t s);
int gs_launch_preprocess_bwd_raw(const GsParams& p, const GsGeom& g, const int* radii, const float* means3D, const float* f_dc, const float* f_rest,
                                 const float* scaling_raw, const float* rotation_raw, const float* pairgrad, const uint8_t* pvalid, float* dL_dmean2D,
                                 float* dL_dopacity_raw, float* dL_dmeans3D, float* dL_df_dc, float* dL_df_rest, float* dL_dscaling_raw,
                                 float* dL_drotation_raw, bool accumulate, hipStream_t s, uint32_t cap = 0xFFFFFFFFu);
int gs_launch_emit(const GsParams& p, const GsGeom& g, int res, GsBinning& b, hipStream_t s, uint32_t cap, int V, size_t vs, int passes);
int gs_launch_composite_fwd(const GsParams& p, const GsGeom& g, const GsBinning& b, int res, GsImage& im, const GsFwdViews& vp, int V, size_t vs,
                            bool record_activity, hipStream_t s, uint32_t* err = nullptr, bool scan = true);
int gs_launch_poison_on_overflow(const uint32_t* status, float* color, float* depth, float* alpha, int W, int H, hipStream_t s);
// status / status_host (device-visible address of 8 bytes of pinned host memory; optional): the final loss launch also leaves the step's status words on the host
int gs_launch_sum_group_loss(const float* terms, int n, int V, size_t vs, float* loss_out, hipStream_t s, const uint32_t* status = nullptr, unsigned long long* status_host = nullptr);   // ONE group holds all views: both stages below in one launch (same bits)
int gs_launch_sum_view_loss(const float* terms, int n, float* view_sum, int V, size_t vs, hipStream_t s);   // per view: its per-tile partials (+ MS-SSIM term) -> one float
int gs_launch_sum_tile_loss(const float* first_view_sum, size_t view_stride_bytes, int V, float* loss_out, hipStream_t s, const uint32_t* status = nullptr, unsigned long long* status_host = nullptr);   // the V view sums, in order, added to *loss_out
int gs_launch_composite_bwd(const GsParams& p, const GsGeom& g, const GsBinning& b, int res, const GsImage& im, const GsBwdPix& px, bool depth,
                            float* pairgrad /* [D][12] */, uint8_t* pvalid /* [D], clear on entry */, hipStream_t s, uint32_t cap = 0xFFFFFFFFu,
                            const GsPixelLossW* pixel_loss = nullptr, int V = 1, size_t vs = 0);
int gs_launch_preprocess_bwd(const GsParams& p, const GsGeom& g, const int* radii, const float* means3D, const float* shs,
                             const float* colors_precomp, const float* scales, const float* rotations, const float* cov3D_precomp,
                             const float* pairgrad, const uint8_t* pvalid, float* dL_dmean2D, float* dL_dcolors, float* dL_dopacity,
                             float* dL_dmeans3D, float* dL_dcov3D, float* dL_dsh, float* dL_dscales, float* dL_drots, hipStream_t s, uint32_t cap = 0xFFFFFFFFu);
int gs_launch_mark_visible(int N, const float* means3D, const float* view, const float* proj, uint8_t* present, hipStream_t s);
