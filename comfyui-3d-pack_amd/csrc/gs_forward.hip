// gs_forward.hip -- 3DGS forward: preprocess (A1), tile binning (A2-A5) and per-tile compositing (A6).
// Stage names follow SURVEY.md section 2.3-A; the arithmetic contract is Appendix A of that file.
// Boundary this implements: the rasterizer call at
// MVs_Algorithms/GaussianSplatting/main_3DGS_renderer.py:927-936 (reference).
#include "gs_internal.h"
#include "gs_math.h"

// ------------------------------------------------------------------------------------------
// A1, the part that depends on the view: projection, EWA covariance, radius, tile rect, SH colour of ONE Gaussian (mean m, 3D covariance c3,
// opacity as given by `opacity()`, SH row `sh` with stride 3 per coefficient) -> the projected state arrays of that view.  Shared by the
// per-view kernel and the all-views-of-a-step kernel, so the two cannot drift apart.
// ------------------------------------------------------------------------------------------
struct GsPreCam { const float* view; const float* proj; const float* campos; float tanfovx, tanfovy, focal_x, focal_y; };
struct GsPreOut { float4* rec0; uint32_t* tiles; uint2* rect; uint32_t* key0; uint8_t* clamped; int* radii; };
template <class OpacityFn>
__device__ __forceinline__ void gs_project_one(int idx, const float3 m, const float c3[6], OpacityFn opacity, const float* __restrict__ sh, const float* __restrict__ colors_precomp,
                                               const GsPreCam& cam, int W, int H, int gx, int gy, int deg, const GsPreOut& o) {
    int rad = 0;
    uint32_t nt = 0, key = 0xFFFFFFFFu;
    do {   // `break` = culled: radius 0, no tiles, key 0xFFFFFFFF
        const Mat16 V = load_mat16(cam.view), PJ = load_mat16(cam.proj);
        const float3 pv = xform4x3(m, V);
        if (pv.z <= 0.2f) break;
        const float4 ph = xform4x4(m, PJ);
        const float pw = 1.0f / (ph.w + 0.0000001f);
        const float ppx = ph.x * pw, ppy = ph.y * pw;
        float T2[2][3], ST0[3], ST1[3];
        float3 t; bool xin, yin;
        ewa_T2(m, V, cam.tanfovx, cam.tanfovy, cam.focal_x, cam.focal_y, T2, t, xin, yin);
        sigma_T(c3, T2, ST0, ST1);
        const float a = T2[0][0] * ST0[0] + T2[0][1] * ST0[1] + T2[0][2] * ST0[2] + 0.3f;
        const float b = T2[0][0] * ST1[0] + T2[0][1] * ST1[1] + T2[0][2] * ST1[2];
        const float c = T2[1][0] * ST1[0] + T2[1][1] * ST1[1] + T2[1][2] * ST1[2] + 0.3f;
        const float det = a * c - b * b;
        if (det == 0.0f) break;
        const float di = 1.f / det;
        const float mid = 0.5f * (a + c);
        const float sq = sqrtf(fmaxf(0.1f, mid * mid - det));
        const int r = (int)ceilf(3.f * sqrtf(fmaxf(mid + sq, mid - sq)));
        const float px = ndc2pix(ppx, W), py = ndc2pix(ppy, H);
        int x0, y0, x1, y1;
        tile_rect(px, py, r, gx, gy, x0, y0, x1, y1);
        if ((x1 - x0) * (y1 - y0) == 0) break;
        // from here on the Gaussian counts as visible (radii > 0), exactly as in the dependency; the tile list
        // it is emitted to is narrowed to the tiles where alpha can reach 1/255 (exact, see tile_rect_tight)
        const float opac = opacity();
        const float ex = alpha_extent(opac, a), ey = alpha_extent(opac, c);
        if (ex >= 0.f) tile_rect_tight(px, py, r, ex, ey, gx, gy, x0, y0, x1, y1);
        else { x1 = x0; y1 = y0; }
        float r0 = 0.f, r1 = 0.f, r2 = 0.f;
        uint8_t cl = 0;
        if (colors_precomp) {
            r0 = colors_precomp[3 * idx]; r1 = colors_precomp[3 * idx + 1]; r2 = colors_precomp[3 * idx + 2];
        } else {
            float dx = m.x - cam.campos[0], dy = m.y - cam.campos[1], dz = m.z - cam.campos[2];
            const float len = sqrtf(dx * dx + dy * dy + dz * dz);
            dx /= len; dy /= len; dz /= len;
#define GS_FWD_TERM(k, Bk, dBx, dBy, dBz)                                                             \
    {                                                                                                 \
        const float b_ = (Bk);                                                                        \
        r0 += b_ * sh[3 * (k)]; r1 += b_ * sh[3 * (k) + 1]; r2 += b_ * sh[3 * (k) + 2];               \
    }
            SH_FOREACH(deg, dx, dy, dz, GS_FWD_TERM);
#undef GS_FWD_TERM
            r0 += 0.5f; r1 += 0.5f; r2 += 0.5f;
            if (r0 < 0.f) cl |= 1; if (r1 < 0.f) cl |= 2; if (r2 < 0.f) cl |= 4;
            r0 = fmaxf(r0, 0.f); r1 = fmaxf(r1, 0.f); r2 = fmaxf(r2, 0.f);
        }
        o.rec0[GS_REC(idx)] = make_float4(px, py, c * di, -b * di);
        o.rec0[GS_REC(idx) + 1] = make_float4(a * di, opac, r0, r1);
        o.rec0[GS_REC(idx) + 2] = make_float4(r2, pv.z, ex, ey);
        o.clamped[idx] = cl;
        rad = r;
        nt = (uint32_t)((x1 - x0) * (y1 - y0));
        o.rect[idx] = make_uint2((uint32_t)x0 | ((uint32_t)y0 << 16), (uint32_t)x1 | ((uint32_t)y1 << 16));
        key = nt ? __float_as_uint(pv.z) : 0xFFFFFFFFu;
    } while (false);
    o.radii[idx] = rad;
    o.tiles[idx] = nt;
    o.key0[idx] = key;
}

// ------------------------------------------------------------------------------------------
// A1 preprocess: one lane per Gaussian.  Streams xyz/scale/rot/opacity/SH once, writes the 48-B
// projected record, the depth-sort key and the tile count.  STAGED: the workgroup's 48 KiB of SH
// coefficients are brought in with fully coalesced 16-B loads through LDS (sh_stage_in) instead of
// 64 lanes striding 192 B apart.
// ------------------------------------------------------------------------------------------
// RAW: the inputs are the reference's raw parameters -- scales = exp(.), opacity = sigmoid(.), rotation = normalize(.) are applied here
// (GaussianModel accessors, main_3DGS_renderer.py:294-321) and SH comes as the split f_dc / f_rest pair (`shs` = f_dc, `f_rest` extra).
template <bool STAGED, bool RAW>
__global__ void __launch_bounds__(256) k_preprocess(GsParams p, const float* __restrict__ means3D, const float* __restrict__ shs,
                                                     const float* __restrict__ f_rest,
                                                     const float* __restrict__ colors_precomp, const float* __restrict__ opacities,
                                                     const float* __restrict__ scales, const float* __restrict__ rotations,
                                                     const float* __restrict__ cov3D_precomp, GsGeom g, int* __restrict__ radii) {
    extern __shared__ float sh_lds[];
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (STAGED) {
        const size_t g0 = (size_t)blockIdx.x * blockDim.x;
        if (RAW) sh_stage_in_split(shs, f_rest, g0, min((int)blockDim.x, p.N - (int)g0), sh_lds);
        else     sh_stage_in(shs, g0, min((int)blockDim.x, p.N - (int)g0), sh_lds);
        __syncthreads();
    }
    if (idx >= p.N) return;
    const float3 m = make_float3(means3D[3 * idx], means3D[3 * idx + 1], means3D[3 * idx + 2]);
    float c3[6];
    if (cov3D_precomp) {
#pragma unroll
        for (int i = 0; i < 6; i++) c3[i] = cov3D_precomp[6 * idx + i];
    } else {
        float3 s = make_float3(scales[3 * idx], scales[3 * idx + 1], scales[3 * idx + 2]);
        float4 q = *reinterpret_cast<const float4*>(rotations + 4 * idx);
        if (RAW) {
            s = make_float3(expf(s.x), expf(s.y), expf(s.z));
            const float inv = 1.f / fmaxf(sqrtf(q.x * q.x + q.y * q.y + q.z * q.z + q.w * q.w), 1e-12f);
            q = make_float4(q.x * inv, q.y * inv, q.z * inv, q.w * inv);
        }
        cov3d_from_scale_rot(s, p.scale_modifier, q, c3);
    }
    const float* sh = STAGED ? sh_lds + threadIdx.x * SH_ROW : shs + (size_t)idx * p.M * 3;
    gs_project_one(idx, m, c3, [&]() { return RAW ? 1.f / (1.f + expf(-opacities[idx])) : opacities[idx]; }, sh, colors_precomp,
                   GsPreCam{p.view, p.proj, p.campos, p.tanfovx, p.tanfovy, p.focal_x, p.focal_y}, p.W, p.H, p.gx, p.gy, p.deg,
                   GsPreOut{g.rec0, g.tiles, g.rect, g.key[0], g.clamped, radii});
}

int gs_launch_preprocess(const GsParams& p, const float* means3D, const float* shs, const float* colors_precomp,
                         const float* opacities, const float* scales, const float* rotations, const float* cov3D_precomp,
                         GsGeom& g, int* radii, hipStream_t s) {
    if (p.N == 0) return 0;
    const bool staged = shs && !colors_precomp && p.M == 16 && ((uintptr_t)shs % 16 == 0);
    if (staged)
        hipLaunchKernelGGL((k_preprocess<true, false>), dim3(c3d_cdiv(p.N, 256)), dim3(256), 256 * SH_ROW * sizeof(float), s, p, means3D, shs,
                           (const float*)nullptr, colors_precomp, opacities, scales, rotations, cov3D_precomp, g, radii);
    else
        hipLaunchKernelGGL((k_preprocess<false, false>), dim3(c3d_cdiv(p.N, 256)), dim3(256), 0, s, p, means3D, shs, (const float*)nullptr,
                           colors_precomp, opacities, scales, rotations, cov3D_precomp, g, radii);
    C3D_LAUNCH_CHECK();
    return 0;
}
// ---- A1 for all views of a step at once ---------------------------------------------------------------------------------------------------
// The per-view kernel streams 44 + 192 B of parameters per Gaussian for every view; over the 8 views of a training step that is 1.9 GB of the
// same 236 MB.  Here a lane keeps its Gaussian (mean, 3D covariance, opacity in registers, the 48 SH coefficients in LDS) and walks the views:
// parameters are read once per step, what is left per view is the 93 B of projected state it writes.  Same arithmetic, statement by statement,
// as k_preprocess<true, true> (the per-view paths and the tests hold the two together).
struct GsPreView { GsPreCam cam; GsPreOut out; };
struct GsPreViews { int V; GsPreView v[GS_MAX_BWD_VIEWS]; };
__global__ void __launch_bounds__(128) k_preprocess_views(GsParams p, GsPreViews vs, const float* __restrict__ means3D, const float* __restrict__ f_dc,
                                                           const float* __restrict__ f_rest, const float* __restrict__ opacities,
                                                           const float* __restrict__ scales, const float* __restrict__ rotations) {
    extern __shared__ float sh_lds[];
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    {
        const size_t g0 = (size_t)blockIdx.x * blockDim.x;
        sh_stage_in_split(f_dc, f_rest, g0, min((int)blockDim.x, p.N - (int)g0), sh_lds);
        __syncthreads();
    }
    if (idx >= p.N) return;
    const float3 m = make_float3(means3D[3 * idx], means3D[3 * idx + 1], means3D[3 * idx + 2]);
    float c3[6];
    {
        float3 s = make_float3(scales[3 * idx], scales[3 * idx + 1], scales[3 * idx + 2]);
        float4 q = *reinterpret_cast<const float4*>(rotations + 4 * idx);
        s = make_float3(expf(s.x), expf(s.y), expf(s.z));
        const float inv = 1.f / fmaxf(sqrtf(q.x * q.x + q.y * q.y + q.z * q.z + q.w * q.w), 1e-12f);
        q = make_float4(q.x * inv, q.y * inv, q.z * inv, q.w * inv);
        cov3d_from_scale_rot(s, p.scale_modifier, q, c3);
    }
    const float opac = 1.f / (1.f + expf(-opacities[idx]));
    const float* sh = sh_lds + threadIdx.x * SH_ROW;
    for (int v = 0; v < vs.V; v++)
        gs_project_one(idx, m, c3, [&]() { return opac; }, sh, (const float*)nullptr, vs.v[v].cam, p.W, p.H, p.gx, p.gy, p.deg, vs.v[v].out);
}
// Round 3: the same kernel with the lane's 48 SH coefficients in REGISTERS.  The LDS image above (196 B per lane) caps the kernel at 13 waves per CU (3 per SIMD)
// -- half of its time is VALU work, half memory, and with so few waves the two do not overlap.  Here LDS is only the transposition buffer of the coalesced load
// (44 rows at a time: 8.6 KB per workgroup), the coefficients move on into registers (28 + 48 VGPRs -> 6 waves per SIMD).  Same arithmetic: gs_project_one reads
// sh[k] either way.
#define PRE_ROWS 44      // multiple of 4: the 16-byte loads of sh_stage_in_split stay aligned
__global__ void __launch_bounds__(128) k_preprocess_views_r(GsParams p, GsPreViews vs, const float* __restrict__ means3D, const float* __restrict__ f_dc,
                                                             const float* __restrict__ f_rest, const float* __restrict__ opacities,
                                                             const float* __restrict__ scales, const float* __restrict__ rotations) {
    __shared__ float sh_lds[PRE_ROWS * SH_ROW];
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    float sh[SH_M3];
#pragma unroll
    for (int k = 0; k < SH_M3; k++) sh[k] = 0.f;
    {
        const size_t g0 = (size_t)blockIdx.x * blockDim.x;
        const int count = min((int)blockDim.x, p.N - (int)g0);
        for (int r0 = 0; r0 < count; r0 += PRE_ROWS) {
            const int rows = min(PRE_ROWS, count - r0);
            sh_stage_in_split(f_dc, f_rest, g0 + r0, rows, sh_lds);
            __syncthreads();
            const int mine = (int)threadIdx.x - r0;
            if (mine >= 0 && mine < rows) {
#pragma unroll
                for (int k = 0; k < SH_M3; k++) sh[k] = sh_lds[mine * SH_ROW + k];
            }
            __syncthreads();
        }
    }
    if (idx >= p.N) return;
    const float3 m = make_float3(means3D[3 * idx], means3D[3 * idx + 1], means3D[3 * idx + 2]);
    float c3[6];
    {
        float3 s = make_float3(scales[3 * idx], scales[3 * idx + 1], scales[3 * idx + 2]);
        float4 q = *reinterpret_cast<const float4*>(rotations + 4 * idx);
        s = make_float3(expf(s.x), expf(s.y), expf(s.z));
        const float inv = 1.f / fmaxf(sqrtf(q.x * q.x + q.y * q.y + q.z * q.z + q.w * q.w), 1e-12f);
        q = make_float4(q.x * inv, q.y * inv, q.z * inv, q.w * inv);
        cov3d_from_scale_rot(s, p.scale_modifier, q, c3);
    }
    const float opac = 1.f / (1.f + expf(-opacities[idx]));
    for (int v = 0; v < vs.V; v++)
        gs_project_one(idx, m, c3, [&]() { return opac; }, sh, (const float*)nullptr, vs.v[v].cam, p.W, p.H, p.gx, p.gy, p.deg, vs.v[v].out);
}
static bool pre_sh_regs() {
    static int v = -1;
    if (v < 0) { const char* e = getenv("C3D_PRE_REGS"); v = e ? atoi(e) != 0 : 1; }
    return v != 0;
}
// geoms[v] / radii[v]: the state buffers of view v; views[v]: its camera (GsParams of that view; N, W, H, scale_modifier, deg must agree)
int gs_launch_preprocess_views(const GsParams* views, int V, const GsGeom* geoms, int* const* radii, const float* means3D, const float* f_dc, const float* f_rest,
                               const float* opacity_raw, const float* scaling_raw, const float* rotation_raw, hipStream_t s) {
    if (V == 0 || views[0].N == 0) return 0;
    if (V > GS_MAX_BWD_VIEWS) { c3d_set_error("gs_launch_preprocess_views: too many views in one launch"); return -1; }
    GsPreViews pv;
    pv.V = V;
    for (int i = 0; i < V; i++) {
        const GsParams& q = views[i];
        pv.v[i] = GsPreView{GsPreCam{q.view, q.proj, q.campos, q.tanfovx, q.tanfovy, q.focal_x, q.focal_y},
                            GsPreOut{geoms[i].rec0, geoms[i].tiles, geoms[i].rect, geoms[i].key[0], geoms[i].clamped, radii[i]}};
    }
    const int T = 128;
    if (pre_sh_regs())
        hipLaunchKernelGGL(k_preprocess_views_r, dim3(c3d_cdiv(views[0].N, T)), dim3(T), 0, s, views[0], pv, means3D, f_dc, f_rest, opacity_raw, scaling_raw, rotation_raw);
    else
        hipLaunchKernelGGL(k_preprocess_views, dim3(c3d_cdiv(views[0].N, T)), dim3(T), T * SH_ROW * sizeof(float), s, views[0], pv, means3D, f_dc, f_rest,
                           opacity_raw, scaling_raw, rotation_raw);
    C3D_LAUNCH_CHECK();
    return 0;
}

// workgroup size of the raw-parameter preprocess (C3D_PRE_THREADS = 64 | 128 (default) | 256): the kernel stages 196 B of SH per lane in LDS, so smaller
// workgroups interleave the load and compute phases of more workgroups per CU at the same wave count
static int gs_pre_threads() {
    static int v = -1;
    if (v < 0) { const char* e = getenv("C3D_PRE_THREADS"); v = e ? atoi(e) : 128; if (v != 64 && v != 128 && v != 256) v = 128; }
    return v;
}
int gs_launch_preprocess_raw(const GsParams& p, const float* means3D, const float* f_dc, const float* f_rest, const float* opacity_raw,
                             const float* scaling_raw, const float* rotation_raw, GsGeom& g, int* radii, hipStream_t s) {
    if (p.N == 0) return 0;
    const int T = gs_pre_threads();
    hipLaunchKernelGGL((k_preprocess<true, true>), dim3(c3d_cdiv(p.N, T)), dim3(T), T * SH_ROW * sizeof(float), s, p, means3D, f_dc, f_rest,
                       (const float*)nullptr, opacity_raw, scaling_raw, rotation_raw, (const float*)nullptr, g, radii);
    C3D_LAUNCH_CHECK();
    return 0;
}

// ------------------------------------------------------------------------------------------
// A3 emit: Gaussians are visited in ascending (depth, id) rank, each writes one (tile, id) pair per
// touched tile.  A stable sort by tile id afterwards leaves every tile's list depth-ordered.
// ------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_emit(GsParams p, const uint32_t* __restrict__ order, const uint32_t* __restrict__ offsets,
                                               const uint4* __restrict__ einfo, uint32_t* __restrict__ tkey, uint32_t* __restrict__ tval, uint32_t cap) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= p.N) return;
    uint32_t off = (r == 0) ? 0u : offsets[r - 1];
    if (offsets[r] == off) return;   // culled, or narrowed to no tiles
    const uint32_t gid = order[r];
    const uint4 ei = einfo[gid];     // ONE 16-byte gather per Gaussian: the tile rect (record-base scan, c3d_scan_u32's epilogue)
    const int x0 = (int)(ei.y & 0xFFFFu), y0 = (int)(ei.y >> 16), x1 = (int)(ei.z & 0xFFFFu), y1 = (int)(ei.z >> 16);
    for (int y = y0; y < y1; y++)
        for (int x = x0; x < x1; x++) {
            if (off < cap) { tkey[off] = (uint32_t)(y * p.gx + x); tval[off] = gid; }
            off++;
        }
}
// Round 3 (gs_bin_local): emission in Gaussian-ID order -- a Gaussian's pairs start at its record base (the ONE scan of the chain), reads are coalesced -- and the depth
// order is established per tile afterwards (c3d_segment_sort_u32); no global depth sort, no second scan.
__global__ void __launch_bounds__(256) k_emit_id(GsParams p, const uint32_t* __restrict__ tiles, const uint4* __restrict__ einfo, uint32_t* __restrict__ tkey,
                                                  uint32_t* __restrict__ tval, uint32_t cap) {
    const int gid = blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= p.N || tiles[gid] == 0u) return;
    const uint4 ei = einfo[gid];
    uint32_t off = ei.w;
    const int x0 = (int)(ei.y & 0xFFFFu), y0 = (int)(ei.y >> 16), x1 = (int)(ei.z & 0xFFFFu), y1 = (int)(ei.z >> 16);
    for (int y = y0; y < y1; y++)
        for (int x = x0; x < x1; x++) {
            if (off < cap) { tkey[off] = (uint32_t)(y * p.gx + x); tval[off] = (uint32_t)gid; }
            off++;
        }
}
int gs_launch_emit(const GsParams& p, const GsGeom& g, int res, const int* radii, GsBinning& b, hipStream_t s, uint32_t cap) {
    (void)radii;
    if (p.N == 0) return 0;
    if (res < 0) hipLaunchKernelGGL(k_emit_id, dim3(c3d_cdiv(p.N, 256)), dim3(256), 0, s, p, g.tiles, g.einfo, b.tkey[0], b.tval[0], cap);      // res < 0: id order
    else hipLaunchKernelGGL(k_emit, dim3(c3d_cdiv(p.N, 256)), dim3(256), 0, s, p, g.order[res], g.offsets, g.einfo, b.tkey[0], b.tval[0], cap);
    C3D_LAUNCH_CHECK();
    return 0;
}

// A5: [start,end) of every tile in the sorted pair list
__global__ void __launch_bounds__(256) k_ranges(const uint32_t* __restrict__ tkey, uint2* __restrict__ ranges, long long D, const uint32_t* __restrict__ d_dev) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (d_dev) D = min((long long)*d_dev, D);
    if (i >= D) return;
    const uint32_t t = tkey[i];
    if (i == 0 || tkey[i - 1] != t) ranges[t].x = (uint32_t)i;
    if (i == D - 1 || tkey[i + 1] != t) ranges[t].y = (uint32_t)(i + 1);
}
// `ranges` must be zero on entry: the binning stage clears it together with the tile-sort state (GsBinning::zero_bytes)
int gs_launch_ranges(const GsBinning& b, int res, long long D, int tiles, hipStream_t s, const uint32_t* d_dev) {
    (void)tiles;
    if (D == 0) return 0;
    hipLaunchKernelGGL(k_ranges, dim3(c3d_cdiv(D, 256)), dim3(256), 0, s, b.tkey[res], b.ranges, D, d_dev);
    C3D_LAUNCH_CHECK();
    return 0;
}

// ------------------------------------------------------------------------------------------
// A6 composite forward.  One 256-lane workgroup per 16x16 tile; wave w owns the 8x8 pixel quadrant
// (w&1, w>>1), one pixel per lane.  Splat records are staged through LDS in rounds of 256 together
// with a 4-bit "quadrants this splat can touch" mask (alpha >= 1/255 box vs quadrant).  Each wave turns
// the masks into a 64-bit ballot per 64 staged splats and walks only its set bits (scalar loop), so a
// wave never evaluates a splat that cannot contribute to its quadrant; the per-splat data is
// wave-uniform in the inner loop (LDS broadcast reads).
// ------------------------------------------------------------------------------------------
#define FWD_ROUND 256   // splats staged per round (128 measures the same within noise)
// RECORD: the walk also notes, per list position, which of the four quadrants blended the splat into at least one pixel (`pact`, one byte
// per (tile, splat) pair, bit w = wave w).  The backward pass walks exactly those (quadrant, splat) pairs: at the BASELINE workload half of
// the pairs that pass the geometric quadrant test are blended nowhere (occluded, or below 1/255 on the pixel grid), and deciding that
// again cost the backward kernel a sixth of its VALU time.  Inference launches use RECORD = false and are unchanged.
template <bool RECORD>
__global__ void __launch_bounds__(256) k_composite_fwd(GsParams p, const uint2* __restrict__ ranges, const uint32_t* __restrict__ point_list,
                                                        const float4* __restrict__ rec0, const float4* __restrict__ rec1,
                                                        const float4* __restrict__ rec2, float* __restrict__ out_color,
                                                        float* __restrict__ out_depth, float* __restrict__ out_alpha,
                                                        float* __restrict__ final_T, uint32_t* __restrict__ n_contrib, uint8_t* __restrict__ pact, size_t pstride, int sh) {
    __shared__ float4 s0[FWD_ROUND];
    __shared__ float4 s1[FWD_ROUND];
    __shared__ float4 s2[FWD_ROUND];
    __shared__ uint32_t smask[FWD_ROUND];
    __shared__ uint32_t sact[RECORD ? FWD_ROUND : 1];   // byte w of sact[j]: wave w blended slot j
    int tx, ty;   // XCD-aware, load-balanced tile order (gs_block_tile).  Speed only, never correctness.
    if (!gs_block_tile(blockIdx.x, p.gx, p.gy, tx, ty, sh)) return;
    const int tile = ty * p.gx + tx;
    const int lane = c3d_lane(), wave = threadIdx.x >> 6;
    const int X0 = tx * C3D_TILE_X, Y0 = ty * C3D_TILE_Y;
    const int pxi = X0 + ((wave & 1) << 3) + (lane & 7), pyi = Y0 + ((wave >> 1) << 3) + (lane >> 3);
    const bool inside = pxi < p.W && pyi < p.H;
    // A finished pixel (saturated, or outside the image) is parked at x = GS_PARKED: every later splat then evaluates to alpha = 0 there and
    // fails the 1/255 test by itself, so the walk needs no per-splat bookkeeping of a `done` lane mask (the kernel is bound by instruction
    // issue, scalar instructions included: this removes 9 of them per evaluation).
    float pxf = inside ? (float)pxi : GS_PARKED;
    const float pyf = (float)pyi;
    const size_t pid = (size_t)pyi * p.W + pxi;     // formed here so that only the float coordinates stay live in the loop
    const uint2 rg = ranges[tile];
    const int todo = (int)(rg.y - rg.x);
    float T = 1.f, C0 = 0.f, C1 = 0.f, C2 = 0.f, Dp = 0.f, A = 0.f;
    uint32_t last = 0;

    for (int base = 0; base < todo; base += FWD_ROUND) {
        if (__syncthreads_count(pxf == GS_PARKED) == 256) break;
        const int n = min(FWD_ROUND, todo - base);
        gs_stage_round(point_list + rg.x + base, n, rec0, s0, s1, s2);
        __syncthreads();
        if ((int)threadIdx.x < n) {
            const float4 a0 = s0[threadIdx.x], a1 = s1[threadIdx.x], a2 = s2[threadIdx.x];
            smask[threadIdx.x] = gs_quadrant_mask(a0, a1, a2, X0, Y0);
            // the conic stays in LDS pre-scaled by -log2(e)/2 (xx, yy) and -log2(e) (xy): the loop then feeds v_exp_f32 directly
            s0[threadIdx.x] = make_float4(a0.x, a0.y, GS_CONIC_HALF * a0.z, GS_CONIC_FULL * a0.w);
            s1[threadIdx.x].x = GS_CONIC_HALF * a1.x;
        }
        if (RECORD) sact[threadIdx.x] = 0u;
        __syncthreads();
        for (int c = 0; c < n; c += 64) {
            const int jj = c + lane;
            uint64_t m = __ballot(jj < n && ((smask[jj] >> wave) & 1u));
            if (__ballot(pxf != GS_PARKED) == 0ull) break;           // every pixel of this quadrant has saturated
            while (m) {
                const int bitpos = (int)__builtin_ctzll(m);
                const int j = c + bitpos;
                m = gs_clear_bit64(m, bitpos);
                const float4 a0 = s0[j], a1 = s1[j];
                const float dx = a0.x - pxf, dy = a0.y - pyf;
                const float power = gs_power(a0, a1.x, dx, dy);      // log2(e) * (-q/2): same sign as the exponent
                const float alpha = fminf(0.99f, a1.y * __builtin_amdgcn_exp2f(power));
                const bool ok = power <= 0.f && alpha >= 1.f / 255.f;
                const float testT = T * (1.f - alpha);
                const bool stop = ok && testT < 0.0001f;
                pxf = stop ? GS_PARKED : pxf;
                if (ok && !stop) {
                    if (RECORD) ((uint8_t*)sact)[4 * j + wave] = 1;   // every blending lane stores the same byte: one LDS pass, no ballot
                    const float4 a2 = s2[j];
                    const float w = alpha * T;
                    C0 += a1.z * w; C1 += a1.w * w; C2 += a2.x * w;
                    Dp += a2.y * w; A += w;
                    T = testT;
                    last = (uint32_t)(base + j + 1);
                }
            }
        }
        if (RECORD) {
            __syncthreads();
            if ((int)threadIdx.x < n) {
                const uint32_t a = sact[threadIdx.x];
#pragma unroll
                for (int w = 0; w < 4; w++) pact[(size_t)w * pstride + rg.x + base + threadIdx.x] = (uint8_t)((a >> (8 * w)) & 1u);   // one byte plane per quadrant (gs_pair_activity)
            }
        }
    }
    if (inside) {
        const size_t P = (size_t)p.W * p.H;
        final_T[pid] = T;
        n_contrib[pid] = last;
        out_color[pid] = C0 + T * p.bg[0];
        out_color[P + pid] = C1 + T * p.bg[1];
        out_color[2 * P + pid] = C2 + T * p.bg[2];
        out_depth[pid] = Dp;
        out_alpha[pid] = A;
    }
}

// ------------------------------------------------------------------------------------------
// A6, wave-autonomous form (round 3, the default): ONE WAVE per 8x8 pixel quadrant of a 16x16 tile -- a 64-lane workgroup, no workgroup-level
// staging, no barrier, nothing shared with the other three quadrants of the tile but the tile's sorted splat list.
// Why: profiles/r02z_sq_instruction_mix_lanes1.csv -- the 256-lane kernel above issues 26 VALU + 16-20 scalar + 4 LDS instructions per walked
// (quadrant, splat) pair: a third of its issue slots are the scalar walk over a ballot (find-first-set, clear, address, exec save / restore around
// the blend) and every round costs three workgroup barriers.  Here a wave
//   * takes 64 list entries at a time, one per lane (id + the 48-B record straight into registers; the NEXT chunk's loads are issued before the
//     current chunk is walked, so the gathers' latency hides under the walk),
//   * tests its own quadrant only (gs_rect_hit: exact ellipse-vs-rectangle, the same test as gs_quadrant_mask), and COMPACTS the hits -- records,
//     conic pre-scaled for v_exp_f32 -- into a wave-private LDS list (ballot + mbcnt), padded to a multiple of four with zero-opacity dummies,
//   * walks that list with a counted, 4x unrolled loop of LDS broadcast reads at immediate offsets: no ballot walk, no per-splat scalar address
//     arithmetic, and the blend is branch-free: which pixels still take splats (`alive`), which pass the alpha test, which saturate at this splat
//     are lane masks in SGPR pairs, combined by scalar instructions and fed to v_cndmask_b32_e64.
// What bounds it (profiles/r03*): the VALU pipe at the measured issue cost of its instructions on gfx950 (profiles/r01f_valu_rate_microbench.txt:
// 2.9 cycles for add / mul / fma, 4.5 for compares and min, 4.7 for an SGPR-mask select, 8.3 for v_exp_f32) -- about 80 cycles per walked
// (quadrant, splat) pair of which 44 % of the lanes blend.  Fewer instructions per pair is the only lever left in this decomposition.
// RECORD: one scalar bit per walked list entry ("some lane blended it": s_cmp_lg_u64 + two s_addc_u32 shift it into a 64-bit mask), written out as
// the quadrant's byte plane of the pair-activity record (gs_pair_activity) -- one coalesced byte store per 64 list entries.
// Per-pixel arithmetic as k_composite_fwd (same images, n_contrib and gradients: tests/test_gs_hip.py::test_forward_kernels_agree) except the alpha
// output, which this kernel takes from the telescoped sum 1 - T_final instead of a sixth accumulator (equal to rounding).
// ------------------------------------------------------------------------------------------
#define FWQ_PAD 4
#define FWQ_SLOTS (64 + FWQ_PAD)
template <bool RECORD>
__global__ void __launch_bounds__(64, 8) k_composite_fwd_w(GsParams p, const uint2* __restrict__ ranges, const uint32_t* __restrict__ point_list,
                                                            const float4* __restrict__ rec, float* __restrict__ out_color, float* __restrict__ out_depth,
                                                            float* __restrict__ out_alpha, float* __restrict__ final_T, uint32_t* __restrict__ n_contrib,
                                                            uint8_t* __restrict__ pact, size_t pstride, int sh) {
    // the wave's compacted splat list, three 16-byte parts per entry at ONE running offset:
    //   part 0 (px, py, -log2e/2 A, -log2e B)   part 1 (-log2e/2 C, opacity, r, g)   part 2 (b, view depth, list position + 1 as int bits, -)
    __shared__ float4 cl[3][FWQ_SLOTS];
    const int b = blockIdx.x, q = (b >> 3) & 3;   // the four quadrants of a tile sit on ONE XCD (b & 7): they gather the same records
    int tx, ty;
    if (!gs_block_tile((b & 7) | ((b >> 5) << 3), p.gx, p.gy, tx, ty, sh)) return;
    const int tile = ty * p.gx + tx, lane = (int)threadIdx.x;
    const int QX = tx * C3D_TILE_X + ((q & 1) << 3), QY = ty * C3D_TILE_Y + ((q >> 1) << 3);
    const int pxi = QX + (lane & 7), pyi = QY + (lane >> 3);
    const bool inside = pxi < p.W && pyi < p.H;
    const float pxf = (float)pxi, pyf = (float)pyi;
    const size_t pid = (size_t)pyi * p.W + pxi;
    const float rx0 = (float)QX, ry0 = (float)QY;
    const uint2 rg = ranges[tile];
    const int todo = (int)(rg.y - rg.x);
    float T = 1.f, C0 = 0.f, C1 = 0.f, C2 = 0.f, Dp = 0.f;
    int last = 0;
    // `alive`: the pixels that still take splats, as a lane mask in an SGPR pair.  A pixel dies when a splat would push its transmittance below 1e-4
    // (it does not take that splat) or when it lies outside the image; dead lanes run the same instructions with every select closed.
    uint64_t alive = __ballot(inside);
    const float k255 = 1.f / 255.f, kT = 0.0001f;

    float4 n0 = make_float4(0, 0, 0, 0), n1 = n0, n2 = n0;
    auto fetch = [&](int base) {
        const int e = base + lane;
        if (e < todo) {
            const size_t id = point_list[rg.x + e];
            n0 = rec[4 * id]; n1 = rec[4 * id + 1]; n2 = rec[4 * id + 2];
        }
    };
    if (alive && todo > 0) fetch(0);
    for (int base = 0; base < todo && alive; base += 64) {
        const float4 a0 = n0, a1 = n1, a2 = n2;
        const bool have = base + lane < todo;
        if (base + 64 < todo) fetch(base + 64);                       // in flight while this chunk is walked
        const bool hit = have && gs_rect_hit(a0, a1, a2, rx0, ry0);
        const uint64_t m = __ballot(hit);
        const int n = __popcll(m);
        const int pos = (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
        uint32_t act_lo = 0, act_hi = 0;
        int walked = 0;
        if (n) {
            __syncthreads();                                           // one wave: orders this chunk's LDS writes behind the last chunk's reads
            if (hit) {
                cl[0][pos] = make_float4(a0.x, a0.y, GS_CONIC_HALF * a0.z, GS_CONIC_FULL * a0.w);
                cl[1][pos] = make_float4(GS_CONIC_HALF * a1.x, a1.y, a1.z, a1.w);
                cl[2][pos] = make_float4(a2.x, a2.y, __int_as_float(base + lane + 1), 0.f);
            }
            const int npad = (n + 3) & ~3;
            if (lane < npad - n) {                                     // zero-opacity dummies: alpha = 0 fails the 1/255 test on every pixel
                cl[0][n + lane] = make_float4(0.f, 0.f, 0.f, 0.f); cl[1][n + lane] = make_float4(0.f, 0.f, 0.f, 0.f); cl[2][n + lane] = make_float4(0.f, 0.f, 0.f, 0.f);
            }
            __syncthreads();
            // running byte offset into cl[0], kept in a VGPR on purpose (one v_add per group of four; the three parts and the four entries of a group are
            // immediate offsets of the ds_read): with a scalar index hipcc re-materialises the address with v_mov for every read
            uint32_t vo;
            asm volatile("v_mov_b32 %0, 0" : "=v"(vo));
            const char* lbase = reinterpret_cast<const char*>(&cl[0][0]);
            int i = 0;
            for (; i < npad && alive; i += 4, vo += 64) {
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    const float4 s0 = *reinterpret_cast<const float4*>(lbase + vo + 16 * u);
                    const float4 s1 = *reinterpret_cast<const float4*>(lbase + vo + 16 * u + sizeof(float4) * FWQ_SLOTS);
                    const float4 s2 = *reinterpret_cast<const float4*>(lbase + vo + 16 * u + 2 * sizeof(float4) * FWQ_SLOTS);
                    const float dx = s0.x - pxf, dy = s0.y - pyf;
                    const float power = gs_power(s0, s1.x, dx, dy);          // log2(e) * (-q/2)
                    const float alpha = fminf(0.99f, s1.y * __builtin_amdgcn_exp2f(power));
                    uint64_t ok, k1, st;
                    asm("v_cmp_ge_f32_e64 %0, 0, %1" : "=s"(ok) : "v"(power));
                    asm("v_cmp_le_f32_e64 %0, %1, %2" : "=s"(k1) : "s"(k255), "v"(alpha));
                    ok &= k1 & alive;
                    const float testT = T * (1.f - alpha);
                    asm volatile("" ::"v"(s2.w));                            // keeps part 2 a 16-byte read: ds_read_b96 takes twice the LDS cycles of ds_read_b128 on gfx950
                    asm("v_cmp_gt_f32_e64 %0, %1, %2" : "=s"(st) : "s"(kT), "v"(testT));
                    st &= ok;                                                // these pixels saturate AT this splat: they do not take it, and die
                    alive &= ~st;
                    ok &= ~st;                                               // = the lanes that blend it
                    const float w = sel64z(ok, alpha * T);
                    T = sel64(ok, testT, T);
                    C0 += s1.z * w; C1 += s1.w * w; C2 += s2.x * w;
                    Dp += s2.y * w;
                    last = sel64i(ok, __float_as_int(s2.z), last);
                    if (RECORD)                                              // act = (act << 1) | (some lane blended this entry): SCC rides the carry chain
                        asm volatile("s_cmp_lg_u64 %2, 0\n\ts_addc_u32 %0, %0, %0\n\ts_addc_u32 %1, %1, %1" : "+s"(act_lo), "+s"(act_hi) : "s"(ok) : "scc");
                }
            }
            walked = i;
        }
        if (RECORD && have) {
            // entry k of the compact list sits at bit (walked - 1 - k) of act; entries the walk never reached blended nothing
            const uint64_t act = ((uint64_t)act_hi << 32) | act_lo;
            const uint32_t bit = (hit && pos < walked) ? (uint32_t)((act >> (walked - 1 - pos)) & 1ull) : 0u;
            pact[(size_t)q * pstride + rg.x + base + lane] = (uint8_t)bit;
        }
    }
    if (inside) {
        const size_t P = (size_t)p.W * p.H;
        final_T[pid] = T;
        n_contrib[pid] = (uint32_t)last;
        out_color[pid] = C0 + T * p.bg[0];
        out_color[P + pid] = C1 + T * p.bg[1];
        out_color[2 * P + pid] = C2 + T * p.bg[2];
        out_depth[pid] = Dp;
        out_alpha[pid] = 1.f - T;                 // sum of the blend weights alpha_i T_i = T_0 - T_final, telescoped (the weights are T_i - T_{i+1})
    }
}

// which forward compositing kernel: C3D_FWD_KERNEL = 1 (default) wave per quadrant (k_composite_fwd_w) | 0 workgroup per tile (k_composite_fwd)
static int gs_fwd_kernel() {
    static int v = -1;
    if (v < 0) { const char* e = getenv("C3D_FWD_KERNEL"); v = e ? atoi(e) : 1; if (v != 0 && v != 1) v = 1; }
    return v;
}
int gs_launch_composite_fwd(const GsParams& p, const GsGeom& g, const GsBinning& b, int res, GsImage& im,
                            float* out_color, float* out_depth, float* out_alpha, bool record_activity, hipStream_t s) {
    const int tiles = p.gx * p.gy;
    if (tiles == 0) return 0;
    uint8_t* pact = record_activity ? gs_pair_activity(b, res) : nullptr;
    if (gs_fwd_kernel() == 1) {
        const dim3 grid(4 * gs_block_count(p.gx, p.gy, gs_supertile_shift()));
        if (record_activity)
            hipLaunchKernelGGL(k_composite_fwd_w<true>, grid, dim3(64), 0, s, p, b.ranges, b.tval[res], g.rec0, out_color, out_depth, out_alpha, im.final_T, im.n_contrib,
                               pact, b.pair_stride, gs_supertile_shift());
        else
            hipLaunchKernelGGL(k_composite_fwd_w<false>, grid, dim3(64), 0, s, p, b.ranges, b.tval[res], g.rec0, out_color, out_depth, out_alpha, im.final_T, im.n_contrib,
                               pact, b.pair_stride, gs_supertile_shift());
    } else if (record_activity)
        hipLaunchKernelGGL(k_composite_fwd<true>, dim3(gs_block_count(p.gx, p.gy, gs_supertile_shift())), dim3(256), gs_lds_pad(false), s, p, b.ranges, b.tval[res], g.rec0, g.rec1, g.rec2,
                           out_color, out_depth, out_alpha, im.final_T, im.n_contrib, pact, b.pair_stride, gs_supertile_shift());
    else
        hipLaunchKernelGGL(k_composite_fwd<false>, dim3(gs_block_count(p.gx, p.gy, gs_supertile_shift())), dim3(256), gs_lds_pad(false), s, p, b.ranges, b.tval[res], g.rec0, g.rec1, g.rec2,
                           out_color, out_depth, out_alpha, im.final_T, im.n_contrib, pact, b.pair_stride, gs_supertile_shift());
    C3D_LAUNCH_CHECK();
    return 0;
}

// ------------------------------------------------------------------------------------------
// Pixel loss of the fused training step: L = scale * [ w_l1 mean|c - t| + w_l2 mean (c - t)^2 + w_a mean (alpha - ta)^2 ], with the rendered
// colour clamped to [0,1] first (GaussianSplattingRenderer.render returns image.clamp(0,1)).  Writes dL/dcolor, dL/dalpha and adds the
// loss value to *loss_out.
// ------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_loss_grad(const float* __restrict__ color, const float* __restrict__ alpha, const float* __restrict__ tcolor,
                                                    const float* __restrict__ talpha, const float* __restrict__ cmask, long long P, float w_l1, float w_l2, float w_a, float scale,
                                                    float* __restrict__ dcolor, float* __restrict__ dalpha, float* __restrict__ loss_out) {
    __shared__ float red[4];
    float l = 0.f;
    const float inv3p = 1.f / (3.f * (float)P), invp = 1.f / (float)P;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < P; i += (long long)gridDim.x * blockDim.x) {
#pragma unroll
        for (int ch = 0; ch < 3; ch++) {
            const float c = color[ch * P + i];
            const float cc = fminf(fmaxf(c, 0.f), 1.f);
            const float mk = cmask ? cmask[i] : 1.f;              // optional per-pixel weight: loss on (image * mask) vs (target * mask)
            const float d = (cc - tcolor[ch * P + i]) * mk;
            l += (w_l1 * fabsf(d) + w_l2 * d * d) * inv3p;
            const float pass = (c >= 0.f && c <= 1.f) ? 1.f : 0.f;
            const float sg = d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f);
            dcolor[ch * P + i] = scale * pass * mk * (w_l1 * sg + 2.f * w_l2 * d) * inv3p;
        }
        float da = 0.f;
        if (talpha) { const float d = alpha[i] - talpha[i]; l += w_a * d * d * invp; da = scale * 2.f * w_a * d * invp; }
        dalpha[i] = da;
    }
    l = c3d_wave_sum(l * scale);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = l;
    __syncthreads();
    if (threadIdx.x == 0 && loss_out) atomicAdd(loss_out, red[0] + red[1] + red[2] + red[3]);   // one atomic per workgroup, <= 1024 workgroups
}
int gs_launch_loss_grad(const float* color, const float* alpha, const float* tcolor, const float* talpha, const float* cmask, long long P, float w_l1, float w_l2, float w_a,
                        float scale, float* dcolor, float* dalpha, float* loss_out, hipStream_t s) {
    if (P == 0) return 0;
    hipLaunchKernelGGL(k_loss_grad, dim3(min(c3d_cdiv(P, 256), 1024)), dim3(256), 0, s, color, alpha, tcolor, talpha, cmask, P, w_l1, w_l2, w_a, scale, dcolor, dalpha, loss_out);
    C3D_LAUNCH_CHECK();
    return 0;
}

// A9 mark_visible (exported by the dependency; not called by the reference's MVs path)
__global__ void __launch_bounds__(256) k_mark_visible(int N, const float* __restrict__ means3D, const float* __restrict__ view, uint8_t* __restrict__ present) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= N) return;
    const Mat16 V = load_mat16(view);
    const float3 pv = xform4x3(make_float3(means3D[3 * idx], means3D[3 * idx + 1], means3D[3 * idx + 2]), V);
    present[idx] = pv.z > 0.2f;
}
int gs_launch_mark_visible(int N, const float* means3D, const float* view, const float* proj, uint8_t* present, hipStream_t s) {
    (void)proj;
    if (N == 0) return 0;
    hipLaunchKernelGGL(k_mark_visible, dim3(c3d_cdiv(N, 256)), dim3(256), 0, s, N, means3D, view, present);
    C3D_LAUNCH_CHECK();
    return 0;
}
