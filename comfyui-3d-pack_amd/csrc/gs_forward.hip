// gs_forward.hip -- 3DGS forward: preprocess (A1), tile binning (A2-A5) and per-tile compositing (A6).
// Stage names follow SURVEY.md section 2.3-A; the arithmetic contract is Appendix A of that file.
// Boundary this implements: the rasterizer call at
// MVs_Algorithms/GaussianSplatting/main_3DGS_renderer.py:927-936 (reference).
#include "gs_internal.h"
#include "gs_math.h"
#include "scan_wave.h"

// ------------------------------------------------------------------------------------------
// A1, the part that depends on the view: projection, EWA covariance, radius, tile rect, SH colour of ONE Gaussian (mean m, 3D covariance c3,
// opacity as given by `opacity()`, SH row `sh` with stride 3 per coefficient) -> the projected state arrays of that view.  Shared by the
// per-view kernel and the all-views-of-a-step kernel, so the two cannot drift apart.
// ------------------------------------------------------------------------------------------
struct GsPreCam { const float* view; const float* proj; const float* campos; float tanfovx, tanfovy, focal_x, focal_y; };
struct GsPreOut { float4* rec0; uint32_t* tiles; uint2* rect; uint32_t* key0; uint8_t* clamped; int* radii; };
template <class OpacityFn>
__device__ __forceinline__ void gs_project_one(int idx, const float3 m, const float c3[6], OpacityFn opacity, const float* __restrict__ sh, const float* __restrict__ colors_precomp,
                                               const GsPreCam& cam, int W, int H, int gx, int gy, int deg, const GsPreOut& o, int rect4) {
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
            const float3 cpos = load_vec3_const(cam.campos);
            float dx = m.x - cpos.x, dy = m.y - cpos.y, dz = m.z - cpos.z;
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
        // (the 4th float4 of the 64-B record line stays unwritten: filling it to make the write a whole line was measured in round 4 -- projection 0.277 / 0.298 ms
        //  per 8 views without, 0.289 / 0.296 with, same box, alternating: no difference, the L2 merges the three 16-B stores and HBM takes the 48 B as sectors)
        o.clamped[idx] = cl;
        rad = r;
        nt = (uint32_t)((x1 - x0) * (y1 - y0));
        if (rect4) reinterpret_cast<uint32_t*>(o.rect)[idx] = c3d_rect_pack((uint32_t)x0, (uint32_t)y0, (uint32_t)x1, (uint32_t)y1);
        else o.rect[idx] = make_uint2((uint32_t)x0 | ((uint32_t)y0 << 16), (uint32_t)x1 | ((uint32_t)y1 << 16));
        key = nt ? __float_as_uint(pv.z) : 0xFFFFFFFFu;
    } while (false);
    if (rad == 0) { if (rect4) reinterpret_cast<uint32_t*>(o.rect)[idx] = 0u; else o.rect[idx] = make_uint2(0u, 0u); }      // culled: an empty rect (the emit-offset scan takes the tile count from the rect)
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
// (The raw-parameter form -- exp / sigmoid / normalize folded in, split f_dc / f_rest storage -- is k_preprocess_views_r below, also for one view.)
template <bool STAGED>
__global__ void __launch_bounds__(256) k_preprocess(GsParams p, const float* __restrict__ means3D, const float* __restrict__ shs,
                                                     const float* __restrict__ colors_precomp, const float* __restrict__ opacities,
                                                     const float* __restrict__ scales, const float* __restrict__ rotations,
                                                     const float* __restrict__ cov3D_precomp, GsGeom g, int* __restrict__ radii) {
    extern __shared__ float sh_lds[];
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (STAGED) {
        const size_t g0 = (size_t)blockIdx.x * blockDim.x;
        sh_stage_in(shs, g0, min((int)blockDim.x, p.N - (int)g0), sh_lds);
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
        cov3d_from_scale_rot(s, p.scale_modifier, q, c3);
    }
    const float* sh = STAGED ? sh_lds + threadIdx.x * SH_ROW : shs + (size_t)idx * p.M * 3;
    gs_project_one(idx, m, c3, [&]() { return opacities[idx]; }, sh, colors_precomp,
                   GsPreCam{p.view, p.proj, p.campos, p.tanfovx, p.tanfovy, p.focal_x, p.focal_y}, p.W, p.H, p.gx, p.gy, p.deg,
                   GsPreOut{g.rec0, g.tiles, g.rect, g.key[0], g.clamped, radii}, p.rect4);
}

int gs_launch_preprocess(const GsParams& p, const float* means3D, const float* shs, const float* colors_precomp,
                         const float* opacities, const float* scales, const float* rotations, const float* cov3D_precomp,
                         GsGeom& g, int* radii, hipStream_t s) {
    if (p.N == 0) return 0;
    const bool staged = shs && !colors_precomp && p.M == 16 && ((uintptr_t)shs % 16 == 0);
    if (staged)
        hipLaunchKernelGGL((k_preprocess<true>), dim3(c3d_cdiv(p.N, 256)), dim3(256), 256 * SH_ROW * sizeof(float), s, p, means3D, shs,
                           colors_precomp, opacities, scales, rotations, cov3D_precomp, g, radii);
    else
        hipLaunchKernelGGL((k_preprocess<false>), dim3(c3d_cdiv(p.N, 256)), dim3(256), 0, s, p, means3D, shs,
                           colors_precomp, opacities, scales, rotations, cov3D_precomp, g, radii);
    C3D_LAUNCH_CHECK();
    return 0;
}
// ---- A1 for all views of a step at once ---------------------------------------------------------------------------------------------------
// The per-view kernel streams 44 + 192 B of parameters per Gaussian for every view; over the 8 views of a training step that is 1.9 GB of the
// same 236 MB.  Here a lane keeps its Gaussian (mean, 3D covariance, opacity and the 48 SH coefficients in registers) and walks the views:
// parameters are read once per launch, what is left per view is the 93 B of projected state it writes.  Same arithmetic, statement by statement,
// as the plain-boundary kernel k_preprocess (both call gs_project_one; the tests hold the two together).
struct GsPreView { GsPreCam cam; GsPreOut out; };
struct GsPreViews { int V; GsPreView v[GS_MAX_BWD_VIEWS]; };
// LDS is only the transposition buffer of the coalesced SH load (44 rows at a time: 8.6 KB per workgroup); the coefficients move on into registers
// (28 + 48 VGPRs -> 5-6 waves per SIMD).  Round 3's first version kept a 196 B per lane LDS image instead: 3 waves per SIMD, 4 % slower (profiles/r03/).
#define PRE_ROWS 44      // multiple of 4: the 16-byte loads of sh_stage_in_split stay aligned
// degree of the storage with K coefficients per channel (REST3 = 3 (K - 1)): the compiler then drops the SH bands a model of that storage cannot have
__host__ __device__ constexpr int sh_storage_degree(int rest3) { return rest3 >= 45 ? 3 : (rest3 >= 24 ? 2 : (rest3 >= 9 ? 1 : 0)); }
template <int REST3>
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
            sh_stage_in_split<REST3>(f_dc, f_rest, g0 + r0, rows, sh_lds);
            __syncthreads();
            const int mine = (int)threadIdx.x - r0;
            if (mine >= 0 && mine < rows) {
#pragma unroll
                for (int k = 0; k < 3 + REST3; k++) sh[k] = sh_lds[mine * SH_ROW + k];
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
    const int deg = min(p.deg, sh_storage_degree(REST3));
    for (int v = 0; v < vs.V; v++)
        gs_project_one(idx, m, c3, [&]() { return opac; }, sh, (const float*)nullptr, vs.v[v].cam, p.W, p.H, p.gx, p.gy, deg, vs.v[v].out, p.rect4);
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
#define GS_PRE_VIEWS(R3_) hipLaunchKernelGGL(k_preprocess_views_r<R3_>, dim3(c3d_cdiv(views[0].N, T)), dim3(T), 0, s, views[0], pv, means3D, f_dc, f_rest, opacity_raw, scaling_raw, rotation_raw)
    GS_BY_SH_COEFFS(views[0].M, GS_PRE_VIEWS);
#undef GS_PRE_VIEWS
    C3D_LAUNCH_CHECK();
    return 0;
}

// raw parameters, one view: the multi-view kernel with V = 1 -- the per-view drop-in path and the fused multi-view paths then run the SAME instructions per
// Gaussian (bit-identical projected state for every SH storage degree, by construction)
int gs_launch_preprocess_raw(const GsParams& p, const float* means3D, const float* f_dc, const float* f_rest, const float* opacity_raw,
                             const float* scaling_raw, const float* rotation_raw, GsGeom& g, int* radii, hipStream_t s) {
    if (p.N == 0) return 0;
    int* rd[1] = {radii};
    return gs_launch_preprocess_views(&p, 1, &g, rd, means3D, f_dc, f_rest, opacity_raw, scaling_raw, rotation_raw, s);
}

// ------------------------------------------------------------------------------------------
// A3 emit: Gaussians are visited in ascending (depth, id) rank, each contributes one (tile, id) pair per touched tile, row-major over its tile rect.
// A stable sort by tile id afterwards leaves every tile's list depth-ordered.  blockIdx.y = view of a multi-view launch.
// Load-balanced expansion (round 4): a workgroup owns 256 consecutive ranks = one contiguous stretch [B0, B1) of the pair list.  Their offsets, ids and
// rects (all rank-ordered: coalesced loads, the one gather of the chain happened in the emit-offset scan) go to LDS, and the 256 lanes then walk the
// OUTPUT positions -- lane t writes pairs B0 + t, B0 + t + 256, ... after an 8-step search for the Gaussian that owns each -- so every store instruction
// writes 64 consecutive words and a Gaussian with 500 tiles costs its workgroup two extra iterations instead of parking one lane of a wave for 500.
// ------------------------------------------------------------------------------------------
#define EMIT_RANKS 256
#define EMIT_SLOTS 2048      // workgroups resident at once: 256 CUs x 8 (8 KB of LDS, 256 lanes)
// Round 5: the kernel also COUNTS what it writes -- the digit histograms of the tile sort (c3d_common.h: hist_done), taken while the key is in a register, instead of a
// histogram kernel that reads the 32 M keys of an 8-view step back.  For that the workgroups STAY (one per residency slot, chunk after chunk at a stride): the counters live in
// LDS for the workgroup's whole life and go out once -- 2048 x <= 512 global atomics per launch; one workgroup per chunk (31 K of them) would flush 9 M.
// Home view of a workgroup = (linear block id) % V as in the sort passes.  Pairs beyond the capacity are neither written nor counted.
__global__ void __launch_bounds__(EMIT_RANKS) k_emit(int N, int gx, const uint32_t* __restrict__ order, const uint32_t* __restrict__ offsets,
                                                      const uint2* __restrict__ rsort, uint32_t* __restrict__ tkey, uint32_t* __restrict__ tval, uint32_t cap,
                                                      uint32_t* __restrict__ ghist, int passes, size_t vs) {
    __shared__ uint32_t s_end[EMIT_RANKS];      // inclusive scan: end of rank r's pairs
    __shared__ uint32_t s_gid[EMIT_RANKS];
    __shared__ uint2 s_rect[EMIT_RANKS];
    __shared__ uint32_t s_hist[C3D_SORT_MAX_PASSES][256];
    const uint32_t lin = blockIdx.x + gridDim.x * blockIdx.y, view = lin % gridDim.y, wg = lin / gridDim.y;      // wg: 0 .. gridDim.x - 1 within the view
    const size_t voff = (size_t)view * vs;
    order = (const uint32_t*)((const char*)order + voff); offsets = (const uint32_t*)((const char*)offsets + voff); rsort = (const uint2*)((const char*)rsort + voff);
    tkey = (uint32_t*)((char*)tkey + voff); tval = (uint32_t*)((char*)tval + voff); ghist = (uint32_t*)((char*)ghist + voff);
    const int t = threadIdx.x;
    for (int p = 0; p < passes; p++) s_hist[p][t] = 0;
    // The NEXT chunk's inputs are requested before this chunk's pairs are written (round 6): behind the walk they were loads that wait -- vmcnt counts loads and stores alike on
    // this ISA -- for every store of the walk before the chunk can even start.
    const int stride = (int)gridDim.x * EMIT_RANKS;
    uint32_t n_end = 0, n_gid = 0, n_B0 = 0; uint2 n_rect = make_uint2(0u, 0u);
    auto fetch = [&](int r0) {
        if (r0 >= N) return;      // (uniform)
        const int r = min(r0 + t, N - 1);       // ranks past the end repeat the last one's END: empty stretches
        n_end = offsets[r];
        if (r0 + t < N) { n_gid = order[r]; n_rect = rsort[r]; }
        n_B0 = r0 ? offsets[r0 - 1] : 0u;
    };
    fetch((int)wg * EMIT_RANKS);
    for (int r0 = (int)wg * EMIT_RANKS; r0 < N; r0 += stride) {
        __syncthreads();                         // the previous chunk's walk has read s_end / s_gid / s_rect (first chunk: the counters are zero)
        s_end[t] = n_end;
        if (r0 + t < N) { s_gid[t] = n_gid; s_rect[t] = n_rect; }
        const uint32_t B0 = n_B0;
#ifndef C3D_EMIT_NO_PREFETCH      // (A/B switch of profiles/r06/r06ad_*)
        fetch(r0 + stride);
#endif
        __syncthreads();
        const uint32_t B1 = min(s_end[EMIT_RANKS - 1], cap);
        for (uint32_t o = B0 + (uint32_t)t; o < B1; o += EMIT_RANKS) {
            int j = 0;                           // first rank whose end lies beyond o
#pragma unroll
            for (int step = EMIT_RANKS / 2; step >= 1; step >>= 1) j += (s_end[j + step - 1] <= o) ? step : 0;
            const uint32_t start = j ? s_end[j - 1] : B0;
            const uint2 rc = s_rect[j];
            const uint32_t x0 = rc.x & 0xFFFFu, y0 = rc.x >> 16, w = (rc.y & 0xFFFFu) - x0, i = o - start;
            uint32_t q = (uint32_t)__fdividef((float)i, (float)w);      // i < 2^24 (a rect holds at most gx * gy tiles): exact to +-1
            if (q * w > i) q--; else if ((q + 1u) * w <= i) q++;
            const uint32_t key = (y0 + q) * (uint32_t)gx + x0 + (i - q * w);
            tkey[o] = key;
            tval[o] = s_gid[j];
            for (int p = 0; p < passes; p++) atomicAdd(&s_hist[p][(key >> (8 * p)) & 255u], 1u);
        }
#ifdef C3D_EMIT_NO_PREFETCH
        fetch(r0 + stride);
#endif
    }
    __syncthreads();
    uint32_t* mine = ghist + (size_t)(wg % C3D_SORT_HIST_SPLIT) * C3D_SORT_MAX_PASSES * 256;
    for (int p = 0; p < passes; p++) {
        const uint32_t c = s_hist[p][t];
        if (c) atomicAdd(&mine[p * 256 + t], c);
    }
}
// ghist: the head of the tile sort's state block (zero on entry: the binning stage clears it before this launch); passes = sort passes of the tile id
int gs_launch_emit(const GsParams& p, const GsGeom& g, int res, GsBinning& b, hipStream_t s, uint32_t cap, int V, size_t vs, int passes) {
    if (p.N == 0 || V <= 0) return 0;
    if (passes < 1 || passes > C3D_SORT_MAX_PASSES) { c3d_set_error("internal: %d sort passes for the tile id", passes); return -2; }
    const int chunks = c3d_cdiv(p.N, EMIT_RANKS);
    const int nbx = (long long)chunks * V <= EMIT_SLOTS ? chunks : (EMIT_SLOTS / V > 0 ? EMIT_SLOTS / V : 1);
    hipLaunchKernelGGL(k_emit, dim3(nbx, V), dim3(EMIT_RANKS), 0, s, p.N, p.gx, g.order[res], g.offsets, g.rsort, b.tkey[0], b.tval[0], cap, (uint32_t*)b.tmp, passes, vs);
    C3D_LAUNCH_CHECK();
    return 0;
}

// A5, the [start, end) of every tile in the sorted pair list, is no kernel of its own any more (round 6): the last pass of the tile sort leaves the ranges behind
// (k_onesweep RANGES, csrc/scan_sort.hip); readers decode them with gs_tile_range.

// ------------------------------------------------------------------------------------------
// A6 composite forward: ONE WAVE per 8x8 pixel quadrant of a 16x16 tile -- a 64-lane workgroup, no workgroup-level staging, no barrier, nothing
// shared with the other three quadrants of the tile but the tile's sorted splat list.  blockIdx.y = view of a multi-view launch.
// Why (profiles/r02z_sq_instruction_mix_lanes1.csv): round 2's 256-lane kernel (one workgroup per tile, splat rounds staged through LDS) issued 26 VALU +
// 16-20 scalar + 4 LDS instructions per walked (quadrant, splat) pair: a third of its issue slots were the scalar walk over a ballot (find-first-set,
// clear, address, exec save / restore around the blend) and every round cost three workgroup barriers.  Here a wave
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
// The alpha output is the telescoped sum 1 - T_final of the blend weights alpha_i T_i = T_i - T_{i+1} instead of a sixth accumulator (equal to rounding;
// weights below ~6e-8 flush to 0 either way; tests/test_gs_hip.py holds it to the float64 restatement's accumulated sum).
// ------------------------------------------------------------------------------------------
#define FWQ_PAD 4
#define FWQ_SLOTS (64 + FWQ_PAD)
// DEPTH: the depth output is wanted.  The fused training step's loss reads image and alpha only (main_3DGS.py:184-192): its instance drops the accumulator and the plane.
template <bool RECORD, bool DEPTH>
__global__ void __launch_bounds__(64, 8) k_composite_fwd_w(GsParams p, const uint2* __restrict__ ranges, const uint32_t* __restrict__ point_list,
                                                            const float4* __restrict__ rec, GsFwdViews vp, float* __restrict__ final_T, uint32_t* __restrict__ n_contrib,
                                                            uint8_t* __restrict__ pact, size_t pstride, size_t vs, ScanWaveJob sj) {
    // RECORD launches carry the record-base scan of the backward pass in their first sj.blocks workgroups (scan_wave.h).  Measured (profiles/r05f_scan_in_composite_ab.txt,
    // same box, three alternations): the launch grows by 0.09 ms -- the scan's 256 MB per 8 views are not free underneath the compositing waves' record gathers -- where
    // the scan's own launch took 0.125: step 5.134 -> 5.075 ms.  One launch less on the chain; the gain is the tail and the dependent-launch gap, not the traffic.
    if (RECORD && (int)blockIdx.x < sj.blocks) { scan_wave_tile(sj, vs); return; }
    ranges = c3d_view_ptr(ranges, vs); point_list = c3d_view_ptr(point_list, vs); rec = c3d_view_ptr(rec, vs); final_T = c3d_view_ptr(final_T, vs);
    n_contrib = c3d_view_ptr(n_contrib, vs); pact = c3d_view_ptr(pact, vs);
    const float* __restrict__ bg = vp.bg[blockIdx.y];
    float* __restrict__ out_color = vp.color[blockIdx.y];
    float* __restrict__ out_depth = vp.depth[blockIdx.y];
    float* __restrict__ out_alpha = vp.alpha[blockIdx.y];
    // the wave's compacted splat list, three 16-byte parts per entry at ONE running offset:
    //   part 0 (px, py, -log2e/2 A, -log2e B)   part 1 (-log2e/2 C, opacity, r, g)   part 2 (b, view depth, list position + 1 as int bits, -)
    __shared__ float4 cl[3][FWQ_SLOTS];
    const int b = (int)blockIdx.x - (RECORD ? sj.blocks : 0), q = (b >> 3) & 3;   // the four quadrants of a tile sit on ONE XCD (b & 7; sj.blocks is a multiple of 32): they gather the same records
    int tx, ty;
    if (!gs_block_tile((b & 7) | ((b >> 5) << 3), p.gx, p.gy, tx, ty)) return;
    const int tile = ty * p.gx + tx, lane = (int)threadIdx.x;
    const int QX = tx * C3D_TILE_X + ((q & 1) << 3), QY = ty * C3D_TILE_Y + ((q >> 1) << 3);
    const int pxi = QX + (lane & 7), pyi = QY + (lane >> 3);
    const bool inside = pxi < p.W && pyi < p.H;
    const float pxf = (float)pxi, pyf = (float)pyi;
    const size_t pid = (size_t)pyi * p.W + pxi;
    const float rx0 = (float)QX, ry0 = (float)QY;
    const uint2 rg = gs_tile_range(ranges[tile]);
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
                    if (DEPTH) Dp += s2.y * w;
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
        if (RECORD) {                             // the backward pass's per-pixel state; a forward-only render (C3D_GS_FLAG_FORWARD_ONLY, c3d_gs_render_views_raw) has no reader for it
            final_T[pid] = T;
            n_contrib[pid] = (uint32_t)last;
        }
        out_color[pid] = C0 + T * bg[0];
        out_color[P + pid] = C1 + T * bg[1];
        out_color[2 * P + pid] = C2 + T * bg[2];
        if (DEPTH && out_depth) out_depth[pid] = Dp;
        out_alpha[pid] = 1.f - T;                 // sum of the blend weights alpha_i T_i = T_0 - T_final, telescoped (the weights are T_i - T_{i+1})
    }
}

// One launch for V views (grid.y): view v's state lies v * vs bytes behind the pointers of g / b / im; its background and output planes come from vp.
// record_activity: a backward pass may follow -- the launch records the blended (quadrant, splat) pairs AND runs the record-base scan (rbase, einfo of g; state
// g.tmp_scan_a, cleared with the rest of the view's binning state) in its first workgroups.  err: error word of a timed-out look-back (nullptr: the view's own, g.meta[2]).
// scan: false = a RECORD launch WITHOUT the record-base scan (a second rendering of a geometry whose first compositing launch
// has written rbase / einfo already -- C3D_GS_FLAG_KEEP_RECORD_BASES: they do not depend on the pair buffers, and the scan's state is spent).
int gs_launch_composite_fwd(const GsParams& p, const GsGeom& g, const GsBinning& b, int res, GsImage& im, const GsFwdViews& vp, int V, size_t vs,
                            bool record_activity, hipStream_t s, uint32_t* err, bool scan) {
    const int tiles = p.gx * p.gy;
    if (tiles == 0 || V <= 0) return 0;
    uint8_t* pact = record_activity ? gs_pair_activity(b, res) : nullptr;
    ScanWaveJob sj{};
    if (record_activity && scan && p.N > 0)
        sj = ScanWaveJob{g.tiles, g.rbase, g.rect, g.einfo, (uint32_t*)g.tmp_scan_a, err ? err : (uint32_t*)g.meta + 2, (uint32_t)p.N, scan_wave_blocks((size_t)p.N), p.rect4};
    const dim3 grid(sj.blocks + 4 * gs_block_count(p.gx, p.gy), V);      // a multiple of 32 blocks per view: the XCD of a block (dispatch order % 8) does not depend on the view
    bool depth = false;
    for (int v = 0; v < V; v++) depth = depth || vp.depth[v] != nullptr;
#define GS_FWD_LAUNCH(REC_, DEP_) hipLaunchKernelGGL((k_composite_fwd_w<REC_, DEP_>), grid, dim3(64), 0, s, p, b.ranges, b.tval[res], g.rec0, vp, im.final_T, im.n_contrib, pact, b.pair_stride, vs, sj)
    if (record_activity) { if (depth) GS_FWD_LAUNCH(true, true); else GS_FWD_LAUNCH(true, false); }
    else                 { if (depth) GS_FWD_LAUNCH(false, true); else GS_FWD_LAUNCH(false, false); }
#undef GS_FWD_LAUNCH
    C3D_LAUNCH_CHECK();
    return 0;
}

// c3d_gs_forward_nosync for a caller that does not wait for the count: a view that needed more pairs than its buffers hold (C3D_ST_OVERFLOW in the call's first status word)
// gets NaN planes instead of an image that merely looks plausible.  One launch that leaves at once otherwise.
__global__ void __launch_bounds__(256) k_poison_on_overflow(const uint32_t* __restrict__ status, float* __restrict__ color, float* __restrict__ depth, float* __restrict__ alpha, size_t P) {
    if (!(*status & C3D_ST_OVERFLOW)) return;
    const float nan = __int_as_float(0x7FC00000);
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < P; i += (size_t)gridDim.x * 256) {
        color[i] = nan; color[P + i] = nan; color[2 * P + i] = nan; alpha[i] = nan;
        if (depth) depth[i] = nan;
    }
}
int gs_launch_poison_on_overflow(const uint32_t* status, float* color, float* depth, float* alpha, int W, int H, hipStream_t s) {
    const size_t P = (size_t)W * H;
    if (P == 0) return 0;
    hipLaunchKernelGGL(k_poison_on_overflow, dim3(256), dim3(256), 0, s, status, color, depth, alpha, P);
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
