// gs_math.h -- per-Gaussian projection math shared by the forward and backward preprocess kernels.
// Semantics: SURVEY.md Appendix A steps 1-7 (EWA splatting contract of the rasterizer that
// MVs_Algorithms/GaussianSplatting/main_3DGS_renderer.py:927-936 calls); SH basis and signs as in
// shared_utils/sh_utils.py:26-43,57-100; quaternion (w,x,y,z) as in main_3DGS_renderer.py:84-102.
#pragma once
#include "c3d_common.h"

#define GS_SH_C0 0.28209479177387814f
#define GS_SH_C1 0.4886025119029199f
#define GS_SH_C2_0 1.0925484305920792f
#define GS_SH_C2_1 -1.0925484305920792f
#define GS_SH_C2_2 0.31539156525252005f
#define GS_SH_C2_3 -1.0925484305920792f
#define GS_SH_C2_4 0.5462742152960396f
#define GS_SH_C3_0 -0.5900435899266435f
#define GS_SH_C3_1 2.890611442640554f
#define GS_SH_C3_2 -0.4570457994644658f
#define GS_SH_C3_3 0.3731763325901154f
#define GS_SH_C3_4 -0.4570457994644658f
#define GS_SH_C3_5 1.445305721320277f
#define GS_SH_C3_6 -0.5900435899266435f

struct Mat16 { float m[16]; };

// Read-only kernel inputs at a wave-uniform address (camera matrices, camera position) are read through the CONSTANT address space: the compiler may then use scalar loads
// (s_load_dwordx16, counted by lgkmcnt) whatever stores surround them.  As plain `const float*` reads inside the multi-view kernels' loops -- next to stores the compiler cannot
// tell apart from them -- they were VECTOR loads, and the s_waitcnt vmcnt(0) in front of their first use also waited for every store of the previous view (gfx9-family: loads
// and stores share the counter): the projection kernel's stores (0.12 ms of its 0.28 per 8 views) could not overlap the next view's arithmetic
// (profiles/r06/r06aa_projection_store_ablation.txt, r06ab_*).  The memory must not be written by the kernel that reads it this way (the scalar cache is not coherent with it).
typedef const float __attribute__((address_space(4)))* c3d_const_f32p;
__device__ __forceinline__ c3d_const_f32p c3d_as_const(const float* p) { return (c3d_const_f32p)(uintptr_t)p; }
__device__ __forceinline__ Mat16 load_mat16(const float* __restrict__ p) {
    Mat16 r;
#ifdef C3D_NO_CONST_LOADS      // (A/B switch of profiles/r06/r06ab_*)
#pragma unroll
    for (int i = 0; i < 16; i++) r.m[i] = p[i];
#else
    c3d_const_f32p q = c3d_as_const(p);
#pragma unroll
    for (int i = 0; i < 16; i++) r.m[i] = q[i];
#endif
    return r;
}
__device__ __forceinline__ float3 load_vec3_const(const float* __restrict__ p) {
#ifdef C3D_NO_CONST_LOADS
    return make_float3(p[0], p[1], p[2]);
#else
    c3d_const_f32p q = c3d_as_const(p);
    return make_float3(q[0], q[1], q[2]);
#endif
}
// row vector (p,1) times the matrix as stored: out_i = sum_j p_j m[4j+i] + m[12+i]
__device__ __forceinline__ float3 xform4x3(const float3 p, const Mat16& v) {
    return make_float3(v.m[0] * p.x + v.m[4] * p.y + v.m[8] * p.z + v.m[12],
                       v.m[1] * p.x + v.m[5] * p.y + v.m[9] * p.z + v.m[13],
                       v.m[2] * p.x + v.m[6] * p.y + v.m[10] * p.z + v.m[14]);
}
__device__ __forceinline__ float4 xform4x4(const float3 p, const Mat16& v) {
    return make_float4(v.m[0] * p.x + v.m[4] * p.y + v.m[8] * p.z + v.m[12],
                       v.m[1] * p.x + v.m[5] * p.y + v.m[9] * p.z + v.m[13],
                       v.m[2] * p.x + v.m[6] * p.y + v.m[10] * p.z + v.m[14],
                       v.m[3] * p.x + v.m[7] * p.y + v.m[11] * p.z + v.m[15]);
}

__device__ __forceinline__ void quat_to_R(const float4 q, float R[3][3]) {
    const float r = q.x, x = q.y, y = q.z, z = q.w;
    R[0][0] = 1.f - 2.f * (y * y + z * z); R[0][1] = 2.f * (x * y - r * z); R[0][2] = 2.f * (x * z + r * y);
    R[1][0] = 2.f * (x * y + r * z); R[1][1] = 1.f - 2.f * (x * x + z * z); R[1][2] = 2.f * (y * z - r * x);
    R[2][0] = 2.f * (x * z - r * y); R[2][1] = 2.f * (y * z + r * x); R[2][2] = 1.f - 2.f * (x * x + y * y);
}
// Sigma3 = R diag(mod*s)^2 R^T, six unique entries (xx,xy,xz,yy,yz,zz)
__device__ __forceinline__ void cov3d_from_scale_rot(const float3 s, float mod, const float4 q, float c[6]) {
    float R[3][3];
    quat_to_R(q, R);
    const float sv[3] = {mod * s.x, mod * s.y, mod * s.z};
    float Mm[3][3];
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int k = 0; k < 3; k++) Mm[i][k] = R[i][k] * sv[k];
    c[0] = Mm[0][0] * Mm[0][0] + Mm[0][1] * Mm[0][1] + Mm[0][2] * Mm[0][2];
    c[1] = Mm[0][0] * Mm[1][0] + Mm[0][1] * Mm[1][1] + Mm[0][2] * Mm[1][2];
    c[2] = Mm[0][0] * Mm[2][0] + Mm[0][1] * Mm[2][1] + Mm[0][2] * Mm[2][2];
    c[3] = Mm[1][0] * Mm[1][0] + Mm[1][1] * Mm[1][1] + Mm[1][2] * Mm[1][2];
    c[4] = Mm[1][0] * Mm[2][0] + Mm[1][1] * Mm[2][1] + Mm[1][2] * Mm[2][2];
    c[5] = Mm[2][0] * Mm[2][0] + Mm[2][1] * Mm[2][1] + Mm[2][2] * Mm[2][2];
}

// T2 = J(2x3) W(3x3) with the +-1.3 tan(fov) clamp; t = clamped view-space position.
__device__ __forceinline__ void ewa_T2(const float3 mean, const Mat16& v, float tanfovx, float tanfovy, float fx, float fy,
                                       float T2[2][3], float3& t, bool& xin, bool& yin) {
    t = xform4x3(mean, v);
    const float limx = 1.3f * tanfovx, limy = 1.3f * tanfovy;
    const float txtz = t.x / t.z, tytz = t.y / t.z;
    xin = !(txtz < -limx || txtz > limx);
    yin = !(tytz < -limy || tytz > limy);
    t.x = fminf(limx, fmaxf(-limx, txtz)) * t.z;
    t.y = fminf(limy, fmaxf(-limy, tytz)) * t.z;
    const float J00 = fx / t.z, J02 = -(fx * t.x) / (t.z * t.z);
    const float J11 = fy / t.z, J12 = -(fy * t.y) / (t.z * t.z);
#pragma unroll
    for (int k = 0; k < 3; k++) {
        T2[0][k] = J00 * v.m[4 * k + 0] + J02 * v.m[4 * k + 2];
        T2[1][k] = J11 * v.m[4 * k + 1] + J12 * v.m[4 * k + 2];
    }
}
// ST_r[i] = sum_j Sigma[i][j] T2[r][j]
__device__ __forceinline__ void sigma_T(const float c[6], const float T2[2][3], float ST0[3], float ST1[3]) {
    const float S[3][3] = {{c[0], c[1], c[2]}, {c[1], c[3], c[4]}, {c[2], c[4], c[5]}};
#pragma unroll
    for (int i = 0; i < 3; i++) {
        ST0[i] = S[i][0] * T2[0][0] + S[i][1] * T2[0][1] + S[i][2] * T2[0][2];
        ST1[i] = S[i][0] * T2[1][0] + S[i][1] * T2[1][1] + S[i][2] * T2[1][2];
    }
}

__device__ __forceinline__ void sh_basis(int deg, float x, float y, float z, float B[16]) {
    B[0] = GS_SH_C0;
    if (deg > 0) {
        B[1] = -GS_SH_C1 * y; B[2] = GS_SH_C1 * z; B[3] = -GS_SH_C1 * x;
        if (deg > 1) {
            const float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
            B[4] = GS_SH_C2_0 * xy; B[5] = GS_SH_C2_1 * yz; B[6] = GS_SH_C2_2 * (2.f * zz - xx - yy);
            B[7] = GS_SH_C2_3 * xz; B[8] = GS_SH_C2_4 * (xx - yy);
            if (deg > 2) {
                B[9] = GS_SH_C3_0 * y * (3.f * xx - yy);
                B[10] = GS_SH_C3_1 * xy * z;
                B[11] = GS_SH_C3_2 * y * (4.f * zz - xx - yy);
                B[12] = GS_SH_C3_3 * z * (2.f * zz - 3.f * xx - 3.f * yy);
                B[13] = GS_SH_C3_4 * x * (4.f * zz - xx - yy);
                B[14] = GS_SH_C3_5 * z * (xx - yy);
                B[15] = GS_SH_C3_6 * x * (xx - 3.f * yy);
            }
        }
    }
}
__device__ __forceinline__ int sh_ncoef(int deg) { return (deg + 1) * (deg + 1); }

__device__ __forceinline__ float ndc2pix(float v, int S) { return ((v + 1.f) * S - 1.f) * 0.5f; }

__device__ __forceinline__ void tile_rect(float px, float py, int rad, int gx, int gy, int& x0, int& y0, int& x1, int& y1) {
    x0 = min(gx, max(0, (int)((px - rad) / C3D_TILE_X)));
    y0 = min(gy, max(0, (int)((py - rad) / C3D_TILE_Y)));
    x1 = min(gx, max(0, (int)((px + rad + C3D_TILE_X - 1) / C3D_TILE_X)));
    y1 = min(gy, max(0, (int)((py + rad + C3D_TILE_Y - 1) / C3D_TILE_Y)));
}

// Tiles that can receive a non-zero contribution: the dependency's bounding square (radius from the
// larger eigenvalue) intersected with the alpha >= 1/255 box [p - e, p + e] (integer pixel centres).
// Dropping the other tiles is exact: every pixel in them fails the alpha >= 1/255 test.
__device__ __forceinline__ void tile_rect_tight(float px, float py, int rad, float ex, float ey, int gx, int gy,
                                                int& x0, int& y0, int& x1, int& y1) {
    tile_rect(px, py, rad, gx, gy, x0, y0, x1, y1);
    const int lox = (int)ceilf(px - ex), hix = (int)floorf(px + ex);
    const int loy = (int)ceilf(py - ey), hiy = (int)floorf(py + ey);
    x0 = max(x0, max(lox, 0) >> 4); x1 = min(x1, (hix >> 4) + 1);
    y0 = max(y0, max(loy, 0) >> 4); y1 = min(y1, (hiy >> 4) + 1);
    if (x1 < x0) x1 = x0;
    if (y1 < y0) y1 = y0;
}
// half extent of the alpha >= 1/255 box along one axis; var = Sigma_xx (or yy) of the blurred 2D covariance.
// Returns < 0 when the splat can never reach 1/255.  The margin keeps the test conservative under rounding.
__device__ __forceinline__ float alpha_extent(float opacity, float var) {
    const float t = 255.f * opacity;
    if (!(t > 1.f)) return -1.f;
    return sqrtf(2.f * __logf(t) * var) * 1.0005f + 0.02f;
}

// conic coefficients as staged in LDS by the compositing kernels: exp(-q/2) = exp2(GS_CONIC_HALF * (A dx^2 + C dy^2) + GS_CONIC_FULL * B dx dy)
#define GS_CONIC_HALF (-0.72134752044448170368f)   // -log2(e) / 2
#define GS_CONIC_FULL (-1.44269504088896340736f)   // -log2(e)

// clear one bit of a wave-uniform 64-bit mask: one scalar instruction (hipcc expands m &= m - 1 into add / addc / and)
__device__ __forceinline__ uint64_t gs_clear_bit64(uint64_t m, int bit) {
    asm("s_bitset0_b64 %0, %1" : "+s"(m) : "s"(bit));
    return m;
}

// ---- staging one round of a tile's splat list through LDS --------------------------------------------------------------------------------
// `n` (<= blockDim.x) list entries; each record is one 64-B line (gs_internal.h) of which 48 B are used.  Lane q fetches 16-B part q % 3 of
// entry q / 3: consecutive lanes sit on consecutive parts of one record, so a wave instruction touches 22 lines instead of the 64 a
// one-lane-per-record gather of a separate array would, and the whole round needs one line per splat instead of three.
// dir = +1: entries list[0], list[1], ...; dir = -1: list[0], list[-1], ... (the backward pass walks a tile's list back to front).
__device__ __forceinline__ void gs_stage_round(const uint32_t* __restrict__ list, int n, const float4* __restrict__ rec, float4* s0, float4* s1, float4* s2, int dir = 1) {
    for (int q = threadIdx.x; q < 3 * n; q += blockDim.x) {
        const int e = q / 3, part = q - 3 * e;
        const float4 v = rec[4 * (size_t)list[e * dir] + part];
        (part == 0 ? s0 : (part == 1 ? s1 : s2))[e] = v;
    }
}

// ---- block -> tile ------------------------------------------------------------------------------------------
// The dispatcher deals workgroups to the 8 XCDs round-robin (b & 7).  Each XCD walks the 2x2-tile supertiles q = xcd, xcd + 8, ... in row-major
// order: neighbouring tiles, which share most of their splats, meet in one XCD's L2, while every XCD covers the whole image -- with contiguous
// bands per XCD a centred object loads the XCDs that own the middle rows and idles the rest.
__device__ __forceinline__ bool gs_block_tile(int b, int gx, int gy, int& tx, int& ty, int sh = 1) {
    const int sgx = (gx + (1 << sh) - 1) >> sh;
    const int per = 1 << (2 * sh);
    const int i = b >> 3, q = (b & 7) + 8 * (i >> (2 * sh)), sub = i & (per - 1);
    const int sy = q / sgx, sx = q - sy * sgx;
    tx = (sx << sh) + (sub & ((1 << sh) - 1));
    ty = (sy << sh) + (sub >> sh);
    return tx < gx && ty < gy;
}
// (supertile sizes 1x1 ... 4x4 measure alike, 8x8 loses the balance again: profiles/r02*; sh = 1 is what every launch uses)
static inline int gs_block_count(int gx, int gy, int sh = 1) {
    const int S = ((gx + (1 << sh) - 1) >> sh) * ((gy + (1 << sh) - 1) >> sh);
    return 8 * (1 << (2 * sh)) * ((S + 7) / 8);
}

// ---- which of a tile's four 8x8 quadrants can a splat touch? ---------------------------------------------
// Exact test: alpha >= 1/255  <=>  q(d) = A dx^2 + 2B dx dy + C dy^2 <= 2 ln(255 o); a quadrant is kept iff the minimum of the
// (convex) form q over its pixel rectangle is below that bound.  The cheap alpha-box test rejects most quadrants first.
// Conservative by construction (margin on the bound), so skipping a quadrant never changes a pixel.
__device__ __forceinline__ float quad_form_min_on_rect(float A, float B, float C, float X, float Y, float xl, float xh, float yl, float yh) {
    const float cxp = fminf(fmaxf(X, xl), xh), cyp = fminf(fmaxf(Y, yl), yh);
    if (cxp == X && cyp == Y) return 0.f;                       // centre inside the rectangle
    float best = 3.0e38f;
    const float invC = 1.f / C, invA = 1.f / A;
#pragma unroll
    for (int e = 0; e < 2; e++) {                               // vertical edges x = xl, xh: minimise over y
        const float dx = (e ? xh : xl) - X;
        const float dy = fminf(fmaxf(-B * dx * invC, yl - Y), yh - Y);
        best = fminf(best, A * dx * dx + 2.f * B * dx * dy + C * dy * dy);
    }
#pragma unroll
    for (int e = 0; e < 2; e++) {                               // horizontal edges y = yl, yh: minimise over x
        const float dy = (e ? yh : yl) - Y;
        const float dx = fminf(fmaxf(-B * dy * invA, xl - X), xh - X);
        best = fminf(best, A * dx * dx + 2.f * B * dx * dy + C * dy * dy);
    }
    return best;
}
// Exponent of a splat at a pixel, scaled by log2(e): a0 = (.., .., -log2e/2 A, -log2e B), c = -log2e/2 C, (dx, dy) = splat centre - pixel.
// ONE spelling with the contractions written out, shared by the forward and the backward compositing kernels: the backward pass walks the
// (quadrant, splat) pairs the forward pass recorded as blended, so both must reach bit-identical alpha decisions.
__device__ __forceinline__ float gs_power(const float4 a0, float c, float dx, float dy) {
    return __builtin_fmaf(dx, __builtin_fmaf(a0.z, dx, a0.w * dy), (c * dy) * dy);
}
// one pixel rectangle [rx0, rx0 + w] x [ry0, ry0 + h] (inclusive pixel centres): can the splat reach alpha >= 1/255 anywhere on it?
__device__ __forceinline__ bool gs_rect_hit(const float4 a0, const float4 a1, const float4 a2, float rx0, float ry0, float w = 7.f, float h = 7.f) {
    // a0 = (px, py, A, B)  a1 = (C, opacity, ..)  a2 = (.., .., ex, ey)
    const float xl = a0.x - a2.z, xh = a0.x + a2.z, yl = a0.y - a2.w, yh = a0.y + a2.w;
    if (!(xh >= rx0 && xl <= rx0 + w && yh >= ry0 && yl <= ry0 + h)) return false;
    const float thr = 2.f * __logf(255.f * a1.y) * 1.0005f + 1e-3f;
    return quad_form_min_on_rect(a0.z, a0.w, a1.x, a0.x, a0.y, rx0, rx0 + w, ry0, ry0 + h) <= thr;
}
// ---- lane selects on wave masks held in SGPR pairs ----------------------------------------------------------------------------------
// Written as asm because hipcc picks the VOP2 form that reads VCC (v_cndmask_b32_e32 ..., vcc), which issues ~5x slower on gfx950 than the
// VOP3 form with an SGPR-pair mask (profiles/r01f_valu_rate_microbench.txt: 22.9 vs 4.7 cycles per wave-instruction).
__device__ __forceinline__ float sel64(uint64_t m, float if_set, float otherwise) {
    float r;
    asm("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(r) : "v"(otherwise), "v"(if_set), "s"(m));
    return r;
}
__device__ __forceinline__ float sel64z(uint64_t m, float if_set) {   // 0 where the mask is clear
    float r;
    asm("v_cndmask_b32_e64 %0, 0, %1, %2" : "=v"(r) : "v"(if_set), "s"(m));
    return r;
}
__device__ __forceinline__ int sel64i(uint64_t m, int if_set, int otherwise) {
    int r;
    asm("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(r) : "v"(otherwise), "v"(if_set), "s"(m));
    return r;
}

// ---- spherical harmonics without arrays (keeps the per-Gaussian kernels out of scratch) -------------------
// SH_FOREACH(deg, x, y, z, TERM) invokes TERM(k, B_k, dB_k/dx, dB_k/dy, dB_k/dz) for every active coefficient.
#define SH_FOREACH(deg, x, y, z, TERM)                                                                                    \
    do {                                                                                                                  \
        TERM(0, GS_SH_C0, 0.f, 0.f, 0.f);                                                                                 \
        if ((deg) > 0) {                                                                                                  \
            TERM(1, -GS_SH_C1 * (y), 0.f, -GS_SH_C1, 0.f);                                                                \
            TERM(2, GS_SH_C1 * (z), 0.f, 0.f, GS_SH_C1);                                                                  \
            TERM(3, -GS_SH_C1 * (x), -GS_SH_C1, 0.f, 0.f);                                                                \
            if ((deg) > 1) {                                                                                              \
                const float xx_ = (x) * (x), yy_ = (y) * (y), zz_ = (z) * (z);                                            \
                TERM(4, GS_SH_C2_0 * (x) * (y), GS_SH_C2_0 * (y), GS_SH_C2_0 * (x), 0.f);                                 \
                TERM(5, GS_SH_C2_1 * (y) * (z), 0.f, GS_SH_C2_1 * (z), GS_SH_C2_1 * (y));                                 \
                TERM(6, GS_SH_C2_2 * (2.f * zz_ - xx_ - yy_), GS_SH_C2_2 * (-2.f * (x)), GS_SH_C2_2 * (-2.f * (y)), GS_SH_C2_2 * (4.f * (z))); \
                TERM(7, GS_SH_C2_3 * (x) * (z), GS_SH_C2_3 * (z), 0.f, GS_SH_C2_3 * (x));                                 \
                TERM(8, GS_SH_C2_4 * (xx_ - yy_), GS_SH_C2_4 * (2.f * (x)), GS_SH_C2_4 * (-2.f * (y)), 0.f);              \
                if ((deg) > 2) {                                                                                          \
                    TERM(9, GS_SH_C3_0 * (y) * (3.f * xx_ - yy_), GS_SH_C3_0 * 6.f * (x) * (y), GS_SH_C3_0 * (3.f * xx_ - 3.f * yy_), 0.f); \
                    TERM(10, GS_SH_C3_1 * (x) * (y) * (z), GS_SH_C3_1 * (y) * (z), GS_SH_C3_1 * (x) * (z), GS_SH_C3_1 * (x) * (y)); \
                    TERM(11, GS_SH_C3_2 * (y) * (4.f * zz_ - xx_ - yy_), GS_SH_C3_2 * (-2.f * (x) * (y)), GS_SH_C3_2 * (4.f * zz_ - xx_ - 3.f * yy_), GS_SH_C3_2 * (8.f * (y) * (z))); \
                    TERM(12, GS_SH_C3_3 * (z) * (2.f * zz_ - 3.f * xx_ - 3.f * yy_), GS_SH_C3_3 * (-6.f * (x) * (z)), GS_SH_C3_3 * (-6.f * (y) * (z)), GS_SH_C3_3 * (6.f * zz_ - 3.f * xx_ - 3.f * yy_)); \
                    TERM(13, GS_SH_C3_4 * (x) * (4.f * zz_ - xx_ - yy_), GS_SH_C3_4 * (4.f * zz_ - 3.f * xx_ - yy_), GS_SH_C3_4 * (-2.f * (x) * (y)), GS_SH_C3_4 * (8.f * (x) * (z))); \
                    TERM(14, GS_SH_C3_5 * (z) * (xx_ - yy_), GS_SH_C3_5 * (2.f * (x) * (z)), GS_SH_C3_5 * (-2.f * (y) * (z)), GS_SH_C3_5 * (xx_ - yy_)); \
                    TERM(15, GS_SH_C3_6 * (x) * (xx_ - 3.f * yy_), GS_SH_C3_6 * (3.f * xx_ - 3.f * yy_), GS_SH_C3_6 * (-6.f * (x) * (y)), 0.f); \
                }                                                                                                         \
            }                                                                                                             \
        }                                                                                                                 \
    } while (0)

// ---- block-cooperative staging of the [N][16][3] SH tensor through LDS ------------------------------------
// A 256-lane workgroup owns 256 consecutive Gaussians = 48 KiB of contiguous SH data.  It is moved with
// 16-byte loads/stores, consecutive lanes on consecutive addresses (1 KiB per wave instruction), into a
// row-padded LDS image (49 floats per Gaussian: lane t then reads its own row conflict-free).
#define SH_ROW 49
#define SH_M3 48
__device__ __forceinline__ void sh_stage_in(const float* __restrict__ shs, size_t g0, int count, float* lds) {
    const float4* src = reinterpret_cast<const float4*>(shs + g0 * SH_M3);
    const int n4 = count * (SH_M3 / 4);
    for (int e4 = threadIdx.x; e4 < n4; e4 += blockDim.x) {
        const float4 v = src[e4];
        const int e = e4 * 4, gl = e / SH_M3, k = e - gl * SH_M3;   // 48 % 4 == 0: the four floats stay in one row
        float* d = lds + gl * SH_ROW + k;
        d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
    }
}
// split storage of the reference's GaussianModel: f_dc [N,1,3] and f_rest [N,K-1,3] (main_3DGS_renderer.py:314-317 concatenates them
// on every render; staging both straight into the same LDS rows removes that copy).  REST3 = 3 (K - 1) floats of f_rest per Gaussian: 45 for the
// degree-3 storage the trainer uses, 24 / 9 / 0 for PLYs of degree 2 / 1 / 0 (mesh_processer/mesh_utils.py:346-350 loads any of them; LGM writes degree 0).
// Coefficient k of a Gaussian sits at lds[row * SH_ROW + 3 k + channel] whatever K is; rows are not cleared beyond 3 K (nothing reads there: the active
// degree never exceeds the storage degree).
#define SH_REST 45
__host__ __device__ constexpr int sh_rest3_of(int coeffs) { return 3 * (coeffs - 1); }
template <int REST3 = SH_REST>
__device__ __forceinline__ void sh_stage_in_split(const float* __restrict__ f_dc, const float* __restrict__ f_rest, size_t g0, int count, float* lds) {
    if (REST3 > 0) {
        const float4* src = reinterpret_cast<const float4*>(f_rest + g0 * REST3);      // g0 is a multiple of 4: 16-byte aligned for every REST3
        const int n4 = (count * REST3) / 4, tail = (count * REST3) & 3;
        for (int e4 = threadIdx.x; e4 < n4; e4 += blockDim.x) {
            const float4 v = src[e4];
            const float vv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int c = 0; c < 4; c++) { const int e = e4 * 4 + c, gl = e / (REST3 > 0 ? REST3 : 1); lds[gl * SH_ROW + 3 + (e - gl * REST3)] = vv[c]; }
        }
        if ((int)threadIdx.x < tail) { const int e = n4 * 4 + threadIdx.x, gl = e / (REST3 > 0 ? REST3 : 1); lds[gl * SH_ROW + 3 + (e - gl * REST3)] = f_rest[g0 * REST3 + e]; }
    }
    for (int e = threadIdx.x; e < count * 3; e += blockDim.x) { const int gl = e / 3; lds[gl * SH_ROW + (e - gl * 3)] = f_dc[g0 * 3 + e]; }
}
template <bool ACC, int REST3 = SH_REST>
__device__ __forceinline__ void sh_stage_out_split(float* __restrict__ d_dc, float* __restrict__ d_rest, size_t g0, int count, const float* lds) {
    if (REST3 > 0) {
        float4* dst = reinterpret_cast<float4*>(d_rest + g0 * REST3);
        const int n4 = (count * REST3) / 4, tail = (count * REST3) & 3;
        for (int e4 = threadIdx.x; e4 < n4; e4 += blockDim.x) {
            float vv[4];
#pragma unroll
            for (int c = 0; c < 4; c++) { const int e = e4 * 4 + c, gl = e / (REST3 > 0 ? REST3 : 1); vv[c] = lds[gl * SH_ROW + 3 + (e - gl * REST3)]; }
            float4 o = make_float4(vv[0], vv[1], vv[2], vv[3]);
            if (ACC) { const float4 old = dst[e4]; o.x += old.x; o.y += old.y; o.z += old.z; o.w += old.w; }
            dst[e4] = o;
        }
        if ((int)threadIdx.x < tail) {
            const int e = n4 * 4 + threadIdx.x, gl = e / (REST3 > 0 ? REST3 : 1);
            const float v = lds[gl * SH_ROW + 3 + (e - gl * REST3)];
            d_rest[g0 * REST3 + e] = ACC ? d_rest[g0 * REST3 + e] + v : v;
        }
    }
    for (int e = threadIdx.x; e < count * 3; e += blockDim.x) {
        const int gl = e / 3;
        const float v = lds[gl * SH_ROW + (e - gl * 3)];
        d_dc[g0 * 3 + e] = ACC ? d_dc[g0 * 3 + e] + v : v;
    }
}
__device__ __forceinline__ void sh_stage_out(float* __restrict__ dst_sh, size_t g0, int count, const float* lds) {
    float4* dst = reinterpret_cast<float4*>(dst_sh + g0 * SH_M3);
    const int n4 = count * (SH_M3 / 4);
    for (int e4 = threadIdx.x; e4 < n4; e4 += blockDim.x) {
        const int e = e4 * 4, gl = e / SH_M3, k = e - gl * SH_M3;
        const float* d = lds + gl * SH_ROW + k;
        dst[e4] = make_float4(d[0], d[1], d[2], d[3]);
    }
}
