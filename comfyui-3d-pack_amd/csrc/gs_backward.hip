// gs_backward.hip -- 3DGS backward: per-tile compositing gradients (A7) and per-Gaussian chain rule (A8).
// Contract: SURVEY.md section 2.3-A (A7, A8) and Appendix A; the outputs are the tensors the autograd
// Function behind main_3DGS_renderer.py:927-936 (reference call site) must return.
#include "gs_internal.h"
#include "gs_math.h"

// --- wave64 sum via DPP: result valid in lanes 48..63 (read lane 63) ---------------------------------
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_f(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, ROW_MASK, 0xF, false));
}
__device__ __forceinline__ float wave_sum_to_lane63(float v) {
    v += dpp_f<0xB1, 0xF>(v);   // quad_perm [1,0,3,2]
    v += dpp_f<0x4E, 0xF>(v);   // quad_perm [2,3,0,1]
    v += dpp_f<0x141, 0xF>(v);  // row_half_mirror
    v += dpp_f<0x140, 0xF>(v);  // row_mirror      -> every lane holds its 16-lane row sum
    v += dpp_f<0x142, 0xA>(v);  // row_bcast15 into rows 1,3
    v += dpp_f<0x143, 0xC>(v);  // row_bcast31 into rows 2,3 -> row 3 holds the wave sum
    return v;
}

// ------------------------------------------------------------------------------------------
// A7 composite backward: same tiling as the forward pass, splats visited back to front.  Every lane
// re-derives alpha/T for its pixel; the ten per-splat partial gradients are summed across the wave
// with DPP adds and leave the wave as ONE atomic per quantity (256x fewer atomics than one per pixel).
// ------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_composite_bwd(GsParams p, const uint2* __restrict__ ranges, const uint32_t* __restrict__ point_list,
                                                        const float4* __restrict__ rec0, const float4* __restrict__ rec1,
                                                        const float2* __restrict__ rec2, const float* __restrict__ final_T,
                                                        const uint32_t* __restrict__ n_contrib, const float* __restrict__ dL_dcolor,
                                                        const float* __restrict__ dL_ddepth, const float* __restrict__ dL_dalpha_px,
                                                        float* __restrict__ dL_dmean2D, float* __restrict__ dL_dconic,
                                                        float* __restrict__ dL_dopacity, float* __restrict__ dL_dcolors,
                                                        float* __restrict__ dL_ddepths, int chunk) {
    __shared__ float4 s0[256];
    __shared__ float4 s1[256];
    __shared__ float2 s2[256];
    __shared__ uint32_t sid[256];
    const int tile = (blockIdx.x & 7) * chunk + (blockIdx.x >> 3);
    if (tile >= p.gx * p.gy) return;
    const int tx = tile % p.gx, ty = tile / p.gx;
    const int lx = threadIdx.x & 15, ly = threadIdx.x >> 4;
    const int pxi = tx * C3D_TILE_X + lx, pyi = ty * C3D_TILE_Y + ly;
    const bool inside = pxi < p.W && pyi < p.H;
    const float pxf = (float)pxi, pyf = (float)pyi;
    const uint2 rg = ranges[tile];
    const size_t P = (size_t)p.W * p.H, pid = (size_t)pyi * p.W + pxi;
    const int lane = c3d_lane();

    const float T_final = inside ? final_T[pid] : 0.f;
    float T = T_final;
    const int last = inside ? (int)n_contrib[pid] : 0;
    float dLp0 = 0.f, dLp1 = 0.f, dLp2 = 0.f, dLd = 0.f, dLa = 0.f;
    if (inside) {
        dLp0 = dL_dcolor[pid]; dLp1 = dL_dcolor[P + pid]; dLp2 = dL_dcolor[2 * P + pid];
        dLd = dL_ddepth ? dL_ddepth[pid] : 0.f;
        dLa = dL_dalpha_px ? dL_dalpha_px[pid] : 0.f;
    }
    const float bg_dot = p.bg[0] * dLp0 + p.bg[1] * dLp1 + p.bg[2] * dLp2;
    float acc0 = 0.f, acc1 = 0.f, acc2 = 0.f, accd = 0.f, acca = 0.f;
    float last_alpha = 0.f, lc0 = 0.f, lc1 = 0.f, lc2 = 0.f, ld = 0.f;
    const float ddelx_dx = 0.5f * p.W, ddely_dy = 0.5f * p.H;

    // the whole tile can skip list positions no pixel reached
    __shared__ int s_maxlast;
    if (threadIdx.x == 0) s_maxlast = 0;
    __syncthreads();
    atomicMax(&s_maxlast, last);
    __syncthreads();
    const int upto = s_maxlast;   // positions [0, upto) matter

    for (int base = 0; base < upto; base += 256) {
        __syncthreads();
        const int n = min(256, upto - base);
        if ((int)threadIdx.x < n) {
            const uint32_t gid = point_list[rg.x + (upto - 1 - base - threadIdx.x)];
            sid[threadIdx.x] = gid;
            s0[threadIdx.x] = rec0[gid]; s1[threadIdx.x] = rec1[gid]; s2[threadIdx.x] = rec2[gid];
        }
        __syncthreads();
        for (int j = 0; j < n; j++) {
            const int k = upto - 1 - base - j;   // list position of this splat
            const float4 a0 = s0[j], a1 = s1[j];
            const float2 a2 = s2[j];
            const float dx = a0.x - pxf, dy = a0.y - pyf;
            const float power = -0.5f * (a0.z * dx * dx + a1.x * dy * dy) - a0.w * dx * dy;
            const float G = __expf(power);
            const float alpha = fminf(0.99f, a1.y * G);
            const bool act = (k < last) && (power <= 0.f) && (alpha >= 1.f / 255.f);
            if (__ballot(act) == 0ull) continue;   // wave-uniform skip

            float g_c0 = 0.f, g_c1 = 0.f, g_c2 = 0.f, g_d = 0.f, g_mx = 0.f, g_my = 0.f, g_cx = 0.f, g_cy = 0.f, g_cw = 0.f, g_o = 0.f;
            if (act) {
                T = T / (1.f - alpha);
                const float dch = alpha * T;
                float dL_dalpha;
                acc0 = last_alpha * lc0 + (1.f - last_alpha) * acc0; lc0 = a1.z;
                acc1 = last_alpha * lc1 + (1.f - last_alpha) * acc1; lc1 = a1.w;
                acc2 = last_alpha * lc2 + (1.f - last_alpha) * acc2; lc2 = a2.x;
                dL_dalpha = (a1.z - acc0) * dLp0 + (a1.w - acc1) * dLp1 + (a2.x - acc2) * dLp2;
                g_c0 = dch * dLp0; g_c1 = dch * dLp1; g_c2 = dch * dLp2;
                accd = last_alpha * ld + (1.f - last_alpha) * accd; ld = a2.y;
                dL_dalpha += (a2.y - accd) * dLd;
                g_d = dch * dLd;
                acca = last_alpha + (1.f - last_alpha) * acca;
                dL_dalpha += (1.f - acca) * dLa;
                dL_dalpha *= T;
                last_alpha = alpha;
                dL_dalpha += (-T_final / (1.f - alpha)) * bg_dot;
                const float dL_dG = a1.y * dL_dalpha;
                const float gdx = G * dx, gdy = G * dy;
                g_mx = dL_dG * (-gdx * a0.z - gdy * a0.w) * ddelx_dx;
                g_my = dL_dG * (-gdy * a1.x - gdx * a0.w) * ddely_dy;
                g_cx = -0.5f * gdx * dx * dL_dG;
                g_cy = -0.5f * gdx * dy * dL_dG;
                g_cw = -0.5f * gdy * dy * dL_dG;
                g_o = G * dL_dalpha;
            }
            g_c0 = wave_sum_to_lane63(g_c0); g_c1 = wave_sum_to_lane63(g_c1); g_c2 = wave_sum_to_lane63(g_c2);
            g_d = wave_sum_to_lane63(g_d);
            g_mx = wave_sum_to_lane63(g_mx); g_my = wave_sum_to_lane63(g_my);
            g_cx = wave_sum_to_lane63(g_cx); g_cy = wave_sum_to_lane63(g_cy); g_cw = wave_sum_to_lane63(g_cw);
            g_o = wave_sum_to_lane63(g_o);
            if (lane == 63) {
                const uint32_t gid = sid[j];
                atomicAdd(&dL_dcolors[3 * gid + 0], g_c0);
                atomicAdd(&dL_dcolors[3 * gid + 1], g_c1);
                atomicAdd(&dL_dcolors[3 * gid + 2], g_c2);
                atomicAdd(&dL_ddepths[gid], g_d);
                atomicAdd(&dL_dmean2D[3 * gid + 0], g_mx);
                atomicAdd(&dL_dmean2D[3 * gid + 1], g_my);
                atomicAdd(&dL_dconic[4 * gid + 0], g_cx);
                atomicAdd(&dL_dconic[4 * gid + 1], g_cy);
                atomicAdd(&dL_dconic[4 * gid + 3], g_cw);
                atomicAdd(&dL_dopacity[gid], g_o);
            }
        }
    }
}

int gs_launch_composite_bwd(const GsParams& p, const GsGeom& g, const GsBinning& b, int res, const GsImage& im,
                            const float* dL_dcolor, const float* dL_ddepth, const float* dL_dalpha,
                            float* dL_dmean2D, float* dL_dconic, float* dL_dopacity, float* dL_dcolors,
                            float* dL_ddepths, hipStream_t s) {
    const int tiles = p.gx * p.gy;
    if (tiles == 0) return 0;
    const int chunk = c3d_cdiv(tiles, 8);
    hipLaunchKernelGGL(k_composite_bwd, dim3(chunk * 8), dim3(256), 0, s, p, b.ranges, b.tval[res], g.rec0, g.rec1, g.rec2,
                       im.final_T, im.n_contrib, dL_dcolor, dL_ddepth, dL_dalpha, dL_dmean2D, dL_dconic, dL_dopacity,
                       dL_dcolors, dL_ddepths, chunk);
    C3D_LAUNCH_CHECK();
    return 0;
}

// ------------------------------------------------------------------------------------------
// A8 preprocess backward: one lane per Gaussian, pure streaming.
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ void sh_basis_grad(int deg, float x, float y, float z, float dB[16][3]) {
#pragma unroll
    for (int k = 0; k < 16; k++) { dB[k][0] = 0.f; dB[k][1] = 0.f; dB[k][2] = 0.f; }
    if (deg > 0) {
        dB[1][1] = -GS_SH_C1; dB[2][2] = GS_SH_C1; dB[3][0] = -GS_SH_C1;
        if (deg > 1) {
            const float xx = x * x, yy = y * y, zz = z * z;
            dB[4][0] = GS_SH_C2_0 * y; dB[4][1] = GS_SH_C2_0 * x;
            dB[5][1] = GS_SH_C2_1 * z; dB[5][2] = GS_SH_C2_1 * y;
            dB[6][0] = GS_SH_C2_2 * (-2.f * x); dB[6][1] = GS_SH_C2_2 * (-2.f * y); dB[6][2] = GS_SH_C2_2 * (4.f * z);
            dB[7][0] = GS_SH_C2_3 * z; dB[7][2] = GS_SH_C2_3 * x;
            dB[8][0] = GS_SH_C2_4 * (2.f * x); dB[8][1] = GS_SH_C2_4 * (-2.f * y);
            if (deg > 2) {
                dB[9][0] = GS_SH_C3_0 * 6.f * x * y; dB[9][1] = GS_SH_C3_0 * (3.f * xx - 3.f * yy);
                dB[10][0] = GS_SH_C3_1 * y * z; dB[10][1] = GS_SH_C3_1 * x * z; dB[10][2] = GS_SH_C3_1 * x * y;
                dB[11][0] = GS_SH_C3_2 * (-2.f * x * y); dB[11][1] = GS_SH_C3_2 * (4.f * zz - xx - 3.f * yy); dB[11][2] = GS_SH_C3_2 * (8.f * y * z);
                dB[12][0] = GS_SH_C3_3 * (-6.f * x * z); dB[12][1] = GS_SH_C3_3 * (-6.f * y * z); dB[12][2] = GS_SH_C3_3 * (6.f * zz - 3.f * xx - 3.f * yy);
                dB[13][0] = GS_SH_C3_4 * (4.f * zz - 3.f * xx - yy); dB[13][1] = GS_SH_C3_4 * (-2.f * x * y); dB[13][2] = GS_SH_C3_4 * (8.f * x * z);
                dB[14][0] = GS_SH_C3_5 * (2.f * x * z); dB[14][1] = GS_SH_C3_5 * (-2.f * y * z); dB[14][2] = GS_SH_C3_5 * (xx - yy);
                dB[15][0] = GS_SH_C3_6 * (3.f * xx - 3.f * yy); dB[15][1] = GS_SH_C3_6 * (-6.f * x * y);
            }
        }
    }
}

__global__ void __launch_bounds__(256) k_preprocess_bwd(GsParams p, GsGeom g, const int* __restrict__ radii, const float* __restrict__ means3D,
                                                         const float* __restrict__ shs, const float* __restrict__ colors_precomp,
                                                         const float* __restrict__ scales, const float* __restrict__ rotations,
                                                         const float* __restrict__ cov3D_precomp, const float* __restrict__ dL_dmean2D,
                                                         const float* __restrict__ dL_dconic, const float* __restrict__ dL_dcolors,
                                                         const float* __restrict__ dL_ddepths, float* __restrict__ dL_dmeans3D,
                                                         float* __restrict__ dL_dcov3D, float* __restrict__ dL_dsh,
                                                         float* __restrict__ dL_dscales, float* __restrict__ dL_drots) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= p.N) return;
    if (radii[idx] <= 0) {
        // culled: this kernel owns its outputs (callers allocate them uninitialised)
        dL_dmeans3D[3 * idx] = 0.f; dL_dmeans3D[3 * idx + 1] = 0.f; dL_dmeans3D[3 * idx + 2] = 0.f;
        if (dL_dcov3D) {
#pragma unroll
            for (int i = 0; i < 6; i++) dL_dcov3D[6 * idx + i] = 0.f;
        }
        if (!colors_precomp) {
            float* dsh = dL_dsh + (size_t)idx * p.M * 3;
            for (int k = 0; k < 3 * p.M; k++) dsh[k] = 0.f;
        }
        if (!cov3D_precomp) {
            dL_dscales[3 * idx] = 0.f; dL_dscales[3 * idx + 1] = 0.f; dL_dscales[3 * idx + 2] = 0.f;
            *reinterpret_cast<float4*>(dL_drots + 4 * idx) = make_float4(0.f, 0.f, 0.f, 0.f);
        }
        return;
    }
    const Mat16 V = load_mat16(p.view), PJ = load_mat16(p.proj);
    const float3 m = make_float3(means3D[3 * idx], means3D[3 * idx + 1], means3D[3 * idx + 2]);
    float c3[6];
    float3 sc = make_float3(0, 0, 0);
    float4 q = make_float4(1, 0, 0, 0);
    if (cov3D_precomp) {
#pragma unroll
        for (int i = 0; i < 6; i++) c3[i] = cov3D_precomp[6 * idx + i];
    } else {
        sc = make_float3(scales[3 * idx], scales[3 * idx + 1], scales[3 * idx + 2]);
        q = *reinterpret_cast<const float4*>(rotations + 4 * idx);
        cov3d_from_scale_rot(sc, p.scale_modifier, q, c3);
    }
    float T2[2][3], ST0[3], ST1[3];
    float3 t; bool xin, yin;
    ewa_T2(m, V, p.tanfovx, p.tanfovy, p.focal_x, p.focal_y, T2, t, xin, yin);
    sigma_T(c3, T2, ST0, ST1);
    const float a = T2[0][0] * ST0[0] + T2[0][1] * ST0[1] + T2[0][2] * ST0[2] + 0.3f;
    const float b = T2[0][0] * ST1[0] + T2[0][1] * ST1[1] + T2[0][2] * ST1[2];
    const float c = T2[1][0] * ST1[0] + T2[1][1] * ST1[1] + T2[1][2] * ST1[2] + 0.3f;
    const float dcx = dL_dconic[4 * idx], dcy = dL_dconic[4 * idx + 1], dcz = dL_dconic[4 * idx + 3];
    const float denom = a * c - b * b;
    const float d2i = 1.f / (denom * denom + 0.0000001f);
    float dL_da = 0.f, dL_db = 0.f, dL_dc = 0.f;
    float dcov[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (d2i != 0.f) {
        dL_da = d2i * (-c * c * dcx + 2.f * b * c * dcy + (denom - a * c) * dcz);
        dL_dc = d2i * (-a * a * dcz + 2.f * a * b * dcy + (denom - a * c) * dcx);
        dL_db = d2i * 2.f * (b * c * dcx - (denom + 2.f * b * b) * dcy + a * b * dcz);
        dcov[0] = T2[0][0] * T2[0][0] * dL_da + T2[0][0] * T2[1][0] * dL_db + T2[1][0] * T2[1][0] * dL_dc;
        dcov[3] = T2[0][1] * T2[0][1] * dL_da + T2[0][1] * T2[1][1] * dL_db + T2[1][1] * T2[1][1] * dL_dc;
        dcov[5] = T2[0][2] * T2[0][2] * dL_da + T2[0][2] * T2[1][2] * dL_db + T2[1][2] * T2[1][2] * dL_dc;
        dcov[1] = 2.f * T2[0][0] * T2[0][1] * dL_da + (T2[0][0] * T2[1][1] + T2[0][1] * T2[1][0]) * dL_db + 2.f * T2[1][0] * T2[1][1] * dL_dc;
        dcov[2] = 2.f * T2[0][0] * T2[0][2] * dL_da + (T2[0][0] * T2[1][2] + T2[0][2] * T2[1][0]) * dL_db + 2.f * T2[1][0] * T2[1][2] * dL_dc;
        dcov[4] = 2.f * T2[0][2] * T2[0][1] * dL_da + (T2[0][1] * T2[1][2] + T2[0][2] * T2[1][1]) * dL_db + 2.f * T2[1][1] * T2[1][2] * dL_dc;
    }
    if (dL_dcov3D) {
#pragma unroll
        for (int i = 0; i < 6; i++) dL_dcov3D[6 * idx + i] = dcov[i];
    }
    float dT[2][3];
#pragma unroll
    for (int k = 0; k < 3; k++) {
        dT[0][k] = 2.f * ST0[k] * dL_da + ST1[k] * dL_db;
        dT[1][k] = 2.f * ST1[k] * dL_dc + ST0[k] * dL_db;
    }
    const float dJ00 = V.m[0] * dT[0][0] + V.m[4] * dT[0][1] + V.m[8] * dT[0][2];
    const float dJ02 = V.m[2] * dT[0][0] + V.m[6] * dT[0][1] + V.m[10] * dT[0][2];
    const float dJ11 = V.m[1] * dT[1][0] + V.m[5] * dT[1][1] + V.m[9] * dT[1][2];
    const float dJ12 = V.m[2] * dT[1][0] + V.m[6] * dT[1][1] + V.m[10] * dT[1][2];
    const float tz = 1.f / t.z, tz2 = tz * tz, tz3 = tz2 * tz;
    const float dtx = (xin ? 1.f : 0.f) * (-p.focal_x * tz2 * dJ02);
    const float dty = (yin ? 1.f : 0.f) * (-p.focal_y * tz2 * dJ12);
    const float dtz = -p.focal_x * tz2 * dJ00 - p.focal_y * tz2 * dJ11 + (2.f * p.focal_x * t.x) * tz3 * dJ02 + (2.f * p.focal_y * t.y) * tz3 * dJ12;
    float dmean[3];
#pragma unroll
    for (int j = 0; j < 3; j++) dmean[j] = V.m[4 * j] * dtx + V.m[4 * j + 1] * dty + V.m[4 * j + 2] * dtz;

    // screen-space mean (NDC-scaled gradient) -> 3D mean
    const float4 mh = xform4x4(m, PJ);
    const float mw = 1.f / (mh.w + 0.0000001f);
    const float mul1 = mh.x * mw * mw, mul2 = mh.y * mw * mw;
    const float g2x = dL_dmean2D[3 * idx], g2y = dL_dmean2D[3 * idx + 1];
#pragma unroll
    for (int j = 0; j < 3; j++)
        dmean[j] += (PJ.m[4 * j] * mw - PJ.m[4 * j + 3] * mul1) * g2x + (PJ.m[4 * j + 1] * mw - PJ.m[4 * j + 3] * mul2) * g2y;
    // depth output -> 3D mean
    {
        const float mul3 = V.m[2] * m.x + V.m[6] * m.y + V.m[10] * m.z + V.m[14];
        const float gd = dL_ddepths[idx];
#pragma unroll
        for (int j = 0; j < 3; j++) dmean[j] += (V.m[4 * j + 2] - V.m[4 * j + 3] * mul3) * gd;
    }
    // colour -> SH coefficients and view direction
    if (!colors_precomp) {
        const float vx = m.x - p.campos[0], vy = m.y - p.campos[1], vz = m.z - p.campos[2];
        const float s2 = vx * vx + vy * vy + vz * vz;
        const float len = sqrtf(s2);
        const float dxn = vx / len, dyn = vy / len, dzn = vz / len;
        float B[16], dB[16][3];
        sh_basis(p.deg, dxn, dyn, dzn, B);
        sh_basis_grad(p.deg, dxn, dyn, dzn, dB);
        const uint8_t cl = g.clamped[idx];
        const float dRGB[3] = {(cl & 1) ? 0.f : dL_dcolors[3 * idx], (cl & 2) ? 0.f : dL_dcolors[3 * idx + 1], (cl & 4) ? 0.f : dL_dcolors[3 * idx + 2]};
        const float* sh = shs + (size_t)idx * p.M * 3;
        float* dsh = dL_dsh + (size_t)idx * p.M * 3;
        const int nc = sh_ncoef(p.deg);
        float dd[3] = {0.f, 0.f, 0.f};
        for (int k = 0; k < nc; k++) {
#pragma unroll
            for (int ch = 0; ch < 3; ch++) {
                dsh[3 * k + ch] = B[k] * dRGB[ch];
                const float w = sh[3 * k + ch] * dRGB[ch];
                dd[0] += dB[k][0] * w; dd[1] += dB[k][1] * w; dd[2] += dB[k][2] * w;
            }
        }
        for (int k = 3 * nc; k < 3 * p.M; k++) dsh[k] = 0.f;   // coefficients above the active degree
        const float inv32 = 1.f / sqrtf(s2 * s2 * s2);
        dmean[0] += ((s2 - vx * vx) * dd[0] - vy * vx * dd[1] - vz * vx * dd[2]) * inv32;
        dmean[1] += (-vx * vy * dd[0] + (s2 - vy * vy) * dd[1] - vz * vy * dd[2]) * inv32;
        dmean[2] += (-vx * vz * dd[0] - vy * vz * dd[1] + (s2 - vz * vz) * dd[2]) * inv32;
    }
    dL_dmeans3D[3 * idx] = dmean[0]; dL_dmeans3D[3 * idx + 1] = dmean[1]; dL_dmeans3D[3 * idx + 2] = dmean[2];

    // cov3D -> scale, rotation (exact derivative; d/dscale carries scale_modifier)
    if (!cov3D_precomp) {
        float R[3][3];
        quat_to_R(q, R);
        const float s[3] = {p.scale_modifier * sc.x, p.scale_modifier * sc.y, p.scale_modifier * sc.z};
        const float Gm[3][3] = {{dcov[0], 0.5f * dcov[1], 0.5f * dcov[2]}, {0.5f * dcov[1], dcov[3], 0.5f * dcov[4]}, {0.5f * dcov[2], 0.5f * dcov[4], dcov[5]}};
        float dM[3][3], dR[3][3];
#pragma unroll
        for (int i = 0; i < 3; i++)
#pragma unroll
            for (int k = 0; k < 3; k++) dM[i][k] = 2.f * (Gm[i][0] * R[0][k] + Gm[i][1] * R[1][k] + Gm[i][2] * R[2][k]) * s[k];
#pragma unroll
        for (int k = 0; k < 3; k++) {
            dL_dscales[3 * idx + k] = p.scale_modifier * (dM[0][k] * R[0][k] + dM[1][k] * R[1][k] + dM[2][k] * R[2][k]);
#pragma unroll
            for (int i = 0; i < 3; i++) dR[i][k] = dM[i][k] * s[k];
        }
        const float r = q.x, x = q.y, y = q.z, z = q.w;
        float4 dq;
        dq.x = 2.f * (-z * dR[0][1] + y * dR[0][2] + z * dR[1][0] - x * dR[1][2] - y * dR[2][0] + x * dR[2][1]);
        dq.y = 2.f * (y * dR[0][1] + z * dR[0][2] + y * dR[1][0] - 2.f * x * dR[1][1] - r * dR[1][2] + z * dR[2][0] + r * dR[2][1] - 2.f * x * dR[2][2]);
        dq.z = 2.f * (-2.f * y * dR[0][0] + x * dR[0][1] + r * dR[0][2] + x * dR[1][0] + z * dR[1][2] - r * dR[2][0] + z * dR[2][1] - 2.f * y * dR[2][2]);
        dq.w = 2.f * (-2.f * z * dR[0][0] - r * dR[0][1] + x * dR[0][2] + r * dR[1][0] - 2.f * z * dR[1][1] + y * dR[1][2] + x * dR[2][0] + y * dR[2][1]);
        *reinterpret_cast<float4*>(dL_drots + 4 * idx) = dq;
    }
}

int gs_launch_preprocess_bwd(const GsParams& p, const GsGeom& g, const int* radii, const float* means3D, const float* shs,
                             const float* colors_precomp, const float* scales, const float* rotations, const float* cov3D_precomp,
                             const float* dL_dmean2D, const float* dL_dconic, const float* dL_dcolors, const float* dL_ddepths,
                             float* dL_dmeans3D, float* dL_dcov3D, float* dL_dsh, float* dL_dscales, float* dL_drots, hipStream_t s) {
    if (p.N == 0) return 0;
    hipLaunchKernelGGL(k_preprocess_bwd, dim3(c3d_cdiv(p.N, 256)), dim3(256), 0, s, p, g, radii, means3D, shs, colors_precomp, scales,
                       rotations, cov3D_precomp, dL_dmean2D, dL_dconic, dL_dcolors, dL_ddepths, dL_dmeans3D, dL_dcov3D, dL_dsh,
                       dL_dscales, dL_drots);
    C3D_LAUNCH_CHECK();
    return 0;
}
