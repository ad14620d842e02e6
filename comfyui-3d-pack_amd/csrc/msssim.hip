// msssim.hip -- multi-scale SSIM value + gradient (include/c3d_loss.h) for the trainers' loss term
//   loss += lambda * (1 - MS_SSIM(refs, imgs))        main_3DGS.py:192, diff_mesh.py:123 (reference)
// One workgroup = one 32 x 16 tile of one (image, channel) plane; the five 11-tap separable Gaussian filters of a level (x, y, x^2, y^2, xy)
// run through LDS: a 42 x 26 input tile, a horizontal pass into a 5 x 26 x 32 intermediate (four outputs per item), a vertical pass of two
// outputs per lane.
// Backward: the three per-pixel derivative maps the forward pass leaves behind (d map / d mu_y, d map / d E[y^2], d map / d E[xy]) are
// filtered with the transposed (= same, the window is symmetric) filters, zero padded, and combined with x, y and the level's scalar
// dL/d(mean map); the 2 x 2 average pooling between levels is chained by reading the parent level's finished gradient.
// No atomics: tile sums are reduced in a fixed order, so value and gradient are bit-reproducible.
#include "../../include/c3d_loss.h"
#include "c3d_common.h"
#include <math.h>

#define MS_LEVELS 5
// Tile: 32 wide x 16 tall.  A 32 x 32 tile needs 42 KB of LDS (three workgroups per CU): its phases -- global loads, horizontal pass, vertical
// pass, each behind a barrier -- then run almost unoverlapped and the level-0 kernels sat at twice their VALU time.  The half-height tile takes
// 26 KB (six workgroups per CU) for 24 % more halo loads and the same filter arithmetic per output (the horizontal pass is one full-width sweep
// of 26 x 8 items instead of two of 256 + 80).
#define MS_T 32                         // tile width
#define MS_TY 16                        // tile height
#define MS_R 5
#define MS_IN (MS_T + 2 * MS_R)         // input columns of a tile
#define MS_INY (MS_TY + 2 * MS_R)       // input rows
#define MS_VO (MS_TY / 8)               // vertically adjacent outputs per lane in the vertical pass (256 lanes = 32 columns x 8 row groups)
// workgroups per CU the filter kernels are compiled for: the prefetch registers of the next tile (round 4) bring them to ~125 VGPRs = 4 workgroups per CU (6 by LDS);
// capped at 5 or 6 they spill 76-190 bytes per lane
#ifndef MS_MIN_BLOCKS
#define MS_MIN_BLOCKS 4
#endif
#define MS_C1 (0.01f * 0.01f)
#define MS_C2 (0.03f * 0.03f)

typedef float v2f __attribute__((ext_vector_type(2)));
// w * a + b on two lanes of a register pair: v_pk_fma_f32 with the scalar weight broadcast
__device__ __forceinline__ v2f ms_fma2(float w, v2f a, v2f b) { return __builtin_elementwise_fma(v2f{w, w}, a, b); }
struct MsWin { float w[2 * MS_R + 1]; };
struct MsLevel { int H, W, Hv, Wv, tx, ty, Wp; };      // image size, valid (filtered) size, tiles over the valid size; Wp: row pitch of the derivative maps, a multiple of the tile
                                                       // width -- a tile's row of a map is then ONE whole 128-byte line (with pitch = Wv = W - 10 every such row straddled two or three
                                                       // lines: level 0 of an 8-view step wrote 977 MB for 700 MB of maps + pooled planes, profiles/r06z_pmc_traffic.csv)

// Level-0 tensors of a multi-image call whose images are NOT one contiguous [B, C, H, W] block (the fused 3DGS training step: every view has its own target, its
// own rendered image inside its workspace slice, its own mask and gradient plane): per-image base pointers in the kernel arguments.  n = 0: contiguous tensors.
#define MS_TAB_MAX 16
struct MsTab { int n; const float* x[MS_TAB_MAX]; const float* y[MS_TAB_MAX]; const float* mask[MS_TAB_MAX]; float* dy[MS_TAB_MAX]; };
// the level-0 kernels index X / Y / out with plane * HW and the mask with (plane / C) * HW: rebase the pointers of this plane's image so that the same indexing lands in it
struct MsBase { const float* x; const float* y; const float* mask; float* out; };
__device__ __forceinline__ MsBase ms_rebase(const MsTab& tab, int plane, int C, size_t HW, const float* X, const float* Y, const float* mask, float* out) {
    if (!tab.n) return MsBase{X, Y, mask, out};
    const int img = plane / C, ch = plane - img * C;
    const ptrdiff_t shift = ((ptrdiff_t)ch - (ptrdiff_t)plane) * (ptrdiff_t)HW;
    return MsBase{tab.x[img] + shift, tab.y[img] + shift, tab.mask[img] ? tab.mask[img] - (ptrdiff_t)img * (ptrdiff_t)HW : nullptr, tab.dy[img] + shift};
}

// level-0 input transform: x * mask, clamp(y) * mask
template <bool L0>
__device__ __forceinline__ void ms_load(const float* __restrict__ X, const float* __restrict__ Y, const float* __restrict__ mask, int clamp_y, int C,
                                        int plane, size_t HW, size_t off, float& x, float& y) {
    x = X[(size_t)plane * HW + off];
    y = Y[(size_t)plane * HW + off];
    if (L0) {
        if (clamp_y) y = fminf(fmaxf(y, 0.f), 1.f);
        if (mask) { const float m = mask[(size_t)(plane / C) * HW + off]; x *= m; y *= m; }
    }
}

// 2 x 2 average pooling with padding = size % 2 on both sides, zeros counted (torch avg_pool2d defaults): level l -> l + 1, x and y together
template <bool L0>
__global__ void __launch_bounds__(256) k_ms_pool(const float* __restrict__ X, const float* __restrict__ Y, const float* __restrict__ mask, int clamp_y, int C,
                                                 int H, int W, int H2, int W2, float* __restrict__ X2, float* __restrict__ Y2, MsTab tab) {
    const int plane = blockIdx.z;
    if (L0) { const MsBase b = ms_rebase(tab, plane, C, (size_t)H * W, X, Y, mask, nullptr); X = b.x; Y = b.y; mask = b.mask; }
    const int j = blockIdx.x * 64 + (threadIdx.x & 63), i = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (i >= H2 || j >= W2) return;
    const int py = H & 1, px = W & 1;
    float sx = 0.f, sy = 0.f;
#pragma unroll
    for (int a = 0; a < 2; a++)
#pragma unroll
        for (int b = 0; b < 2; b++) {
            const int r = 2 * i - py + a, c = 2 * j - px + b;
            if (r >= 0 && r < H && c >= 0 && c < W) {
                float x, y;
                ms_load<L0>(X, Y, mask, clamp_y, C, plane, (size_t)H * W, (size_t)r * W + c, x, y);
                sx += x; sy += y;
            }
        }
    X2[((size_t)plane * H2 + i) * W2 + j] = 0.25f * sx;
    Y2[((size_t)plane * H2 + i) * W2 + j] = 0.25f * sy;
}

// forward of one level: derivative maps + the tile's sum of cs (levels 0-3) or ssim (last level)
// Round 4: a workgroup walks NT horizontally adjacent tiles and issues tile t + 1's global loads (raw values into registers) BEFORE it filters tile t.  With one tile
// per workgroup the three phases of a tile -- global loads, horizontal pass, vertical pass -- sit behind barriers and their times ADD (level 0 at 1080p, 8 views:
// 0.62 ms for ~0.3 ms of HBM time, ~0.15 of VALU, ~0.15 of LDS; profiles/r04n_train_kernel_stats.csv): every wave of a CU waits out the same HBM round trip.
// Same arithmetic per output, tile by tile: results are bit-identical to the one-tile form.
template <bool LAST, bool L0>
__global__ void __launch_bounds__(256, MS_MIN_BLOCKS) k_ms_fwd(const float* __restrict__ X, const float* __restrict__ Y, const float* __restrict__ mask, int clamp_y, int C,
                                                MsLevel lv, MsWin win, float* __restrict__ mapA, float* __restrict__ mapB, float* __restrict__ mapC,
                                                float* __restrict__ partial, MsTab tab, int NT, float* __restrict__ X2, float* __restrict__ Y2, int H2, int W2) {
    // Quantities travel in pairs -- (x, y), (x^2, y^2) -- so that one v_pk_fma_f32 filters two of them (the kernel is VALU bound: 3 instructions
    // per tap and output instead of the 7 of the scalar form with its products inside the loop); xy goes alone.
    __shared__ v2f sxy[MS_INY][MS_IN + 1];
    __shared__ v2f h01[MS_INY][MS_T + 1];      // horizontally filtered (x, y)
    __shared__ v2f h23[MS_INY][MS_T + 1];      //                       (x^2, y^2)
    __shared__ float h4[MS_INY][MS_T + 1];     //                       xy
    __shared__ float red[4];
    const int plane = blockIdx.z, oy = blockIdx.y * MS_TY;
    const int t0 = blockIdx.x * NT, t1 = min(t0 + NT, lv.tx);
    const size_t HW = (size_t)lv.H * lv.W;
    if (L0) { const MsBase b = ms_rebase(tab, plane, C, HW, X, Y, mask, nullptr); X = b.x; Y = b.y; mask = b.mask; }
    constexpr int NL = (MS_INY * MS_IN + 255) / 256;
    float xr[NL], yr[NL], mr[NL];
    // all of a lane's loads are issued together, raw; the level-0 transform (clamp, mask) is applied when the values move on into LDS
    auto fetch = [&](int t) {
        const int ox = t * MS_T;
#pragma unroll
        for (int i = 0; i < NL; i++) {
            const int e = threadIdx.x + 256 * i;
            const int r = e / MS_IN, c = e - r * MS_IN;
            const int iy = oy + r, ix = ox + c;
            xr[i] = 0.f; yr[i] = 0.f; mr[i] = 1.f;
            if (e < MS_INY * MS_IN && iy < lv.H && ix < lv.W) {
                const size_t off = (size_t)iy * lv.W + ix;
                xr[i] = X[(size_t)plane * HW + off];
                yr[i] = Y[(size_t)plane * HW + off];
                if (L0 && mask) mr[i] = mask[(size_t)(plane / C) * HW + off];
            }
        }
    };
    fetch(t0);
    for (int t = t0; t < t1; t++) {
        const int ox = t * MS_T;
#pragma unroll
        for (int i = 0; i < NL; i++) {
            const int e = threadIdx.x + 256 * i;
            if (e < MS_INY * MS_IN) {
                const int r = e / MS_IN, c = e - r * MS_IN;
                float x = xr[i], y = yr[i];
                if (L0) {
                    if (clamp_y) y = fminf(fmaxf(y, 0.f), 1.f);
                    if (mask) { x *= mr[i]; y *= mr[i]; }
                }
                sxy[r][c] = v2f{x, y};
            }
        }
        __syncthreads();                       // also: every lane is done with the previous tile's vertical pass (h01 / h23 / h4 may be overwritten)
        if (t + 1 < t1) fetch(t + 1);          // in flight while this tile is filtered
        // X2 != NULL (even image sides): the 2 x 2 average pooling to the next level rides here -- the tile is in LDS anyway (with the level-0 transform applied), so the
        // pooling kernel's second read of x and y (0.16 ms per 8 views at 1080p) disappears.  A tile pools its own 32 x 16 block; the last tile of a row / column also the
        // part of its halo that reaches the image's edge (tiles cover the VALID size, H - 10 x W - 10).  Same order of additions as k_ms_pool: same bits.
        if (X2) {
            const int nrow = ((int)blockIdx.y == lv.ty - 1 ? lv.H - oy : MS_TY) >> 1, ncol = (t == lv.tx - 1 ? lv.W - ox : MS_T) >> 1;
            for (int e = threadIdx.x; e < nrow * ncol; e += 256) {
                const int pr = e / ncol, pc = e - pr * ncol;
                const v2f a = sxy[2 * pr][2 * pc], b = sxy[2 * pr][2 * pc + 1], c = sxy[2 * pr + 1][2 * pc], d = sxy[2 * pr + 1][2 * pc + 1];
                float sx = 0.f, sy = 0.f;
                sx += a.x; sy += a.y; sx += b.x; sy += b.y; sx += c.x; sy += c.y; sx += d.x; sy += d.y;
                const size_t o = ((size_t)plane * H2 + (oy >> 1) + pr) * W2 + (ox >> 1) + pc;
                X2[o] = 0.25f * sx; Y2[o] = 0.25f * sy;
            }
        }
        // horizontal pass, four adjacent outputs per item: the 14 inputs they share are read once (one 8-byte LDS read each) and squared once
        for (int e = threadIdx.x; e < MS_INY * (MS_T / 4); e += 256) {
            const int r = e >> 3, c0 = (e & 7) * 4;
            v2f p[14], q[14];
            float xy[14];
#pragma unroll
            for (int k = 0; k < 14; k++) { p[k] = sxy[r][c0 + k]; q[k] = p[k] * p[k]; xy[k] = p[k].x * p[k].y; }
#pragma unroll
            for (int o = 0; o < 4; o++) {
                v2f a01 = v2f{0.f, 0.f}, a23 = v2f{0.f, 0.f};
                float a4 = 0.f;
#pragma unroll
                for (int k = 0; k < 2 * MS_R + 1; k++) {
                    const float w = win.w[k];
                    a01 = ms_fma2(w, p[o + k], a01); a23 = ms_fma2(w, q[o + k], a23); a4 = __builtin_fmaf(w, xy[o + k], a4);
                }
                h01[r][c0 + o] = a01; h23[r][c0 + o] = a23; h4[r][c0 + o] = a4;
            }
        }
        __syncthreads();                       // also: every lane is done reading sxy (the next tile's values may move in)
        // vertical pass, MS_VO vertically adjacent outputs per lane: 10 + MS_VO rows of the intermediate per quantity
        const int c = threadIdx.x & 31, r0 = (threadIdx.x >> 5) * MS_VO;
        float acc[MS_VO][5];
        {
            v2f c01[2 * MS_R + MS_VO], c23[2 * MS_R + MS_VO];
            float c4[2 * MS_R + MS_VO];
#pragma unroll
            for (int k = 0; k < 2 * MS_R + MS_VO; k++) { c01[k] = h01[r0 + k][c]; c23[k] = h23[r0 + k][c]; c4[k] = h4[r0 + k][c]; }
#pragma unroll
            for (int o = 0; o < MS_VO; o++) {
                v2f a01 = v2f{0.f, 0.f}, a23 = v2f{0.f, 0.f};
                float a4 = 0.f;
#pragma unroll
                for (int k = 0; k < 2 * MS_R + 1; k++) {
                    const float w = win.w[k];
                    a01 = ms_fma2(w, c01[o + k], a01); a23 = ms_fma2(w, c23[o + k], a23); a4 = __builtin_fmaf(w, c4[o + k], a4);
                }
                acc[o][0] = a01.x; acc[o][1] = a01.y; acc[o][2] = a23.x; acc[o][3] = a23.y; acc[o][4] = a4;
            }
        }
        float local = 0.f;
#pragma unroll
        for (int j = 0; j < MS_VO; j++) {
            const int r = r0 + j;
            const float m1 = acc[j][0], m2 = acc[j][1], e11 = acc[j][2], e22 = acc[j][3], e12 = acc[j][4];
            const int vy = oy + r, vx = ox + c;
            if (vy < lv.Hv && vx < lv.Wv) {
                const float s1 = e11 - m1 * m1, s2 = e22 - m2 * m2, s12 = e12 - m1 * m2;
                const float dcs = 1.f / (s1 + s2 + MS_C2);
                const float cs = (2.f * s12 + MS_C2) * dcs;
                // d cs / d E[xy] = 2 / D,  d cs / d E[y^2] = -cs / D,  d cs / d mu_y = (2 cs mu_y - 2 mu_x) / D
                float dA = (2.f * cs * m2 - 2.f * m1) * dcs, dB = -cs * dcs, dC = 2.f * dcs, val = cs;
                if (LAST) {
                    const float dl = 1.f / (m1 * m1 + m2 * m2 + MS_C1);
                    const float l = (2.f * m1 * m2 + MS_C1) * dl;
                    dA = l * dA + cs * (2.f * m1 - 2.f * l * m2) * dl;      // ssim = l * cs
                    dB *= l; dC *= l; val = l * cs;
                }
                const size_t o = ((size_t)plane * lv.Hv + vy) * lv.Wp + vx;
                mapA[o] = dA; mapB[o] = dB; mapC[o] = dC;
                local += val;
            }
        }
        // fixed-order block sum
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) local += __shfl_xor(local, o, 64);
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = local;
        __syncthreads();
        if (threadIdx.x == 0) partial[((size_t)plane * lv.ty + blockIdx.y) * lv.tx + t] = (red[0] + red[1]) + (red[2] + red[3]);
        // (red is rewritten two barriers further on: lane 0 has read it by then)
    }
}

// per plane: level means (fixed-order sums of the tile partials) -> ms, and the scalar dL/d(map pixel) of every level
struct MsFinal { const float* partial[MS_LEVELS]; int tiles[MS_LEVELS]; float inv_npix[MS_LEVELS]; float wts[MS_LEVELS]; };
// pmean: planes the value is a mean over (P for one mean over the whole batch; C when every image of the batch is its own loss term)
// img_C > 0 (every image its own loss term, the fused 3DGS training steps): a workgroup per IMAGE, 256 lanes per plane (img_C <= 4 planes side by side; more: one after the
// other), also leaves the image's value, *(out0 + image * stride bytes) = a + b * mean of its planes' ms -- what k_ms_mean_images did in a launch of its own.  Per plane the
// same 256-lane strided sums and the same tree as the plane-per-workgroup form: the same bits.
__global__ void __launch_bounds__(1024) k_ms_finalize(MsFinal f, int P, int pmean, float grad_scale, float* __restrict__ g /* [levels][P] */, float* __restrict__ ms_plane,
                                                       int img_C, float a, float b, float* __restrict__ out0, size_t stride) {
    __shared__ float red[4][256];
    __shared__ float s_ms[4];
    const int sub = threadIdx.x >> 8, tid = threadIdx.x & 255, side = blockDim.x >> 8;      // `side` planes at a time
    float img_sum = 0.f;
    const int planes = img_C > 0 ? img_C : 1;
  for (int p0 = 0; p0 < planes; p0 += side) {
    const int pc = p0 + sub;
    const bool live = pc < planes;
    const int plane = img_C > 0 ? (int)blockIdx.x * img_C + (live ? pc : 0) : (int)blockIdx.x;
    float v[MS_LEVELS];
    for (int l = 0; l < MS_LEVELS; l++) {
        float s = 0.f;
        for (int t = tid; t < f.tiles[l]; t += 256) s += f.partial[l][(size_t)plane * f.tiles[l] + t];
        red[sub][tid] = s;
        __syncthreads();
        for (int o = 128; o >= 1; o >>= 1) {
            if (tid < o) red[sub][tid] += red[sub][tid + o];
            __syncthreads();
        }
        { const float mean = red[sub][0] * f.inv_npix[l]; v[l] = mean != mean ? mean : fmaxf(mean, 0.f); }      // relu(mean); a NaN mean stays NaN (a non-finite render must show in the loss)
        __syncthreads();
    }
    if (tid == 0 && live) {
        float ms = 1.f;
        bool pos = true, bad = false;
        for (int l = 0; l < MS_LEVELS; l++) { bad = bad || v[l] != v[l]; pos = pos && v[l] > 0.f; ms *= powf(v[l], f.wts[l]); }
        if (!pos) ms = 0.f;
        if (bad) ms = __builtin_nanf("");
        ms_plane[plane] = ms;
        s_ms[sub] = ms;
        for (int l = 0; l < MS_LEVELS; l++) g[(size_t)l * P + plane] = bad ? ms : (pos ? grad_scale * (f.wts[l] * ms / v[l]) * f.inv_npix[l] / (float)pmean : 0.f);
    }
    __syncthreads();
    if (threadIdx.x == 0)
        for (int q = 0; q < side && p0 + q < planes; q++) img_sum += s_ms[q];      // planes in order
    __syncthreads();
  }
    if (img_C > 0 && out0 && threadIdx.x == 0) *(float*)((char*)out0 + (size_t)blockIdx.x * stride) = a + b * (img_sum / (float)img_C);
}
// value: out (+)= a + b * mean(ms), a fixed-order mean.  store = 1: the word is the caller's own slot (the fused training steps keep one slot per view and add the
// views up in view order: the loss VALUE is then bit-reproducible like the gradients); store = 0: added atomically to a word other launches may add to as well.
__global__ void k_ms_mean(const float* __restrict__ ms_plane, int P, float a, float b, float* __restrict__ out, int store) {
    float s = 0.f;
    for (int p = 0; p < P; p++) s += ms_plane[p];
    const float val = a + b * (s / (float)P);
    if (store) *out = val; else atomicAdd(out, val);
}
// backward of one level: gradient w.r.t. the level's y over the whole image (+ the pooled parent level's gradient)
// Round 4: NT horizontally adjacent tiles per workgroup; tile t + 1's maps and tile t's epilogue inputs (x, y, mask, parent gradient, the old output when
// accumulating) are requested before tile t is filtered (see k_ms_fwd).  Same arithmetic per output: bit-identical to the one-tile form.
template <bool L0>
__global__ void __launch_bounds__(256, MS_MIN_BLOCKS) k_ms_bwd(const float* __restrict__ X, const float* __restrict__ Y, const float* __restrict__ mask, int clamp_y, int C,
                                                MsLevel lv, MsWin win, const float* __restrict__ mapA, const float* __restrict__ mapB, const float* __restrict__ mapC,
                                                const float* __restrict__ g, const float* __restrict__ parent, int H2, int W2, float* __restrict__ out, int accumulate, MsTab tab, int NT) {
    __shared__ v2f smab[MS_INY][MS_IN + 1];    // maps A, B as a pair (one packed FMA filters both), C alone
    __shared__ float smc[MS_INY][MS_IN + 1];
    __shared__ v2f hab[MS_INY][MS_T + 1];
    __shared__ float hc[MS_INY][MS_T + 1];
    const int plane = blockIdx.z, oy = blockIdx.y * MS_TY;
    const int ntx = (lv.W + MS_T - 1) / MS_T;
    const int t0 = blockIdx.x * NT, t1 = min(t0 + NT, ntx);
    const float gl = g[plane];
    const size_t HW = (size_t)lv.H * lv.W;
    if (L0) { const MsBase b = ms_rebase(tab, plane, C, HW, X, Y, mask, out); X = b.x; Y = b.y; mask = b.mask; out = b.out; }
    constexpr int NL = (MS_INY * MS_IN + 255) / 256;
    float a_[NL], b_[NL], c_[NL];
    // d in(q) = sum_k w[k] M(q - 10 + k): LDS row 0 <-> map row oy - 10
    auto fetch = [&](int t) {
        const int ox = t * MS_T;
#pragma unroll
        for (int i = 0; i < NL; i++) {
            const int e = threadIdx.x + 256 * i;
            const int r = e / MS_IN, c = e - r * MS_IN;
            const int uy = oy - 2 * MS_R + r, ux = ox - 2 * MS_R + c;
            a_[i] = 0.f; b_[i] = 0.f; c_[i] = 0.f;
            if (e < MS_INY * MS_IN && uy >= 0 && uy < lv.Hv && ux >= 0 && ux < lv.Wv) {
                const size_t o = ((size_t)plane * lv.Hv + uy) * lv.Wp + ux;
                a_[i] = mapA[o]; b_[i] = mapB[o]; c_[i] = mapC[o];
            }
        }
    };
    const int c = threadIdx.x & 31, r0 = (threadIdx.x >> 5) * MS_VO;     // this lane's outputs: column c, rows r0 .. r0 + MS_VO - 1 of every tile
    fetch(t0);
    for (int t = t0; t < t1; t++) {
        const int ox = t * MS_T;
#pragma unroll
        for (int i = 0; i < NL; i++) {
            const int e = threadIdx.x + 256 * i;
            if (e < MS_INY * MS_IN) { const int r = e / MS_IN, cc = e - r * MS_IN; smab[r][cc] = v2f{a_[i], b_[i]}; smc[r][cc] = c_[i]; }
        }
        __syncthreads();                       // also: every lane is done with the previous tile's vertical pass
        if (t + 1 < t1) fetch(t + 1);
        // this tile's epilogue inputs, requested now, used after the two filter passes
        float ex[MS_VO], ey[MS_VO], em[MS_VO], ep[MS_VO], eo[MS_VO];
#pragma unroll
        for (int j = 0; j < MS_VO; j++) {
            const int qy = oy + r0 + j, qx = ox + c;
            ex[j] = 0.f; ey[j] = 0.f; em[j] = 1.f; ep[j] = 0.f; eo[j] = 0.f;
            if (qy < lv.H && qx < lv.W) {
                const size_t off = (size_t)qy * lv.W + qx;
                ex[j] = X[(size_t)plane * HW + off];
                ey[j] = Y[(size_t)plane * HW + off];
                if (L0 && mask) em[j] = mask[(size_t)(plane / C) * HW + off];
                if (parent) ep[j] = parent[((size_t)plane * H2 + ((qy + (lv.H & 1)) >> 1)) * W2 + ((qx + (lv.W & 1)) >> 1)];
                if (accumulate) eo[j] = out[(size_t)plane * HW + off];
            }
        }
        for (int e = threadIdx.x; e < MS_INY * (MS_T / 4); e += 256) {     // horizontal, four adjacent outputs per item
            const int r = e >> 3, c0 = (e & 7) * 4;
            v2f vab[14];
            float vc[14];
#pragma unroll
            for (int k = 0; k < 14; k++) { vab[k] = smab[r][c0 + k]; vc[k] = smc[r][c0 + k]; }
#pragma unroll
            for (int o = 0; o < 4; o++) {
                v2f fab = v2f{0.f, 0.f};
                float fc = 0.f;
#pragma unroll
                for (int k = 0; k < 2 * MS_R + 1; k++) { fab = ms_fma2(win.w[k], vab[o + k], fab); fc = __builtin_fmaf(win.w[k], vc[o + k], fc); }
                hab[r][c0 + o] = fab; hc[r][c0 + o] = fc;
            }
        }
        __syncthreads();                       // also: every lane is done reading smab / smc
        float acc[MS_VO][3];                   // vertical, MS_VO vertically adjacent outputs per lane
        {
            v2f cab[2 * MS_R + MS_VO];
            float cc[2 * MS_R + MS_VO];
#pragma unroll
            for (int k = 0; k < 2 * MS_R + MS_VO; k++) { cab[k] = hab[r0 + k][c]; cc[k] = hc[r0 + k][c]; }
#pragma unroll
            for (int o = 0; o < MS_VO; o++) {
                v2f fab = v2f{0.f, 0.f};
                float fc = 0.f;
#pragma unroll
                for (int k = 0; k < 2 * MS_R + 1; k++) { fab = ms_fma2(win.w[k], cab[o + k], fab); fc = __builtin_fmaf(win.w[k], cc[o + k], fc); }
                acc[o][0] = fab.x; acc[o][1] = fab.y; acc[o][2] = fc;
            }
        }
#pragma unroll
        for (int j = 0; j < MS_VO; j++) {
            const int qy = oy + r0 + j, qx = ox + c;
            if (qy >= lv.H || qx >= lv.W) continue;
            const float fa = acc[j][0], fb = acc[j][1], fc = acc[j][2];
            float x = ex[j], y = ey[j];
            if (L0) {      // level-0 input transform: x * mask, clamp(y) * mask
                if (clamp_y) y = fminf(fmaxf(y, 0.f), 1.f);
                if (mask) { x *= em[j]; y *= em[j]; }
            }
            float gr = gl * (fa + 2.f * y * fb + x * fc);
            if (parent) gr += 0.25f * ep[j];
            if (L0) {   // chain through y_eff = clamp(y) * mask
                if (mask) gr *= em[j];
                if (clamp_y) { const float yr = ey[j]; if (!(yr >= 0.f && yr <= 1.f)) gr = 0.f; }
            }
            out[(size_t)plane * HW + (size_t)qy * lv.W + qx] = accumulate ? eo[j] + gr : gr;
        }
    }
}

namespace {
struct MsPlan {
    MsLevel lv[MS_LEVELS];
    size_t off_x[MS_LEVELS], off_y[MS_LEVELS], off_map[MS_LEVELS], off_grad[MS_LEVELS], off_part[MS_LEVELS], off_g, off_ms, bytes;
};
void ms_plan(int P, int H, int W, MsPlan& pl) {
    size_t off = 0;
    auto take = [&](size_t b) { size_t o = off; off += c3d_align(b); return o; };
    int h = H, w = W;
    for (int l = 0; l < MS_LEVELS; l++) {
        MsLevel& L = pl.lv[l];
        L.H = h; L.W = w; L.Hv = h - 2 * MS_R; L.Wv = w - 2 * MS_R;
        L.tx = (L.Wv + MS_T - 1) / MS_T; L.ty = (L.Hv + MS_TY - 1) / MS_TY;
#ifdef C3D_MS_NO_PITCH      // (A/B switch of profiles/r06/r06y_*)
        L.Wp = L.Wv > 0 ? L.Wv : 0;
#else
        L.Wp = L.Wv > 0 ? L.tx * MS_T : 0;
#endif
        const size_t img = sizeof(float) * (size_t)P * h * w, val = sizeof(float) * (size_t)P * (size_t)(L.Hv > 0 ? L.Hv : 0) * (size_t)L.Wp;
        pl.off_x[l] = l ? take(img) : 0; pl.off_y[l] = l ? take(img) : 0;
        pl.off_map[l] = take(3 * val);
        pl.off_grad[l] = l ? take(img) : 0;
        pl.off_part[l] = take(sizeof(float) * (size_t)P * (size_t)(L.tx > 0 ? L.tx : 1) * (size_t)(L.ty > 0 ? L.ty : 1));
        const int py = h & 1, px = w & 1;
        h = (h + 2 * py - 2) / 2 + 1; w = (w + 2 * px - 2) / 2 + 1;
    }
    pl.off_g = take(sizeof(float) * MS_LEVELS * (size_t)P);
    pl.off_ms = take(sizeof(float) * (size_t)P);
    pl.bytes = off;
}
// tiles a workgroup walks (k_ms_fwd / k_ms_bwd): four while the launch still has several workgroups per residency slot (256 CUs x 6 workgroups), else fewer
#ifndef MS_TILES_PER_GROUP
#define MS_TILES_PER_GROUP 4
#endif
int ms_tiles_per_group(int tx, int ty, int planes) {
    const long long tiles = (long long)tx * ty * planes;
    int nt = MS_TILES_PER_GROUP;
    while (nt > 1 && tiles < (long long)nt * 2 * 1536) nt >>= 1;
    return nt;
}
MsWin ms_window() {
    MsWin w;
    double g[2 * MS_R + 1], s = 0;
    for (int k = 0; k <= 2 * MS_R; k++) { const double x = k - MS_R; g[k] = exp(-(x * x) / (2.0 * 1.5 * 1.5)); s += g[k]; }
    for (int k = 0; k <= 2 * MS_R; k++) w.w[k] = (float)(g[k] / s);
    return w;
}
}  // namespace

int ms_value_grad(const float* x, const float* y, const float* mask, int clamp_y, int B, int C, int H, int W, float grad_scale, int accumulate, float* dL_dy,
                  float va, float vb, float* ms_out, void* workspace, hipStream_t s, int store_value = 0);
static int ms_run(const float* x, const float* y, const float* mask, const MsTab& tab, int clamp_y, int B, int C, int H, int W, float grad_scale, int accumulate, float* dL_dy,
                  float va, float vb, float* ms_out, size_t out_stride, bool per_image, void* workspace, hipStream_t s, int store_value);

extern "C" {

size_t c3d_msssim_workspace_bytes(int32_t B, int32_t C, int32_t H, int32_t W) {
    MsPlan pl;
    ms_plan(B * C, H, W, pl);
    return pl.bytes;
}

int c3d_msssim_value_grad(const float* x, const float* y, const float* mask, int32_t clamp_y, int32_t B, int32_t C, int32_t H, int32_t W,
                          float grad_scale, int32_t accumulate, float* dL_dy, float* ms_out, void* workspace, c3d_stream_t stream) {
    return ms_value_grad(x, y, mask, clamp_y, B, C, H, W, grad_scale, accumulate, dL_dy, 0.f, 1.f, ms_out, workspace, (hipStream_t)stream);
}

}  // extern "C"

// out_word += va + vb * mean MS-SSIM  (va = 0, vb = 1: the plain value; the fused training step passes the loss term's weights)
int ms_value_grad(const float* x, const float* y, const float* mask, int clamp_y, int B, int C, int H, int W, float grad_scale, int accumulate, float* dL_dy,
                  float va, float vb, float* ms_out, void* workspace, hipStream_t s, int store_value) {
    if (B <= 0 || C <= 0) return 0;
    if (!x || !y || !dL_dy || !workspace) { c3d_set_error("c3d_msssim_value_grad: NULL pointer"); return -1; }
    MsTab tab{};
    return ms_run(x, y, mask, tab, clamp_y, B, C, H, W, grad_scale, accumulate, dL_dy, va, vb, ms_out, 0, false, workspace, s, store_value);
}
// The MS-SSIM terms of B <= 16 images that live in separate buffers, EVERY image its own loss term (the views of a fused 3DGS training step), in one set of launches:
// image i: x[i] / y[i] / mask[i] (NULL = none) / dy[i] are [C, H, W] planes ([1, H, W] for the mask);  *(out0 + i * out_stride bytes) = va + vb * MS-SSIM(image i),
// dy[i] (+)= grad_scale * d MS-SSIM(image i) / dy.  Same values, bit for bit, as B calls of ms_value_grad with B = 1 (planes never mix; the per-image mean is over C).
int ms_value_grad_images(const float* const* x, const float* const* y, const float* const* mask, float* const* dy, int clamp_y, int B, int C, int H, int W, float grad_scale,
                         int accumulate, float va, float vb, float* out0, size_t out_stride, void* workspace, hipStream_t s) {
    if (B <= 0 || C <= 0) return 0;
    if (B > MS_TAB_MAX) { c3d_set_error("ms_value_grad_images: at most %d images per call", MS_TAB_MAX); return -1; }
    if (!x || !y || !dy || !workspace) { c3d_set_error("ms_value_grad_images: NULL pointer"); return -1; }
    MsTab tab{};
    tab.n = B;
    for (int i = 0; i < B; i++) {
        if (!x[i] || !y[i] || !dy[i]) { c3d_set_error("ms_value_grad_images: NULL image %d", i); return -1; }
        tab.x[i] = x[i]; tab.y[i] = y[i]; tab.mask[i] = mask ? mask[i] : nullptr; tab.dy[i] = dy[i];
    }
    return ms_run(nullptr, nullptr, nullptr, tab, clamp_y, B, C, H, W, grad_scale, accumulate, nullptr, va, vb, out0, out_stride, true, workspace, s, 1);
}
static int ms_run(const float* x, const float* y, const float* mask, const MsTab& tab, int clamp_y, int B, int C, int H, int W, float grad_scale, int accumulate, float* dL_dy,
                  float va, float vb, float* ms_out, size_t out_stride, bool per_image, void* workspace, hipStream_t s, int store_value) {
    if ((H < W ? H : W) <= (2 * MS_R) * (1 << (MS_LEVELS - 1))) { c3d_set_error("c3d_msssim_value_grad: image sides must exceed %d for %d scales", (2 * MS_R) << (MS_LEVELS - 1), MS_LEVELS); return -1; }
    const int P = B * C;
    MsPlan pl;
    ms_plan(P, H, W, pl);
    char* ws = (char*)workspace;
    const MsWin win = ms_window();
    const float* X[MS_LEVELS]; const float* Y[MS_LEVELS];
    X[0] = x; Y[0] = y;
    for (int l = 1; l < MS_LEVELS; l++) { X[l] = (const float*)(ws + pl.off_x[l]); Y[l] = (const float*)(ws + pl.off_y[l]); }
    static const float wts[MS_LEVELS] = {0.0448f, 0.2856f, 0.3001f, 0.2363f, 0.1333f};
    const MsTab none{};
    MsFinal fin;
    C3dProfScope ps(C3D_P_MSSSIM, s);
    for (int l = 0; l < MS_LEVELS; l++) {
        const MsLevel& L = pl.lv[l];
        const size_t val = (size_t)P * L.Hv * L.Wp;
        float* mA = (float*)(ws + pl.off_map[l]); float* mB = mA + val; float* mC = mB + val;
        float* part = (float*)(ws + pl.off_part[l]);
        const int nt = ms_tiles_per_group(L.tx, L.ty, P);
        const dim3 grid(c3d_cdiv(L.tx, nt), L.ty, P);
        // pooling to the next level inside the forward kernel when both sides are even (no padding row / column: a pooled pixel is a whole 2 x 2 block of one tile)
        const bool pool_here = l < MS_LEVELS - 1 && !(L.H & 1) && !(L.W & 1);
        float* X2 = pool_here ? (float*)(ws + pl.off_x[l + 1]) : nullptr; float* Y2 = pool_here ? (float*)(ws + pl.off_y[l + 1]) : nullptr;
        const int H2 = l < MS_LEVELS - 1 ? pl.lv[l + 1].H : 0, W2 = l < MS_LEVELS - 1 ? pl.lv[l + 1].W : 0;
        if (l == 0)                    hipLaunchKernelGGL((k_ms_fwd<false, true>), grid, dim3(256), 0, s, X[l], Y[l], mask, clamp_y, C, L, win, mA, mB, mC, part, tab, nt, X2, Y2, H2, W2);
        else if (l < MS_LEVELS - 1)    hipLaunchKernelGGL((k_ms_fwd<false, false>), grid, dim3(256), 0, s, X[l], Y[l], (const float*)nullptr, 0, C, L, win, mA, mB, mC, part, none, nt, X2, Y2, H2, W2);
        else                           hipLaunchKernelGGL((k_ms_fwd<true, false>), grid, dim3(256), 0, s, X[l], Y[l], (const float*)nullptr, 0, C, L, win, mA, mB, mC, part, none, nt, X2, Y2, H2, W2);
        fin.partial[l] = part; fin.tiles[l] = L.tx * L.ty; fin.inv_npix[l] = 1.f / ((float)L.Hv * (float)L.Wv); fin.wts[l] = wts[l];
        if (l < MS_LEVELS - 1 && !pool_here) {
            const MsLevel& N = pl.lv[l + 1];
            const dim3 pg(c3d_cdiv(N.W, 64), c3d_cdiv(N.H, 4), P);
            if (l == 0) hipLaunchKernelGGL((k_ms_pool<true>), pg, dim3(256), 0, s, X[l], Y[l], mask, clamp_y, C, L.H, L.W, N.H, N.W, (float*)(ws + pl.off_x[l + 1]), (float*)(ws + pl.off_y[l + 1]), tab);
            else        hipLaunchKernelGGL((k_ms_pool<false>), pg, dim3(256), 0, s, X[l], Y[l], (const float*)nullptr, 0, C, L.H, L.W, N.H, N.W, (float*)(ws + pl.off_x[l + 1]), (float*)(ws + pl.off_y[l + 1]), none);
        }
    }
    float* g = (float*)(ws + pl.off_g);
    float* msp = (float*)(ws + pl.off_ms);
    if (per_image) hipLaunchKernelGGL(k_ms_finalize, dim3(B), dim3(256 * (C < 4 ? C : 4)), 0, s, fin, P, C, grad_scale, g, msp, C, va, vb, ms_out, out_stride);      // (the images' values in the same launch)
    else hipLaunchKernelGGL(k_ms_finalize, dim3(P), dim3(256), 0, s, fin, P, P, grad_scale, g, msp, 0, 0.f, 0.f, (float*)nullptr, (size_t)0);
    if (ms_out && !per_image) hipLaunchKernelGGL(k_ms_mean, dim3(1), dim3(1), 0, s, msp, P, va, vb, ms_out, store_value);
    for (int l = MS_LEVELS - 1; l >= 0; l--) {
        const MsLevel& L = pl.lv[l];
        const size_t val = (size_t)P * L.Hv * L.Wp;
        const float* mA = (const float*)(ws + pl.off_map[l]); const float* mB = mA + val; const float* mC = mB + val;
        const float* parent = l < MS_LEVELS - 1 ? (const float*)(ws + pl.off_grad[l + 1]) : nullptr;
        const int H2 = l < MS_LEVELS - 1 ? pl.lv[l + 1].H : 0, W2 = l < MS_LEVELS - 1 ? pl.lv[l + 1].W : 0;
        const int ntx = c3d_cdiv(L.W, MS_T), nty = c3d_cdiv(L.H, MS_TY), nt = ms_tiles_per_group(ntx, nty, P);
        const dim3 grid(c3d_cdiv(ntx, nt), nty, P);
        if (l == 0) hipLaunchKernelGGL((k_ms_bwd<true>), grid, dim3(256), 0, s, X[l], Y[l], mask, clamp_y, C, L, win, mA, mB, mC, g + (size_t)l * P, parent, H2, W2, dL_dy, accumulate, tab, nt);
        else        hipLaunchKernelGGL((k_ms_bwd<false>), grid, dim3(256), 0, s, X[l], Y[l], (const float*)nullptr, 0, C, L, win, mA, mB, mC, g + (size_t)l * P, parent, H2, W2, (float*)(ws + pl.off_grad[l]), 0, none, nt);
    }
    C3D_LAUNCH_CHECK();
    return 0;
}
