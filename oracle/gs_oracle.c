/*
 * gs_oracle.c -- CPU restatement of the 3D-Gaussian-Splatting tile rasterizer
 * (forward + backward) that ComfyUI-3D-Pack calls through
 * `diff_gaussian_rasterization` (ashawkey fork: colour + depth + alpha outputs).
 *
 * THIS IS TEST INFRASTRUCTURE.  Only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg may load it.  The product path
 * (comfyui-3d-pack_amd/) never imports, links or calls anything in oracle/.
 *
 * PARITY UNPINNED: the reference tree (/root/reference) does not contain the
 * rasterizer's arithmetic (un-vendored CUDA wheel, built from un-pinned git
 * HEAD: _Pre_Builds/_Build_Scripts/dependencies.txt:2, my-reqs.txt:22) and
 * ships no tests or golden vectors for this path.  This
 * file restates the published algorithm of that dependency (the A1..A8 stages
 * of SURVEY.md section 2.3 / Appendix A) and is anchored on the reference's own
 * call site and conventions:
 *   - call site / argument meaning: MVs_Algorithms/GaussianSplatting/
 *     main_3DGS_renderer.py:845-936 (settings tuple, tensors, outputs)
 *   - camera matrices (row-vector / transposed storage):
 *     shared_utils/camera_utils.py:188-214
 *   - SH basis and sign pattern: shared_utils/sh_utils.py:26-43,57-100
 *   - quaternion (w,x,y,z) -> rotation: main_3DGS_renderer.py:84-102
 *   - covariance = L L^T, L = R*S: main_3DGS_renderer.py:104-113,220-224
 * Those four conventions ARE pinned against the reference's own Python, run in
 * the build container (tests/golden/make_golden_ref_py.py ->
 * tests/golden/ref_py_conventions.npz -> tests/test_ref_conventions.py): the
 * per-Gaussian colour, view depth, pixel position and conic this file computes
 * agree with eval_sh / MiniCam / covariance_activation of the reference.
 * Everything the CUDA kernel adds on top (culling constants, EWA projection,
 * tile binning, the blending loop, every gradient) stays unpinned and is held
 * instead by analytic known-answer cases, by an independent float64
 * torch-autograd restatement (oracle/gs_torch_ref.py) and by finite
 * differences (tests/test_gs_oracle.py).
 *
 * Build twice: -DREAL=float (arithmetic class of the product) and
 * -DREAL=double (tight truth for gradient checks).
 *
 * Ordering rule: splats are composited per 16x16 tile in ascending
 * (view depth, Gaussian index) order; the index tie-break is what a stable
 * radix sort of (tile | depth-bits) keys over index-ordered emission yields.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#ifndef REAL
#define REAL float
#endif
typedef REAL real;

#define BLOCK_X 16
#define BLOCK_Y 16

/* dL/dscale convention (see the backward pass): 0 = as the dependency returns it (default), 1 = exact derivative */
static int g_exact_dscale = 0;
int gs_oracle_set_exact_dscale(int on) { int old = g_exact_dscale; g_exact_dscale = on != 0; return old; }

static const double SH_C0 = 0.28209479177387814;
static const double SH_C1 = 0.4886025119029199;
static const double SH_C2[5] = {1.0925484305920792, -1.0925484305920792, 0.31539156525252005,
                                -1.0925484305920792, 0.5462742152960396};
static const double SH_C3[7] = {-0.5900435899266435, 2.890611442640554, -0.4570457994644658,
                                0.3731763325901154, -0.4570457994644658, 1.445305721320277,
                                -0.5900435899266435};

typedef struct {
    int N, M, deg, W, H, gx, gy;
    real tanfovx, tanfovy, focal_x, focal_y, scale_modifier;
    real bg[3], view[16], proj[16], campos[3];
    /* per-Gaussian ("geometry") state */
    real *depths, *xy, *conic_opacity, *rgb, *cov3D;
    unsigned char *clamped;
    int *radii, *tiles_touched;
    /* binning state */
    int64_t D;
    uint32_t *point_list;
    uint32_t *ranges; /* [tiles][2] */
    /* image state */
    real *final_T;
    uint32_t *n_contrib;
} gs_state;

static real rmin(real a, real b) { return a < b ? a : b; }
static real rmax(real a, real b) { return a > b ? a : b; }
static int imin(int a, int b) { return a < b ? a : b; }
static int imax(int a, int b) { return a > b ? a : b; }

/* p (row vector, w=1) times the 4x4 stored row-major as the caller passes it
 * (world_view_transform = w2c^T, camera_utils.py:205): out_i = sum_j p_j m[4j+i] + m[12+i] */
static void xform4x3(const real *p, const real *m, real *o) {
    o[0] = m[0] * p[0] + m[4] * p[1] + m[8] * p[2] + m[12];
    o[1] = m[1] * p[0] + m[5] * p[1] + m[9] * p[2] + m[13];
    o[2] = m[2] * p[0] + m[6] * p[1] + m[10] * p[2] + m[14];
}
static void xform4x4(const real *p, const real *m, real *o) {
    o[0] = m[0] * p[0] + m[4] * p[1] + m[8] * p[2] + m[12];
    o[1] = m[1] * p[0] + m[5] * p[1] + m[9] * p[2] + m[13];
    o[2] = m[2] * p[0] + m[6] * p[1] + m[10] * p[2] + m[14];
    o[3] = m[3] * p[0] + m[7] * p[1] + m[11] * p[2] + m[15];
}

/* A1: Sigma3 = R diag(s)^2 R^T, six unique entries (xx,xy,xz,yy,yz,zz).  The quaternion is
 * used as given (the caller normalises: main_3DGS_renderer.py:298-299). */
static void quat_to_R(const real *q, real R[3][3]) {
    real r = q[0], x = q[1], y = q[2], z = q[3];
    R[0][0] = (real)1 - (real)2 * (y * y + z * z);
    R[0][1] = (real)2 * (x * y - r * z);
    R[0][2] = (real)2 * (x * z + r * y);
    R[1][0] = (real)2 * (x * y + r * z);
    R[1][1] = (real)1 - (real)2 * (x * x + z * z);
    R[1][2] = (real)2 * (y * z - r * x);
    R[2][0] = (real)2 * (x * z - r * y);
    R[2][1] = (real)2 * (y * z + r * x);
    R[2][2] = (real)1 - (real)2 * (x * x + y * y);
}
static void compute_cov3D(const real *scale, real mod, const real *q, real *cov) {
    real R[3][3], Mm[3][3];
    quat_to_R(q, R);
    for (int i = 0; i < 3; i++)
        for (int k = 0; k < 3; k++) Mm[i][k] = R[i][k] * (mod * scale[k]);
    real S[3][3];
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) S[i][j] = Mm[i][0] * Mm[j][0] + Mm[i][1] * Mm[j][1] + Mm[i][2] * Mm[j][2];
    cov[0] = S[0][0]; cov[1] = S[0][1]; cov[2] = S[0][2];
    cov[3] = S[1][1]; cov[4] = S[1][2]; cov[5] = S[2][2];
}

/* A1: EWA projection.  T2 = J(2x3) * W(3x3); cov2D = T2 Sigma3 T2^T; +0.3 on the diagonal.
 * Also returns T2 and the clamp flags for the backward pass. */
static void compute_T2(const gs_state *st, const real *mean, real T2[2][3], real t[3], int *xin, int *yin) {
    xform4x3(mean, st->view, t);
    real limx = (real)1.3 * st->tanfovx, limy = (real)1.3 * st->tanfovy;
    real txtz = t[0] / t[2], tytz = t[1] / t[2];
    *xin = !(txtz < -limx || txtz > limx);
    *yin = !(tytz < -limy || tytz > limy);
    t[0] = rmin(limx, rmax(-limx, txtz)) * t[2];
    t[1] = rmin(limy, rmax(-limy, tytz)) * t[2];
    real J[2][3] = {{st->focal_x / t[2], 0, -(st->focal_x * t[0]) / (t[2] * t[2])},
                    {0, st->focal_y / t[2], -(st->focal_y * t[1]) / (t[2] * t[2])}};
    const real *v = st->view; /* W[m][k] = w2c[m][k] = v[4k+m] */
    for (int r = 0; r < 2; r++)
        for (int k = 0; k < 3; k++)
            T2[r][k] = J[r][0] * v[4 * k + 0] + J[r][1] * v[4 * k + 1] + J[r][2] * v[4 * k + 2];
}
static void sym6_to_mat(const real *c, real S[3][3]) {
    S[0][0] = c[0]; S[0][1] = c[1]; S[0][2] = c[2];
    S[1][0] = c[1]; S[1][1] = c[3]; S[1][2] = c[4];
    S[2][0] = c[2]; S[2][1] = c[4]; S[2][2] = c[5];
}
static void compute_cov2D(const gs_state *st, const real *mean, const real *cov3D, real *cov) {
    real T2[2][3], t[3]; int xi, yi;
    compute_T2(st, mean, T2, t, &xi, &yi);
    real S[3][3]; sym6_to_mat(cov3D, S);
    real ST0[3], ST1[3];
    for (int i = 0; i < 3; i++) {
        ST0[i] = S[i][0] * T2[0][0] + S[i][1] * T2[0][1] + S[i][2] * T2[0][2];
        ST1[i] = S[i][0] * T2[1][0] + S[i][1] * T2[1][1] + S[i][2] * T2[1][2];
    }
    cov[0] = T2[0][0] * ST0[0] + T2[0][1] * ST0[1] + T2[0][2] * ST0[2] + (real)0.3;
    cov[1] = T2[0][0] * ST1[0] + T2[0][1] * ST1[1] + T2[0][2] * ST1[2];
    cov[2] = T2[1][0] * ST1[0] + T2[1][1] * ST1[1] + T2[1][2] * ST1[2] + (real)0.3;
}

/* A1: SH -> RGB, basis as in shared_utils/sh_utils.py:57-100; +0.5; clamp >= 0 with flags. */
static void sh_basis(int deg, const real *d, real *B) {
    real x = d[0], y = d[1], z = d[2];
    B[0] = (real)SH_C0;
    if (deg > 0) {
        B[1] = -(real)SH_C1 * y; B[2] = (real)SH_C1 * z; B[3] = -(real)SH_C1 * x;
        if (deg > 1) {
            real xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
            B[4] = (real)SH_C2[0] * xy; B[5] = (real)SH_C2[1] * yz;
            B[6] = (real)SH_C2[2] * ((real)2 * zz - xx - yy);
            B[7] = (real)SH_C2[3] * xz; B[8] = (real)SH_C2[4] * (xx - yy);
            if (deg > 2) {
                B[9] = (real)SH_C3[0] * y * ((real)3 * xx - yy);
                B[10] = (real)SH_C3[1] * xy * z;
                B[11] = (real)SH_C3[2] * y * ((real)4 * zz - xx - yy);
                B[12] = (real)SH_C3[3] * z * ((real)2 * zz - (real)3 * xx - (real)3 * yy);
                B[13] = (real)SH_C3[4] * x * ((real)4 * zz - xx - yy);
                B[14] = (real)SH_C3[5] * z * (xx - yy);
                B[15] = (real)SH_C3[6] * x * (xx - (real)3 * yy);
            }
        }
    }
}
/* d(basis)/d(dir) for the backward pass */
static void sh_basis_grad(int deg, const real *d, real dB[16][3]) {
    real x = d[0], y = d[1], z = d[2];
    memset(dB, 0, sizeof(real) * 16 * 3);
    if (deg > 0) {
        dB[1][1] = -(real)SH_C1; dB[2][2] = (real)SH_C1; dB[3][0] = -(real)SH_C1;
        if (deg > 1) {
            real xx = x * x, yy = y * y, zz = z * z;
            dB[4][0] = (real)SH_C2[0] * y; dB[4][1] = (real)SH_C2[0] * x;
            dB[5][1] = (real)SH_C2[1] * z; dB[5][2] = (real)SH_C2[1] * y;
            dB[6][0] = (real)SH_C2[2] * (-(real)2 * x); dB[6][1] = (real)SH_C2[2] * (-(real)2 * y);
            dB[6][2] = (real)SH_C2[2] * ((real)4 * z);
            dB[7][0] = (real)SH_C2[3] * z; dB[7][2] = (real)SH_C2[3] * x;
            dB[8][0] = (real)SH_C2[4] * ((real)2 * x); dB[8][1] = (real)SH_C2[4] * (-(real)2 * y);
            if (deg > 2) {
                dB[9][0] = (real)SH_C3[0] * (real)6 * x * y;
                dB[9][1] = (real)SH_C3[0] * ((real)3 * xx - (real)3 * yy);
                dB[10][0] = (real)SH_C3[1] * y * z; dB[10][1] = (real)SH_C3[1] * x * z;
                dB[10][2] = (real)SH_C3[1] * x * y;
                dB[11][0] = (real)SH_C3[2] * (-(real)2 * x * y);
                dB[11][1] = (real)SH_C3[2] * ((real)4 * zz - xx - (real)3 * yy);
                dB[11][2] = (real)SH_C3[2] * ((real)8 * y * z);
                dB[12][0] = (real)SH_C3[3] * (-(real)6 * x * z);
                dB[12][1] = (real)SH_C3[3] * (-(real)6 * y * z);
                dB[12][2] = (real)SH_C3[3] * ((real)6 * zz - (real)3 * xx - (real)3 * yy);
                dB[13][0] = (real)SH_C3[4] * ((real)4 * zz - (real)3 * xx - yy);
                dB[13][1] = (real)SH_C3[4] * (-(real)2 * x * y);
                dB[13][2] = (real)SH_C3[4] * ((real)8 * x * z);
                dB[14][0] = (real)SH_C3[5] * ((real)2 * x * z);
                dB[14][1] = (real)SH_C3[5] * (-(real)2 * y * z);
                dB[14][2] = (real)SH_C3[5] * (xx - yy);
                dB[15][0] = (real)SH_C3[6] * ((real)3 * xx - (real)3 * yy);
                dB[15][1] = (real)SH_C3[6] * (-(real)6 * x * y);
            }
        }
    }
}
static int ncoef(int deg) { return (deg + 1) * (deg + 1); }

static void color_from_sh(const gs_state *st, int idx, const real *means, const real *shs, real *rgb,
                          unsigned char *clamped) {
    real dir[3] = {means[3 * idx] - st->campos[0], means[3 * idx + 1] - st->campos[1],
                   means[3 * idx + 2] - st->campos[2]};
    real len = (real)sqrt((double)(dir[0] * dir[0] + dir[1] * dir[1] + dir[2] * dir[2]));
    dir[0] /= len; dir[1] /= len; dir[2] /= len;
    real B[16];
    sh_basis(st->deg, dir, B);
    const real *sh = shs + (size_t)idx * st->M * 3;
    int nc = ncoef(st->deg);
    for (int c = 0; c < 3; c++) {
        real r = 0;
        for (int k = 0; k < nc; k++) r += B[k] * sh[3 * k + c];
        r += (real)0.5;
        clamped[3 * idx + c] = (r < 0);
        rgb[3 * idx + c] = rmax(r, (real)0);
    }
}

/* accumulation into per-Gaussian sums; atomic only when the tile loop runs multi-threaded (cpu_baseline leg) */
static inline void acc(real *p, real v, int par) {
    if (par) {
#pragma omp atomic
        *p += v;
    } else *p += v;
}

static uint32_t depth_bits(real d) { float f = (float)d; uint32_t u; memcpy(&u, &f, 4); return u; }

typedef struct { uint64_t key; uint32_t id; } kv_t;
static int kv_cmp(const void *a, const void *b) {
    const kv_t *x = (const kv_t *)a, *y = (const kv_t *)b;
    if (x->key != y->key) return x->key < y->key ? -1 : 1;
    return x->id < y->id ? -1 : (x->id > y->id);
}
/* double precision build keeps full-precision depth ordering */
typedef struct { uint32_t tile; real depth; uint32_t id; } kvd_t;
static int kvd_cmp(const void *a, const void *b) {
    const kvd_t *x = (const kvd_t *)a, *y = (const kvd_t *)b;
    if (x->tile != y->tile) return x->tile < y->tile ? -1 : 1;
    if (x->depth != y->depth) return x->depth < y->depth ? -1 : 1;
    return x->id < y->id ? -1 : (x->id > y->id);
}

static void get_rect(const gs_state *st, const real *p, int rad, int *mn, int *mx) {
    mn[0] = imin(st->gx, imax(0, (int)((p[0] - rad) / BLOCK_X)));
    mn[1] = imin(st->gy, imax(0, (int)((p[1] - rad) / BLOCK_Y)));
    mx[0] = imin(st->gx, imax(0, (int)((p[0] + rad + BLOCK_X - 1) / BLOCK_X)));
    mx[1] = imin(st->gy, imax(0, (int)((p[1] + rad + BLOCK_Y - 1) / BLOCK_Y)));
}

void gs_oracle_free(gs_state *st) {
    if (!st) return;
    free(st->depths); free(st->xy); free(st->conic_opacity); free(st->rgb); free(st->cov3D);
    free(st->clamped); free(st->radii); free(st->tiles_touched); free(st->point_list);
    free(st->ranges); free(st->final_T); free(st->n_contrib); free(st);
}

int64_t gs_oracle_num_rendered(const gs_state *st) { return st->D; }
const uint32_t *gs_oracle_point_list(const gs_state *st) { return st->point_list; }
const uint32_t *gs_oracle_ranges(const gs_state *st) { return st->ranges; }
const real *gs_oracle_xy(const gs_state *st) { return st->xy; }
const real *gs_oracle_depths(const gs_state *st) { return st->depths; }
const real *gs_oracle_conic_opacity(const gs_state *st) { return st->conic_opacity; }
const real *gs_oracle_rgb(const gs_state *st) { return st->rgb; }
const int *gs_oracle_tiles_touched(const gs_state *st) { return st->tiles_touched; }
const real *gs_oracle_final_T(const gs_state *st) { return st->final_T; }
const uint32_t *gs_oracle_n_contrib(const gs_state *st) { return st->n_contrib; }
int gs_oracle_sizeof_real(void) { return (int)sizeof(real); }

/* wall time of the (serial) pair sort of the last gs_oracle_forward call: bench.py's cpu_baseline reports it beside the total, because it is the one
 * stage of this port that does not use the cores it is given */
static double g_last_sort_seconds = 0.0;
double gs_oracle_last_sort_seconds(void) { return g_last_sort_seconds; }

/* Forward: A1 preprocess, A2-A5 binning, A6 composite.  Exactly one of shs / colors_precomp and
 * one of (scales, rotations) / cov3D_precomp is non-NULL (boundary rule of the upstream Python
 * wrapper).  Returns an opaque state used by gs_oracle_backward. */
gs_state *gs_oracle_forward(int N, int M, int deg, int W, int H, real tanfovx, real tanfovy,
                            real scale_modifier, const real *bg, const real *view, const real *proj,
                            const real *campos, const real *means3D, const real *shs,
                            const real *colors_precomp, const real *opacities, const real *scales,
                            const real *rotations, const real *cov3D_precomp, int prefiltered,
                            real *out_color, real *out_depth, real *out_alpha, int *radii_out,
                            int nthreads) {
    (void)prefiltered;
    (void)nthreads;
    gs_state *st = (gs_state *)calloc(1, sizeof(gs_state));
    st->N = N; st->M = M; st->deg = deg; st->W = W; st->H = H;
    st->gx = (W + BLOCK_X - 1) / BLOCK_X; st->gy = (H + BLOCK_Y - 1) / BLOCK_Y;
    st->tanfovx = tanfovx; st->tanfovy = tanfovy;
    st->focal_x = W / ((real)2 * tanfovx); st->focal_y = H / ((real)2 * tanfovy);
    st->scale_modifier = scale_modifier;
    memcpy(st->bg, bg, 3 * sizeof(real)); memcpy(st->view, view, 16 * sizeof(real));
    memcpy(st->proj, proj, 16 * sizeof(real)); memcpy(st->campos, campos, 3 * sizeof(real));
    size_t n = (size_t)(N > 0 ? N : 1);
    st->depths = (real *)calloc(n, sizeof(real)); st->xy = (real *)calloc(2 * n, sizeof(real));
    st->conic_opacity = (real *)calloc(4 * n, sizeof(real)); st->rgb = (real *)calloc(3 * n, sizeof(real));
    st->cov3D = (real *)calloc(6 * n, sizeof(real)); st->clamped = (unsigned char *)calloc(3 * n, 1);
    st->radii = (int *)calloc(n, sizeof(int)); st->tiles_touched = (int *)calloc(n, sizeof(int));
    int tiles = st->gx * st->gy;
    size_t P = (size_t)W * H;
    st->ranges = (uint32_t *)calloc((size_t)2 * (tiles > 0 ? tiles : 1), sizeof(uint32_t));
    st->final_T = (real *)calloc(P > 0 ? P : 1, sizeof(real));
    st->n_contrib = (uint32_t *)calloc(P > 0 ? P : 1, sizeof(uint32_t));

    /* ---- A1 preprocess ---- */
#pragma omp parallel for schedule(static) num_threads(nthreads > 0 ? nthreads : 1)
    for (int idx = 0; idx < N; idx++) {
        const real *p = means3D + 3 * idx;
        real pv[3], ph[4];
        xform4x3(p, st->view, pv);
        if (pv[2] <= (real)0.2) continue; /* near cull */
        xform4x4(p, st->proj, ph);
        real pw = (real)1 / (ph[3] + (real)0.0000001);
        real pp[2] = {ph[0] * pw, ph[1] * pw};
        const real *c3;
        if (cov3D_precomp) c3 = cov3D_precomp + 6 * idx;
        else { compute_cov3D(scales + 3 * idx, scale_modifier, rotations + 4 * idx, st->cov3D + 6 * idx); c3 = st->cov3D + 6 * idx; }
        real cov[3];
        compute_cov2D(st, p, c3, cov);
        real det = cov[0] * cov[2] - cov[1] * cov[1];
        if (det == 0) continue;
        real di = (real)1 / det;
        real conic[3] = {cov[2] * di, -cov[1] * di, cov[0] * di};
        real mid = (real)0.5 * (cov[0] + cov[2]);
        real sq = (real)sqrt((double)rmax((real)0.1, mid * mid - det));
        real l1 = mid + sq, l2 = mid - sq;
        int rad = (int)ceil((double)((real)3 * (real)sqrt((double)rmax(l1, l2))));
        real pix[2] = {((pp[0] + (real)1) * W - (real)1) * (real)0.5, ((pp[1] + (real)1) * H - (real)1) * (real)0.5};
        int mn[2], mx[2];
        get_rect(st, pix, rad, mn, mx);
        if ((mx[0] - mn[0]) * (mx[1] - mn[1]) == 0) continue;
        if (colors_precomp) { for (int c = 0; c < 3; c++) st->rgb[3 * idx + c] = colors_precomp[3 * idx + c]; }
        else color_from_sh(st, idx, means3D, shs, st->rgb, st->clamped);
        st->depths[idx] = pv[2];
        st->radii[idx] = rad;
        st->xy[2 * idx] = pix[0]; st->xy[2 * idx + 1] = pix[1];
        st->conic_opacity[4 * idx] = conic[0]; st->conic_opacity[4 * idx + 1] = conic[1];
        st->conic_opacity[4 * idx + 2] = conic[2]; st->conic_opacity[4 * idx + 3] = opacities[idx];
        st->tiles_touched[idx] = (mx[1] - mn[1]) * (mx[0] - mn[0]);
    }
    if (radii_out) memcpy(radii_out, st->radii, (size_t)N * sizeof(int));

    /* ---- A2-A5: duplicate per touched tile, order by (tile, depth, index), per-tile ranges ---- */
    int64_t D = 0;
    for (int i = 0; i < N; i++) D += st->tiles_touched[i];
    st->D = D;
    st->point_list = (uint32_t *)malloc((size_t)(D > 0 ? D : 1) * sizeof(uint32_t));
    if (sizeof(real) == 4) {
        kv_t *kv = (kv_t *)malloc((size_t)(D > 0 ? D : 1) * sizeof(kv_t));
        int64_t off = 0;
        for (int i = 0; i < N; i++) {
            if (st->radii[i] <= 0) continue;
            int mn[2], mx[2];
            get_rect(st, st->xy + 2 * i, st->radii[i], mn, mx);
            for (int y = mn[1]; y < mx[1]; y++)
                for (int x = mn[0]; x < mx[0]; x++) {
                    kv[off].key = ((uint64_t)(uint32_t)(y * st->gx + x) << 32) | depth_bits(st->depths[i]);
                    kv[off].id = (uint32_t)i; off++;
                }
        }
        { const double t_ = omp_get_wtime(); qsort(kv, (size_t)D, sizeof(kv_t), kv_cmp); g_last_sort_seconds = omp_get_wtime() - t_; }
        for (int64_t i = 0; i < D; i++) {
            st->point_list[i] = kv[i].id;
            uint32_t t = (uint32_t)(kv[i].key >> 32);
            if (i == 0 || t != (uint32_t)(kv[i - 1].key >> 32)) st->ranges[2 * t] = (uint32_t)i;
            if (i == D - 1 || t != (uint32_t)(kv[i + 1].key >> 32)) st->ranges[2 * t + 1] = (uint32_t)(i + 1);
        }
        free(kv);
    } else {
        kvd_t *kv = (kvd_t *)malloc((size_t)(D > 0 ? D : 1) * sizeof(kvd_t));
        int64_t off = 0;
        for (int i = 0; i < N; i++) {
            if (st->radii[i] <= 0) continue;
            int mn[2], mx[2];
            get_rect(st, st->xy + 2 * i, st->radii[i], mn, mx);
            for (int y = mn[1]; y < mx[1]; y++)
                for (int x = mn[0]; x < mx[0]; x++) {
                    kv[off].tile = (uint32_t)(y * st->gx + x); kv[off].depth = st->depths[i];
                    kv[off].id = (uint32_t)i; off++;
                }
        }
        { const double t_ = omp_get_wtime(); qsort(kv, (size_t)D, sizeof(kvd_t), kvd_cmp); g_last_sort_seconds = omp_get_wtime() - t_; }
        for (int64_t i = 0; i < D; i++) {
            st->point_list[i] = kv[i].id;
            uint32_t t = kv[i].tile;
            if (i == 0 || t != kv[i - 1].tile) st->ranges[2 * t] = (uint32_t)i;
            if (i == D - 1 || t != kv[i + 1].tile) st->ranges[2 * t + 1] = (uint32_t)(i + 1);
        }
        free(kv);
    }

    /* ---- A6 composite, front to back, one pixel at a time ---- */
#pragma omp parallel for schedule(dynamic, 1) num_threads(nthreads > 0 ? nthreads : 1)
    for (int tile = 0; tile < tiles; tile++) {
        int tx = tile % st->gx, ty = tile / st->gx;
        uint32_t r0 = st->ranges[2 * tile], r1 = st->ranges[2 * tile + 1];
        for (int py = ty * BLOCK_Y; py < imin(H, (ty + 1) * BLOCK_Y); py++)
            for (int px = tx * BLOCK_X; px < imin(W, (tx + 1) * BLOCK_X); px++) {
                real pf[2] = {(real)px, (real)py};
                real T = 1, C[3] = {0, 0, 0}, Dd = 0, A = 0;
                uint32_t contributor = 0, last = 0;
                for (uint32_t j = r0; j < r1; j++) {
                    uint32_t g = st->point_list[j];
                    contributor++;
                    real dx = st->xy[2 * g] - pf[0], dy = st->xy[2 * g + 1] - pf[1];
                    const real *co = st->conic_opacity + 4 * g;
                    real power = -(real)0.5 * (co[0] * dx * dx + co[2] * dy * dy) - co[1] * dx * dy;
                    if (power > 0) continue;
                    real alpha = rmin((real)0.99, co[3] * (real)exp((double)power));
                    if (alpha < (real)1 / (real)255) continue;
                    real testT = T * ((real)1 - alpha);
                    if (testT < (real)0.0001) break;
                    real w = alpha * T;
                    C[0] += st->rgb[3 * g] * w; C[1] += st->rgb[3 * g + 1] * w; C[2] += st->rgb[3 * g + 2] * w;
                    Dd += st->depths[g] * w; A += w;
                    T = testT; last = contributor;
                }
                size_t pid = (size_t)py * W + px;
                st->final_T[pid] = T; st->n_contrib[pid] = last;
                for (int c = 0; c < 3; c++) out_color[(size_t)c * P + pid] = C[c] + T * st->bg[c];
                out_depth[pid] = Dd; out_alpha[pid] = A;
            }
    }
    return st;
}

/* Test support: which Gaussians are blended into a flagged pixel?  The full-size parity tests flag the pixels where a float32 rasterizer and this
 * oracle in float64 took a different per-splat decision (alpha >= 1/255, T < 1e-4, ceil(3 sigma)) and then show where the gradient entries outside
 * tolerance come from.  pixel_flags [H*W] (non-zero = flagged)  ->  gauss_flags [N] |= 1 for every Gaussian the composite loop blends into a flagged
 * pixel.  The walk is the forward loop with both thresholds loosened (alpha from 0.98/255, termination at T < 0.5e-4): the splats whose own decision
 * can flip, and the few behind this oracle's stopping point a kernel that stops one splat later still blends, count. */
void gs_oracle_taint(const gs_state *st, const unsigned char *pixel_flags, unsigned char *gauss_flags) {
    int W = st->W, H = st->H;
    int tiles = st->gx * st->gy;
    for (int tile = 0; tile < tiles; tile++) {
        int tx = tile % st->gx, ty = tile / st->gx;
        uint32_t r0 = st->ranges[2 * tile], r1 = st->ranges[2 * tile + 1];
        for (int py = ty * BLOCK_Y; py < imin(H, (ty + 1) * BLOCK_Y); py++)
            for (int px = tx * BLOCK_X; px < imin(W, (tx + 1) * BLOCK_X); px++) {
                if (!pixel_flags[(size_t)py * W + px]) continue;
                real T = 1;
                for (uint32_t j = r0; j < r1; j++) {
                    uint32_t g = st->point_list[j];
                    real dx = st->xy[2 * g] - (real)px, dy = st->xy[2 * g + 1] - (real)py;
                    const real *co = st->conic_opacity + 4 * g;
                    real power = -(real)0.5 * (co[0] * dx * dx + co[2] * dy * dy) - co[1] * dx * dy;
                    if (power > (real)1e-6) continue;
                    real alpha = rmin((real)0.99, co[3] * (real)exp((double)power));
                    if (alpha < (real)0.98 / (real)255) continue;
                    gauss_flags[g] |= 1;
                    real testT = T * ((real)1 - alpha);
                    if (testT < (real)0.00005) break;
                    T = testT;
                }
            }
    }
}

/* Backward: A7 composite (back to front), A8 preprocess.  All outputs must be zero-initialised
 * by the caller except where noted.  dL_dconic has 4 entries per Gaussian (xx, xy/2-convention,
 * unused, yy) exactly like the upstream scratch layout. */
void gs_oracle_backward(gs_state *st, const real *means3D, const real *shs, const real *colors_precomp,
                        const real *scales, const real *rotations, const real *cov3D_precomp,
                        const real *dL_dcolor_px, const real *dL_ddepth_px, const real *dL_dalpha_px,
                        real *dL_dmeans2D /*N*3*/, real *dL_dconic /*N*4*/, real *dL_dopacity /*N*/,
                        real *dL_dcolors /*N*3*/, real *dL_ddepths /*N*/, real *dL_dmeans3D /*N*3*/,
                        real *dL_dcov3D /*N*6*/, real *dL_dsh /*N*M*3*/, real *dL_dscales /*N*3*/,
                        real *dL_drots /*N*4*/, int nthreads) {
    const int par = nthreads > 1;
    int N = st->N, W = st->W, H = st->H;
    size_t P = (size_t)W * H;
    int tiles = st->gx * st->gy;
    const real ddelx_dx = (real)0.5 * W, ddely_dy = (real)0.5 * H;

    /* ---- A7 ---- (nthreads<=1: sequential, summation order = tile-major, row-major pixels, back to front) */
#pragma omp parallel for schedule(dynamic, 1) num_threads(nthreads > 0 ? nthreads : 1) if (par)
    for (int tile = 0; tile < tiles; tile++) {
        int tx = tile % st->gx, ty = tile / st->gx;
        uint32_t r0 = st->ranges[2 * tile], r1 = st->ranges[2 * tile + 1];
        for (int py = ty * BLOCK_Y; py < imin(H, (ty + 1) * BLOCK_Y); py++)
            for (int px = tx * BLOCK_X; px < imin(W, (tx + 1) * BLOCK_X); px++) {
                size_t pid = (size_t)py * W + px;
                real pf[2] = {(real)px, (real)py};
                real T_final = st->final_T[pid], T = T_final;
                uint32_t last = st->n_contrib[pid];
                real dLp[3] = {dL_dcolor_px[pid], dL_dcolor_px[P + pid], dL_dcolor_px[2 * P + pid]};
                real dLd = dL_ddepth_px ? dL_ddepth_px[pid] : 0, dLa = dL_dalpha_px ? dL_dalpha_px[pid] : 0;
                real accum[3] = {0, 0, 0}, accum_d = 0, accum_a = 0;
                real last_alpha = 0, last_color[3] = {0, 0, 0}, last_depth = 0;
                real bg_dot = st->bg[0] * dLp[0] + st->bg[1] * dLp[1] + st->bg[2] * dLp[2];
                for (uint32_t k = last; k-- > 0;) { /* contributors last-1 .. 0 */
                    uint32_t g = st->point_list[r0 + k];
                    (void)r1;
                    real dx = st->xy[2 * g] - pf[0], dy = st->xy[2 * g + 1] - pf[1];
                    const real *co = st->conic_opacity + 4 * g;
                    real power = -(real)0.5 * (co[0] * dx * dx + co[2] * dy * dy) - co[1] * dx * dy;
                    if (power > 0) continue;
                    real G = (real)exp((double)power);
                    real alpha = rmin((real)0.99, co[3] * G);
                    if (alpha < (real)1 / (real)255) continue;
                    T = T / ((real)1 - alpha);
                    real dch = alpha * T;
                    real dL_dalpha = 0;
                    for (int c = 0; c < 3; c++) {
                        real col = st->rgb[3 * g + c];
                        accum[c] = last_alpha * last_color[c] + ((real)1 - last_alpha) * accum[c];
                        last_color[c] = col;
                        dL_dalpha += (col - accum[c]) * dLp[c];
                        acc(&dL_dcolors[3 * g + c], dch * dLp[c], par);
                    }
                    real cd = st->depths[g];
                    accum_d = last_alpha * last_depth + ((real)1 - last_alpha) * accum_d;
                    last_depth = cd;
                    dL_dalpha += (cd - accum_d) * dLd;
                    acc(&dL_ddepths[g], dch * dLd, par);
                    accum_a = last_alpha + ((real)1 - last_alpha) * accum_a;
                    dL_dalpha += ((real)1 - accum_a) * dLa;
                    dL_dalpha *= T;
                    last_alpha = alpha;
                    dL_dalpha += (-T_final / ((real)1 - alpha)) * bg_dot;
                    real dL_dG = co[3] * dL_dalpha;
                    real gdx = G * dx, gdy = G * dy;
                    real dG_ddelx = -gdx * co[0] - gdy * co[1];
                    real dG_ddely = -gdy * co[2] - gdx * co[1];
                    acc(&dL_dmeans2D[3 * g], dL_dG * dG_ddelx * ddelx_dx, par);
                    acc(&dL_dmeans2D[3 * g + 1], dL_dG * dG_ddely * ddely_dy, par);
                    acc(&dL_dconic[4 * g], -(real)0.5 * gdx * dx * dL_dG, par);
                    acc(&dL_dconic[4 * g + 1], -(real)0.5 * gdx * dy * dL_dG, par);
                    acc(&dL_dconic[4 * g + 3], -(real)0.5 * gdy * dy * dL_dG, par);
                    acc(&dL_dopacity[g], G * dL_dalpha, par);
                }
            }
    }

    /* ---- A8 ---- */
#pragma omp parallel for schedule(static) num_threads(nthreads > 0 ? nthreads : 1) if (par)
    for (int idx = 0; idx < N; idx++) {
        if (st->radii[idx] <= 0) continue;
        const real *mean = means3D + 3 * idx;
        const real *c3 = cov3D_precomp ? cov3D_precomp + 6 * idx : st->cov3D + 6 * idx;
        /* cov2D -> cov3D, mean (through J) */
        real T2[2][3], t[3]; int xin, yin;
        compute_T2(st, mean, T2, t, &xin, &yin);
        real S[3][3]; sym6_to_mat(c3, S);
        real ST0[3], ST1[3];
        for (int i = 0; i < 3; i++) {
            ST0[i] = S[i][0] * T2[0][0] + S[i][1] * T2[0][1] + S[i][2] * T2[0][2];
            ST1[i] = S[i][0] * T2[1][0] + S[i][1] * T2[1][1] + S[i][2] * T2[1][2];
        }
        real a = T2[0][0] * ST0[0] + T2[0][1] * ST0[1] + T2[0][2] * ST0[2] + (real)0.3;
        real b = T2[0][0] * ST1[0] + T2[0][1] * ST1[1] + T2[0][2] * ST1[2];
        real c = T2[1][0] * ST1[0] + T2[1][1] * ST1[1] + T2[1][2] * ST1[2] + (real)0.3;
        real dcx = dL_dconic[4 * idx], dcy = dL_dconic[4 * idx + 1], dcz = dL_dconic[4 * idx + 3];
        real denom = a * c - b * b;
        real d2i = (real)1 / (denom * denom + (real)0.0000001);
        real dL_da = 0, dL_db = 0, dL_dc = 0;
        real *dcov = dL_dcov3D + 6 * idx;
        if (d2i != 0) {
            dL_da = d2i * (-c * c * dcx + (real)2 * b * c * dcy + (denom - a * c) * dcz);
            dL_dc = d2i * (-a * a * dcz + (real)2 * a * b * dcy + (denom - a * c) * dcx);
            dL_db = d2i * (real)2 * (b * c * dcx - (denom + (real)2 * b * b) * dcy + a * b * dcz);
            dcov[0] = T2[0][0] * T2[0][0] * dL_da + T2[0][0] * T2[1][0] * dL_db + T2[1][0] * T2[1][0] * dL_dc;
            dcov[3] = T2[0][1] * T2[0][1] * dL_da + T2[0][1] * T2[1][1] * dL_db + T2[1][1] * T2[1][1] * dL_dc;
            dcov[5] = T2[0][2] * T2[0][2] * dL_da + T2[0][2] * T2[1][2] * dL_db + T2[1][2] * T2[1][2] * dL_dc;
            dcov[1] = (real)2 * T2[0][0] * T2[0][1] * dL_da + (T2[0][0] * T2[1][1] + T2[0][1] * T2[1][0]) * dL_db + (real)2 * T2[1][0] * T2[1][1] * dL_dc;
            dcov[2] = (real)2 * T2[0][0] * T2[0][2] * dL_da + (T2[0][0] * T2[1][2] + T2[0][2] * T2[1][0]) * dL_db + (real)2 * T2[1][0] * T2[1][2] * dL_dc;
            dcov[4] = (real)2 * T2[0][2] * T2[0][1] * dL_da + (T2[0][1] * T2[1][2] + T2[0][2] * T2[1][1]) * dL_db + (real)2 * T2[1][1] * T2[1][2] * dL_dc;
        }
        real dT[2][3];
        for (int k = 0; k < 3; k++) {
            dT[0][k] = (real)2 * ST0[k] * dL_da + ST1[k] * dL_db;
            dT[1][k] = (real)2 * ST1[k] * dL_dc + ST0[k] * dL_db;
        }
        const real *v = st->view;
        real dJ00 = v[0] * dT[0][0] + v[4] * dT[0][1] + v[8] * dT[0][2];
        real dJ02 = v[2] * dT[0][0] + v[6] * dT[0][1] + v[10] * dT[0][2];
        real dJ11 = v[1] * dT[1][0] + v[5] * dT[1][1] + v[9] * dT[1][2];
        real dJ12 = v[2] * dT[1][0] + v[6] * dT[1][1] + v[10] * dT[1][2];
        real tz = (real)1 / t[2], tz2 = tz * tz, tz3 = tz2 * tz;
        real dtx = (real)xin * (-st->focal_x * tz2 * dJ02);
        real dty = (real)yin * (-st->focal_y * tz2 * dJ12);
        real dtz = -st->focal_x * tz2 * dJ00 - st->focal_y * tz2 * dJ11 +
                   ((real)2 * st->focal_x * t[0]) * tz3 * dJ02 + ((real)2 * st->focal_y * t[1]) * tz3 * dJ12;
        real dmean[3];
        for (int j = 0; j < 3; j++) dmean[j] = v[4 * j] * dtx + v[4 * j + 1] * dty + v[4 * j + 2] * dtz;

        /* mean2D (NDC-scaled) -> mean3D through the projection */
        const real *pr = st->proj;
        real mh[4]; xform4x4(mean, pr, mh);
        real mw = (real)1 / (mh[3] + (real)0.0000001);
        real mul1 = mh[0] * mw * mw, mul2 = mh[1] * mw * mw;
        real g2x = dL_dmeans2D[3 * idx], g2y = dL_dmeans2D[3 * idx + 1];
        for (int j = 0; j < 3; j++)
            dmean[j] += (pr[4 * j] * mw - pr[4 * j + 3] * mul1) * g2x + (pr[4 * j + 1] * mw - pr[4 * j + 3] * mul2) * g2y;
        /* depth output -> mean3D (fork) */
        {
            real mul3 = v[2] * mean[0] + v[6] * mean[1] + v[10] * mean[2] + v[14];
            for (int j = 0; j < 3; j++) dmean[j] += (v[4 * j + 2] - v[4 * j + 3] * mul3) * dL_ddepths[idx];
        }
        /* colour -> SH coefficients and view direction */
        if (!colors_precomp) {
            real dorig[3] = {mean[0] - st->campos[0], mean[1] - st->campos[1], mean[2] - st->campos[2]};
            real s2 = dorig[0] * dorig[0] + dorig[1] * dorig[1] + dorig[2] * dorig[2];
            real len = (real)sqrt((double)s2);
            real dir[3] = {dorig[0] / len, dorig[1] / len, dorig[2] / len};
            real B[16], dB[16][3];
            sh_basis(st->deg, dir, B); sh_basis_grad(st->deg, dir, dB);
            real dRGB[3];
            for (int c3i = 0; c3i < 3; c3i++) dRGB[c3i] = st->clamped[3 * idx + c3i] ? 0 : dL_dcolors[3 * idx + c3i];
            const real *sh = shs + (size_t)idx * st->M * 3;
            real *dsh = dL_dsh + (size_t)idx * st->M * 3;
            int nc = ncoef(st->deg);
            real ddir[3] = {0, 0, 0};
            for (int k = 0; k < nc; k++)
                for (int c3i = 0; c3i < 3; c3i++) {
                    dsh[3 * k + c3i] = B[k] * dRGB[c3i];
                    for (int a3 = 0; a3 < 3; a3++) ddir[a3] += dB[k][a3] * sh[3 * k + c3i] * dRGB[c3i];
                }
            real inv32 = (real)1 / (real)sqrt((double)(s2 * s2 * s2));
            real vx = dorig[0], vy = dorig[1], vz = dorig[2];
            dmean[0] += ((s2 - vx * vx) * ddir[0] - vy * vx * ddir[1] - vz * vx * ddir[2]) * inv32;
            dmean[1] += (-vx * vy * ddir[0] + (s2 - vy * vy) * ddir[1] - vz * vy * ddir[2]) * inv32;
            dmean[2] += (-vx * vz * ddir[0] - vy * vz * ddir[1] + (s2 - vz * vz) * ddir[2]) * inv32;
        }
        for (int j = 0; j < 3; j++) dL_dmeans3D[3 * idx + j] = dmean[j];

        /* cov3D -> scale, rotation.  d/dscale as the dependency returns it: the derivative w.r.t. (scale_modifier * scale),
         * i.e. without the scale_modifier factor; gs_oracle_set_exact_dscale(1) switches to the exact derivative. */
        if (!cov3D_precomp) {
            const real *q = rotations + 4 * idx, *sc = scales + 3 * idx;
            real R[3][3]; quat_to_R(q, R);
            real s[3] = {st->scale_modifier * sc[0], st->scale_modifier * sc[1], st->scale_modifier * sc[2]};
            real Gm[3][3] = {{dcov[0], (real)0.5 * dcov[1], (real)0.5 * dcov[2]},
                             {(real)0.5 * dcov[1], dcov[3], (real)0.5 * dcov[4]},
                             {(real)0.5 * dcov[2], (real)0.5 * dcov[4], dcov[5]}};
            real dM[3][3]; /* M' = R diag(s);  dL/dM' = 2 G M' */
            for (int i = 0; i < 3; i++)
                for (int k = 0; k < 3; k++)
                    dM[i][k] = (real)2 * (Gm[i][0] * R[0][k] + Gm[i][1] * R[1][k] + Gm[i][2] * R[2][k]) * s[k];
            real dR[3][3];
            for (int k = 0; k < 3; k++) {
                dL_dscales[3 * idx + k] = (g_exact_dscale ? st->scale_modifier : (real)1) * (dM[0][k] * R[0][k] + dM[1][k] * R[1][k] + dM[2][k] * R[2][k]);
                for (int i = 0; i < 3; i++) dR[i][k] = dM[i][k] * s[k];
            }
            real r = q[0], x = q[1], y = q[2], z = q[3];
            dL_drots[4 * idx + 0] = (real)2 * (-z * dR[0][1] + y * dR[0][2] + z * dR[1][0] - x * dR[1][2] - y * dR[2][0] + x * dR[2][1]);
            dL_drots[4 * idx + 1] = (real)2 * (y * dR[0][1] + z * dR[0][2] + y * dR[1][0] - (real)2 * x * dR[1][1] - r * dR[1][2] + z * dR[2][0] + r * dR[2][1] - (real)2 * x * dR[2][2]);
            dL_drots[4 * idx + 2] = (real)2 * (-(real)2 * y * dR[0][0] + x * dR[0][1] + r * dR[0][2] + x * dR[1][0] + z * dR[1][2] - r * dR[2][0] + z * dR[2][1] - (real)2 * y * dR[2][2]);
            dL_drots[4 * idx + 3] = (real)2 * (-(real)2 * z * dR[0][0] - r * dR[0][1] + x * dR[0][2] + r * dR[1][0] - (real)2 * z * dR[1][1] + y * dR[1][2] + x * dR[2][0] + y * dR[2][1]);
        }
    }
}

/* A9: frustum test exported by the dependency as mark_visible (view z > 0.2). */
void gs_oracle_mark_visible(int N, const real *means3D, const real *view, const real *proj, unsigned char *present) {
    (void)proj;
    for (int i = 0; i < N; i++) {
        real pv[3];
        xform4x3(means3D + 3 * i, view, pv);
        present[i] = pv[2] > (real)0.2;
    }
}
