/*
 * mesh_oracle.c -- CPU restatement of the four differentiable mesh-rendering ops that
 * ComfyUI-3D-Pack's DiffRastRenderer calls through `nvdiffrast.torch`:
 *   rasterize, interpolate, texture (linear / nearest), antialias -- forward and backward.
 *
 * THIS IS TEST INFRASTRUCTURE (only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
 * leg may load it; the product under comfyui-3d-pack_amd/ never does).
 *
 * PARITY UNPINNED except rasterize + interpolate forward, which tests/test_ref_pin.py holds to outputs of the
 * reference's own custom_rasterizer (compiled by oracle/ref_build.py, vectors in tests/golden/mesh_hy_raster.npz).
 * nvdiffrast itself is an un-vendored third-party wheel
 * (/root/reference/_Pre_Builds/_Build_Scripts/dependencies.txt:3, my-reqs.txt:74: nvdiffrast 0.3.3)
 * and the reference ships no tests or golden vectors for this path.  This
 * file restates the published semantics of those ops anchored
 * on the reference's call sites:
 *   /root/reference/MVs_Algorithms/DiffRastMesh/diff_mesh_renderer.py:97   rasterize(ctx, v_clip[1,V,4], f, (h,w))
 *   :101,138  antialias(color, rast, v_clip, f)         :104,110,131  interpolate(attr, rast, tri[, rast_db, diff_attrs='all'])
 *   :105      texture(tex[1,Ht,Wt,3], uv, uv_da=..., filter_mode='linear')  (uv_da is ignored for 'linear')
 * Conventions restated: clip-space input; pixel (x,y) centre at NDC ((x+.5)2/W-1, (y+.5)2/H-1), row 0 at
 * NDC y = -1; rast = (u, v, z/w, tri_id+1), u,v weight vertices 0 and 1, perspective correct; rast_db =
 * (du/dX, du/dY, dv/dX, dv/dY) per pixel; coverage on vertices snapped to 1/16 pixel with exact integer
 * edge functions and a top-left style tie rule; nearest z/w wins, ties to the lower triangle index;
 * texture: texel centres at half integers, boundary wrap/clamp; antialias: for horizontally/vertically
 * adjacent pixels with different triangle ids, the nearer triangle's silhouette edge that crosses the
 * segment between the pixel centres blends the two colours linearly by the crossing position.
 * Gradients: u,v -> clip x,y,w; attributes; texels and uv; colours and the silhouette edge's vertices.
 * (Gradients w.r.t. rast_db / out_da are not propagated: no consumer on the reference's path.)
 * Pinned by analytic cases, invariants and central finite differences in float64 (tests/test_mesh_oracle.py).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#ifndef REAL
#define REAL float
#endif
typedef REAL real;

static real clampr(real v, real lo, real hi) { return v < lo ? lo : (v > hi ? hi : v); }

int mesh_oracle_sizeof_real(void) { return (int)sizeof(real); }

/* ------------------------------------------------------------------ rasterize */
static int64_t snap(real ndc, int S) {   /* 1/16-pixel units from the lower/left border, round to nearest even */
    double v = (double)((real)(ndc * (real)(S * 8))) + (double)S * 8.0;
    return (int64_t)nearbyint(v);
}
/* tie rule for a pixel centre exactly on an edge with (orientation-normalised) direction (dx,dy) */
static int edge_owns_tie(int64_t dx, int64_t dy) { return dy > 0 || (dy == 0 && dx < 0); }

typedef struct { real b0, b1, zw, iw; real dudx, dudy, dvdx, dvdy; } frag_t;

static void shade(const real *p0, const real *p1, const real *p2, real fx, real fy, real xs, real ys, frag_t *f) {
    real p0x = p0[0] - fx * p0[3], p0y = p0[1] - fy * p0[3];
    real p1x = p1[0] - fx * p1[3], p1y = p1[1] - fy * p1[3];
    real p2x = p2[0] - fx * p2[3], p2y = p2[1] - fy * p2[3];
    real a0 = p1x * p2y - p1y * p2x, a1 = p2x * p0y - p2y * p0x, a2 = p0x * p1y - p0y * p1x;
    real iw = (real)1 / (a0 + a1 + a2);
    f->iw = iw;
    f->b0 = a0 * iw; f->b1 = a1 * iw;
    real z = p0[2] * a0 + p1[2] * a1 + p2[2] * a2, w = p0[3] * a0 + p1[3] * a1 + p2[3] * a2;
    f->zw = z / w;
    real dfxdx = xs * iw, dfydy = ys * iw;
    real da0dx = p2[1] * p1[3] - p1[1] * p2[3], da0dy = p1[0] * p2[3] - p2[0] * p1[3];
    real da1dx = p0[1] * p2[3] - p2[1] * p0[3], da1dy = p2[0] * p0[3] - p0[0] * p2[3];
    real da2dx = p1[1] * p0[3] - p0[1] * p1[3], da2dy = p0[0] * p1[3] - p1[0] * p0[3];
    real datdx = da0dx + da1dx + da2dx, datdy = da0dy + da1dy + da2dy;
    f->dudx = dfxdx * (f->b0 * datdx - da0dx); f->dudy = dfydy * (f->b0 * datdy - da0dy);
    f->dvdx = dfxdx * (f->b1 * datdx - da1dx); f->dvdy = dfydy * (f->b1 * datdy - da1dy);
}

/* prev = the previous depth-peeling layer's rast output, or NULL: with it, a fragment survives only where that layer had a surface and
 * only if its z/w is strictly greater than that surface's (the dependency's DepthPeeler.rasterize_next_layer as documented; reference
 * use: Gen_3D_Modules/InstantMesh/models/geometry/render/neural_render.py:103-106) */
static void rasterize_impl(const real *pos, const int32_t *tri, int B, int V, int T, int H, int W, const real *prev, real *rast, real *rast_db) {
    const real xs = (real)2 / W, ys = (real)2 / H;
    size_t P = (size_t)H * W;
    real *zbest = (real *)malloc(P * sizeof(real));
    int32_t *tbest = (int32_t *)malloc(P * sizeof(int32_t));
    for (int b = 0; b < B; b++) {
        const real *pb = pos + (size_t)b * V * 4;
        for (size_t i = 0; i < P; i++) { zbest[i] = 0; tbest[i] = -1; }
        for (int t = 0; t < T; t++) {
            int i0 = tri[3 * t], i1 = tri[3 * t + 1], i2 = tri[3 * t + 2];
            if (i0 < 0 || i0 >= V || i1 < 0 || i1 >= V || i2 < 0 || i2 >= V) continue;
            const real *p0 = pb + 4 * i0, *p1 = pb + 4 * i1, *p2 = pb + 4 * i2;
            if (!(p0[3] > 0 && p1[3] > 0 && p2[3] > 0)) {
                /* A vertex at or behind the camera plane (w <= 0): the dependency clips such a triangle against the view volume.  Restated per
                 * pixel: the part in front of the near plane z = -w is what survives.  Pixel bounding box from the polygon clipped against
                 * z + w >= 0; coverage by the homogeneous barycentrics (all >= 0, inclusive) with interpolated w > 0; the usual per-pixel
                 * -1 <= z/w <= 1 test; barycentrics / depth from the ORIGINAL vertices (the homogeneous form needs no divide by a vertex w). */
                const real *pp[3] = {p0, p1, p2};
                real bx0 = 0, bx1 = 0, by0 = 0, by1 = 0;
                int nb = 0, full = 0;
                for (int k = 0; k < 3; k++) {
                    const real *a = pp[k], *c = pp[(k + 1) % 3];
                    real da = a[2] + a[3], dc = c[2] + c[3];
                    real cand[2][4];
                    int nc = 0;
                    if (da >= 0) { for (int j = 0; j < 4; j++) cand[nc][j] = a[j]; nc++; }
                    if ((da >= 0) != (dc >= 0)) { real tt = da / (da - dc); for (int j = 0; j < 4; j++) cand[nc][j] = a[j] + tt * (c[j] - a[j]); nc++; }
                    for (int q = 0; q < nc; q++) {
                        if (!(cand[q][3] > (real)1e-12)) { full = 1; continue; }
                        real nx = cand[q][0] / cand[q][3], ny = cand[q][1] / cand[q][3];
                        if (!nb) { bx0 = bx1 = nx; by0 = by1 = ny; nb = 1; }
                        else { if (nx < bx0) bx0 = nx; if (nx > bx1) bx1 = nx; if (ny < by0) by0 = ny; if (ny > by1) by1 = ny; }
                    }
                }
                if (!nb && !full) continue;                                   /* entirely in front of the near plane's wrong side */
                int64_t cpx0 = 0, cpx1 = W - 1, cpy0 = 0, cpy1 = H - 1;
                if (!full) {
                    double fx0 = ((double)bx0 + 1) * 0.5 * W - 1.5, fx1 = ((double)bx1 + 1) * 0.5 * W + 0.5, fy0 = ((double)by0 + 1) * 0.5 * H - 1.5, fy1 = ((double)by1 + 1) * 0.5 * H + 0.5;
                    if (fx0 > 0) cpx0 = fx0 > W ? W : (int64_t)fx0;
                    if (fy0 > 0) cpy0 = fy0 > H ? H : (int64_t)fy0;
                    if (fx1 < W - 1) cpx1 = fx1 < -1 ? -1 : (int64_t)fx1;
                    if (fy1 < H - 1) cpy1 = fy1 < -1 ? -1 : (int64_t)fy1;
                }
                for (int64_t py = cpy0; py <= cpy1; py++)
                    for (int64_t px = cpx0; px <= cpx1; px++) {
                        real fx = xs * ((real)px + (real)0.5) - (real)1, fy = ys * ((real)py + (real)0.5) - (real)1;
                        real q0x = p0[0] - fx * p0[3], q0y = p0[1] - fy * p0[3], q1x = p1[0] - fx * p1[3], q1y = p1[1] - fy * p1[3];
                        real q2x = p2[0] - fx * p2[3], q2y = p2[1] - fy * p2[3];
                        real a0 = q1x * q2y - q1y * q2x, a1 = q2x * q0y - q2y * q0x, a2 = q0x * q1y - q0y * q1x, sum = a0 + a1 + a2;
                        if (sum == 0) continue;
                        real b0 = a0 / sum, b1 = a1 / sum, b2 = a2 / sum;
                        if (!(b0 >= 0 && b1 >= 0 && b2 >= 0)) continue;
                        if (!(b0 * p0[3] + b1 * p1[3] + b2 * p2[3] > 0)) continue;
                        frag_t f;
                        shade(p0, p1, p2, fx, fy, xs, ys, &f);
                        if (!(f.zw >= -1 && f.zw <= 1)) continue;
                        size_t pid = (size_t)py * W + px;
                        if (prev) {
                            const real *pr = prev + ((size_t)b * P + pid) * 4;
                            if (pr[3] == 0 || !(f.zw > pr[2])) continue;
                        }
                        if (tbest[pid] < 0 || f.zw < zbest[pid]) { zbest[pid] = f.zw; tbest[pid] = t; }
                    }
                continue;
            }
            int64_t X[3], Y[3];
            const real *pp[3] = {p0, p1, p2};
            int bad = 0;
            for (int k = 0; k < 3; k++) {
                real nx = pp[k][0] / pp[k][3], ny = pp[k][1] / pp[k][3];
                if (!(fabs((double)nx) < 1e4 && fabs((double)ny) < 1e4)) { bad = 1; break; }
                X[k] = snap(nx, W); Y[k] = snap(ny, H);
            }
            if (bad) continue;
            int64_t area = (X[1] - X[0]) * (Y[2] - Y[0]) - (Y[1] - Y[0]) * (X[2] - X[0]);
            if (area == 0) continue;
            int64_t s = area > 0 ? 1 : -1;
            int64_t xmin = X[0] < X[1] ? X[0] : X[1]; if (X[2] < xmin) xmin = X[2];
            int64_t xmax = X[0] > X[1] ? X[0] : X[1]; if (X[2] > xmax) xmax = X[2];
            int64_t ymin = Y[0] < Y[1] ? Y[0] : Y[1]; if (Y[2] < ymin) ymin = Y[2];
            int64_t ymax = Y[0] > Y[1] ? Y[0] : Y[1]; if (Y[2] > ymax) ymax = Y[2];
            /* pixel centre c = 16*px + 8 */
            int64_t px0 = (xmin - 8 + 15) >> 4, px1 = (xmax - 8) >> 4, py0 = (ymin - 8 + 15) >> 4, py1 = (ymax - 8) >> 4;
            if (px0 < 0) px0 = 0;
            if (py0 < 0) py0 = 0;
            if (px1 > W - 1) px1 = W - 1;
            if (py1 > H - 1) py1 = H - 1;
            for (int64_t py = py0; py <= py1; py++)
                for (int64_t px = px0; px <= px1; px++) {
                    int64_t cx = 16 * px + 8, cy = 16 * py + 8;
                    int inside = 1;
                    for (int k = 0; k < 3 && inside; k++) {
                        int a = k, c = (k + 1) % 3;
                        int64_t dx = (X[c] - X[a]) * s, dy = (Y[c] - Y[a]) * s;
                        int64_t e = dx * (cy - Y[a]) - dy * (cx - X[a]);
                        if (e < 0 || (e == 0 && !edge_owns_tie(dx, dy))) inside = 0;
                    }
                    if (!inside) continue;
                    frag_t f;
                    shade(p0, p1, p2, xs * ((real)px + (real)0.5) - (real)1, ys * ((real)py + (real)0.5) - (real)1, xs, ys, &f);
                    if (!(f.zw >= -1 && f.zw <= 1)) continue;   /* per-pixel near/far clip */
                    size_t pid = (size_t)py * W + px;
                    if (prev) {
                        const real *pr = prev + ((size_t)b * P + pid) * 4;
                        if (pr[3] == 0 || !(f.zw > pr[2])) continue;
                    }
                    if (tbest[pid] < 0 || f.zw < zbest[pid]) { zbest[pid] = f.zw; tbest[pid] = t; }
                }
        }
        for (int py = 0; py < H; py++)
            for (int px = 0; px < W; px++) {
                size_t pid = (size_t)py * W + px, o = ((size_t)b * P + pid) * 4;
                int t = tbest[pid];
                if (t < 0) { for (int k = 0; k < 4; k++) { rast[o + k] = 0; if (rast_db) rast_db[o + k] = 0; } continue; }
                const real *p0 = pb + 4 * tri[3 * t], *p1 = pb + 4 * tri[3 * t + 1], *p2 = pb + 4 * tri[3 * t + 2];
                frag_t f;
                shade(p0, p1, p2, xs * ((real)px + (real)0.5) - (real)1, ys * ((real)py + (real)0.5) - (real)1, xs, ys, &f);
                rast[o] = clampr(f.b0, 0, 1); rast[o + 1] = clampr(f.b1, 0, 1); rast[o + 2] = clampr(f.zw, -1, 1); rast[o + 3] = (real)(t + 1);
                if (rast_db) { rast_db[o] = f.dudx; rast_db[o + 1] = f.dudy; rast_db[o + 2] = f.dvdx; rast_db[o + 3] = f.dvdy; }
            }
    }
    free(zbest); free(tbest);
}
void mesh_rasterize_fwd(const real *pos, const int32_t *tri, int B, int V, int T, int H, int W, real *rast, real *rast_db) {
    rasterize_impl(pos, tri, B, V, T, H, W, NULL, rast, rast_db);
}
void mesh_rasterize_peel_fwd(const real *pos, const int32_t *tri, int B, int V, int T, int H, int W, const real *prev_rast, real *rast, real *rast_db) {
    rasterize_impl(pos, tri, B, V, T, H, W, prev_rast, rast, rast_db);
}

/* dL/dpos from dL/d(u,v) (dy[...,0:2]); dpos [B,V,4] must be zero-initialised */
/* ddb (optional, [B,H,W,4]): gradient w.r.t. rast_db = (du/dX, du/dY, dv/dX, dv/dY) (the dependency's grad_db path).  With A_kx = d a_k / d fx etc. as in the forward
 * pass: du/dX = xs iw (b0 Tx - A0x), ... ; the chain rule below goes through b0, b1, iw and the six A terms (functions of the un-shifted x, y, w). */
void mesh_rasterize_bwd(const real *pos, const int32_t *tri, const real *rast, const real *dy, const real *ddb, int B, int V, int T, int H, int W, real *dpos) {
    (void)T;
    const real xs = (real)2 / W, ys = (real)2 / H;
    size_t P = (size_t)H * W;
    for (int b = 0; b < B; b++)
        for (int py = 0; py < H; py++)
            for (int px = 0; px < W; px++) {
                size_t o = ((size_t)b * P + (size_t)py * W + px) * 4;
                int t = (int)rast[o + 3] - 1;
                if (t < 0) continue;
                real g0 = dy ? dy[o] : 0, g1 = dy ? dy[o + 1] : 0;
                real G[4] = {0, 0, 0, 0};
                if (ddb) for (int k = 0; k < 4; k++) G[k] = ddb[o + k];
                if (g0 == 0 && g1 == 0 && G[0] == 0 && G[1] == 0 && G[2] == 0 && G[3] == 0) continue;
                int vi[3] = {tri[3 * t], tri[3 * t + 1], tri[3 * t + 2]};
                const real *pb = pos + (size_t)b * V * 4;
                const real *p0 = pb + 4 * vi[0], *p1 = pb + 4 * vi[1], *p2 = pb + 4 * vi[2];
                real fx = xs * ((real)px + (real)0.5) - (real)1, fy = ys * ((real)py + (real)0.5) - (real)1;
                real p0x = p0[0] - fx * p0[3], p0y = p0[1] - fy * p0[3];
                real p1x = p1[0] - fx * p1[3], p1y = p1[1] - fy * p1[3];
                real p2x = p2[0] - fx * p2[3], p2y = p2[1] - fy * p2[3];
                real a0 = p1x * p2y - p1y * p2x, a1 = p2x * p0y - p2y * p0x, a2 = p0x * p1y - p0y * p1x;
                real iw = (real)1 / (a0 + a1 + a2), b0 = a0 * iw, b1 = a1 * iw;
                if (b0 < 0 || b0 > 1) g0 = 0;   /* the forward pass clamps u, v to [0,1]: no gradient through a clamped value */
                if (b1 < 0 || b1 > 1) g1 = 0;
                real da0 = (g0 * ((real)1 - b0) - g1 * b1) * iw;
                real da1 = (-g0 * b0 + g1 * ((real)1 - b1)) * iw;
                real da2 = (-g0 * b0 - g1 * b1) * iw;
                real dA[3][2] = {{0, 0}, {0, 0}, {0, 0}};          /* gradients of A_kx, A_ky */
                if (ddb) {
                    real A0x = p1[3] * p2[1] - p1[1] * p2[3], A0y = p1[0] * p2[3] - p2[0] * p1[3];
                    real A1x = p0[1] * p2[3] - p2[1] * p0[3], A1y = p2[0] * p0[3] - p0[0] * p2[3];
                    real A2x = p1[1] * p0[3] - p0[1] * p1[3], A2y = p0[0] * p1[3] - p1[0] * p0[3];
                    real Tx = A0x + A1x + A2x, Ty = A0y + A1y + A2y, al = xs * iw, be = ys * iw;
                    real dudx = al * (b0 * Tx - A0x), dudy = be * (b0 * Ty - A0y), dvdx = al * (b1 * Tx - A1x), dvdy = be * (b1 * Ty - A1y);
                    real gb0 = G[0] * al * Tx + G[1] * be * Ty, gb1 = G[2] * al * Tx + G[3] * be * Ty;
                    real gTx = al * (G[0] * b0 + G[2] * b1), gTy = be * (G[1] * b0 + G[3] * b1);
                    real L = G[0] * dudx + G[1] * dudy + G[2] * dvdx + G[3] * dvdy;
                    real gi = L / iw + gb0 * a0 + gb1 * a1, gS = -iw * iw * gi;
                    da0 += gb0 * iw + gS; da1 += gb1 * iw + gS; da2 += gS;
                    dA[0][0] = gTx - al * G[0]; dA[0][1] = gTy - be * G[1];
                    dA[1][0] = gTx - al * G[2]; dA[1][1] = gTy - be * G[3];
                    dA[2][0] = gTx; dA[2][1] = gTy;
                }
                /* a0 = p1x p2y - p1y p2x ; a1 = p2x p0y - p2y p0x ; a2 = p0x p1y - p0y p1x */
                real d0x = da1 * (-p2y) + da2 * p1y, d0y = da1 * p2x + da2 * (-p1x);
                real d1x = da0 * p2y + da2 * (-p0y), d1y = da0 * (-p2x) + da2 * p0x;
                real d2x = da0 * (-p1y) + da1 * p0y, d2y = da0 * p1x + da1 * (-p0x);
                real dxs[3] = {d0x, d1x, d2x}, dys[3] = {d0y, d1y, d2y};
                for (int k = 0; k < 3; k++) {
                    real *d = dpos + ((size_t)b * V + vi[k]) * 4;
                    d[0] += dxs[k]; d[1] += dys[k]; d[3] += -fx * dxs[k] - fy * dys[k];
                }
                if (ddb) {      /* A_kx = y_m w_l - y_l w_m, A_ky = x_l w_m - x_m w_l with (l, m) = (k+1, k+2) mod 3 */
                    const real *pp[3] = {p0, p1, p2};
                    for (int k = 0; k < 3; k++) {
                        int l = (k + 1) % 3, m = (k + 2) % 3;
                        real *dl = dpos + ((size_t)b * V + vi[l]) * 4, *dm = dpos + ((size_t)b * V + vi[m]) * 4;
                        real gx = dA[k][0], gy = dA[k][1];
                        dm[1] += gx * pp[l][3]; dl[3] += gx * pp[m][1]; dl[1] -= gx * pp[m][3]; dm[3] -= gx * pp[l][1];
                        dl[0] += gy * pp[m][3]; dm[3] += gy * pp[l][0]; dm[0] -= gy * pp[l][3]; dl[3] -= gy * pp[m][0];
                    }
                }
            }
}

/* ------------------------------------------------------------------ interpolate */
/* attr [Ba,V,A] with Ba in {1,B}; diff list: nd indices into [0,A) (nd = 0 -> no out_da) */
void mesh_interpolate_fwd(const real *attr, int Ba, const real *rast, const int32_t *tri, const real *rast_db, const int32_t *diff, int nd,
                          int B, int V, int A, int H, int W, real *out, real *out_da) {
    size_t P = (size_t)H * W;
    for (int b = 0; b < B; b++)
        for (size_t pid = 0; pid < P; pid++) {
            size_t o = (size_t)b * P + pid;
            int t = (int)rast[4 * o + 3] - 1;
            real *po = out + o * A;
            if (t < 0) { for (int a = 0; a < A; a++) po[a] = 0; if (nd) for (int a = 0; a < 2 * nd; a++) out_da[o * 2 * nd + a] = 0; continue; }
            const real *ab = attr + (size_t)(Ba > 1 ? b : 0) * V * A;
            const real *a0 = ab + (size_t)tri[3 * t] * A, *a1 = ab + (size_t)tri[3 * t + 1] * A, *a2 = ab + (size_t)tri[3 * t + 2] * A;
            real u = rast[4 * o], v = rast[4 * o + 1], w2 = (real)1 - u - v;
            for (int a = 0; a < A; a++) po[a] = u * a0[a] + v * a1[a] + w2 * a2[a];
            if (nd) {
                real dudx = rast_db[4 * o], dudy = rast_db[4 * o + 1], dvdx = rast_db[4 * o + 2], dvdy = rast_db[4 * o + 3];
                for (int k = 0; k < nd; k++) {
                    int a = diff[k];
                    real dsdu = a0[a] - a2[a], dsdv = a1[a] - a2[a];
                    out_da[(o * nd + k) * 2] = dsdu * dudx + dsdv * dvdx;
                    out_da[(o * nd + k) * 2 + 1] = dsdu * dudy + dsdv * dvdy;
                }
            }
        }
}
/* dattr [Ba,V,A] and drast [B,H,W,4] zero-initialised by the caller */
/* backward of the pixel differentials interpolate() returns: out_da[k] = (ds/du * du/dX + ds/dv * dv/dX, ds/du * du/dY + ds/dv * dv/dY) with
 * ds/du = a0 - a2, ds/dv = a1 - a2 for attribute diff[k].  dout_da [B,H,W,2 nd] -> dattr [Ba,V,A] (accumulated), drast_db [B,H,W,4] (written in full). */
void mesh_interpolate_da_bwd(const real *attr, int Ba, const real *rast, const int32_t *tri, const real *rast_db, const int32_t *diff, int nd,
                             const real *dout_da, int B, int V, int A, int H, int W, real *dattr, real *drast_db) {
    size_t P = (size_t)H * W;
    for (int b = 0; b < B; b++) for (size_t pid = 0; pid < P; pid++) {
        size_t o = (size_t)b * P + pid;
        real *gdb = drast_db + 4 * o;
        gdb[0] = gdb[1] = gdb[2] = gdb[3] = 0;
        int t = (int)rast[4 * o + 3] - 1;
        if (t < 0) continue;
        size_t ab = (size_t)(Ba > 1 ? b : 0) * V * A;
        const int32_t *vi = tri + 3 * (size_t)t;
        const real *db = rast_db + 4 * o;
        for (int k = 0; k < nd; k++) {
            int a = diff[k];
            real gx = dout_da[(o * nd + k) * 2], gy = dout_da[(o * nd + k) * 2 + 1];
            real dsdu = attr[ab + (size_t)vi[0] * A + a] - attr[ab + (size_t)vi[2] * A + a], dsdv = attr[ab + (size_t)vi[1] * A + a] - attr[ab + (size_t)vi[2] * A + a];
            real g_dsdu = gx * db[0] + gy * db[1], g_dsdv = gx * db[2] + gy * db[3];
            dattr[ab + (size_t)vi[0] * A + a] += g_dsdu; dattr[ab + (size_t)vi[1] * A + a] += g_dsdv; dattr[ab + (size_t)vi[2] * A + a] -= g_dsdu + g_dsdv;
            gdb[0] += gx * dsdu; gdb[1] += gy * dsdu; gdb[2] += gx * dsdv; gdb[3] += gy * dsdv;
        }
    }
}
void mesh_interpolate_bwd(const real *attr, int Ba, const real *rast, const int32_t *tri, const real *dy, int B, int V, int A, int H, int W,
                          real *dattr, real *drast) {
    size_t P = (size_t)H * W;
    for (int b = 0; b < B; b++)
        for (size_t pid = 0; pid < P; pid++) {
            size_t o = (size_t)b * P + pid;
            int t = (int)rast[4 * o + 3] - 1;
            if (t < 0) continue;
            size_t ab = (size_t)(Ba > 1 ? b : 0) * V * A;
            size_t i0 = ab + (size_t)tri[3 * t] * A, i1 = ab + (size_t)tri[3 * t + 1] * A, i2 = ab + (size_t)tri[3 * t + 2] * A;
            real u = rast[4 * o], v = rast[4 * o + 1], w2 = (real)1 - u - v;
            real gu = 0, gv = 0;
            for (int a = 0; a < A; a++) {
                real g = dy[o * A + a];
                dattr[i0 + a] += u * g; dattr[i1 + a] += v * g; dattr[i2 + a] += w2 * g;
                gu += g * (attr[i0 + a] - attr[i2 + a]); gv += g * (attr[i1 + a] - attr[i2 + a]);
            }
            drast[4 * o] += gu; drast[4 * o + 1] += gv;
        }
}

/* ------------------------------------------------------------------ texture */
static int wrapi(int i, int n, int boundary) {   /* boundary: 0 wrap, 1 clamp, 2 zero (-1 = a texel of the all-zero extension) */
    if (boundary == 0) { i %= n; if (i < 0) i += n; return i; }
    if (boundary == 2) return (i < 0 || i >= n) ? -1 : i;
    return i < 0 ? 0 : (i >= n ? n - 1 : i);
}
/* texel (iv, iu) channel c of one texture, 0 outside (boundary mode 'zero' hands out index -1) */
static real texel(const real *tb, int iv, int iu, int Wt, int C, int c) { return (iv < 0 || iu < 0) ? (real)0 : tb[((size_t)iv * Wt + iu) * C + c]; }
/* tex [Bt,Ht,Wt,C] with Bt in {1,B}; filter: 0 nearest, 1 linear */
void mesh_texture_fwd(const real *tex, int Bt, const real *uv, int B, int H, int W, int Ht, int Wt, int C, int filter, int boundary, real *out) {
    size_t P = (size_t)H * W;
    for (int b = 0; b < B; b++) {
        const real *tb = tex + (size_t)(Bt > 1 ? b : 0) * Ht * Wt * C;
        for (size_t pid = 0; pid < P; pid++) {
            size_t o = (size_t)b * P + pid;
            real u = uv[2 * o] * Wt, v = uv[2 * o + 1] * Ht;
            real *po = out + o * C;
            if (filter == 0) {
                int iu = wrapi((int)floor((double)u), Wt, boundary), iv = wrapi((int)floor((double)v), Ht, boundary);
                for (int c = 0; c < C; c++) po[c] = texel(tb, iv, iu, Wt, C, c);
            } else {
                u -= (real)0.5; v -= (real)0.5;
                real fu0 = (real)floor((double)u), fv0 = (real)floor((double)v);
                real fu = u - fu0, fv = v - fv0;
                int iu0 = wrapi((int)fu0, Wt, boundary), iu1 = wrapi((int)fu0 + 1, Wt, boundary);
                int iv0 = wrapi((int)fv0, Ht, boundary), iv1 = wrapi((int)fv0 + 1, Ht, boundary);
                for (int c = 0; c < C; c++) {
                    real t00 = texel(tb, iv0, iu0, Wt, C, c), t10 = texel(tb, iv0, iu1, Wt, C, c);
                    real t01 = texel(tb, iv1, iu0, Wt, C, c), t11 = texel(tb, iv1, iu1, Wt, C, c);
                    real top = t00 + fu * (t10 - t00), bot = t01 + fu * (t11 - t01);
                    po[c] = top + fv * (bot - top);
                }
            }
        }
    }
}
/* dtex [Bt,Ht,Wt,C], duv [B,H,W,2] zero-initialised */
void mesh_texture_bwd(const real *tex, int Bt, const real *uv, const real *dy, int B, int H, int W, int Ht, int Wt, int C, int filter, int boundary,
                      real *dtex, real *duv) {
    size_t P = (size_t)H * W;
    for (int b = 0; b < B; b++) {
        size_t tbo = (size_t)(Bt > 1 ? b : 0) * Ht * Wt * C;
        for (size_t pid = 0; pid < P; pid++) {
            size_t o = (size_t)b * P + pid;
            real u = uv[2 * o] * Wt, v = uv[2 * o + 1] * Ht;
            const real *g = dy + o * C;
            if (filter == 0) {
                int iu = wrapi((int)floor((double)u), Wt, boundary), iv = wrapi((int)floor((double)v), Ht, boundary);
                if (iv >= 0 && iu >= 0) for (int c = 0; c < C; c++) dtex[tbo + ((size_t)iv * Wt + iu) * C + c] += g[c];
            } else {
                u -= (real)0.5; v -= (real)0.5;
                real fu0 = (real)floor((double)u), fv0 = (real)floor((double)v);
                real fu = u - fu0, fv = v - fv0;
                int iu0 = wrapi((int)fu0, Wt, boundary), iu1 = wrapi((int)fu0 + 1, Wt, boundary);
                int iv0 = wrapi((int)fv0, Ht, boundary), iv1 = wrapi((int)fv0 + 1, Ht, boundary);
                real gu = 0, gv = 0;
                for (int c = 0; c < C; c++) {
                    const real *tbb = tex + tbo;
                    real *dtb = dtex + tbo;
                    real t00 = texel(tbb, iv0, iu0, Wt, C, c), t10 = texel(tbb, iv0, iu1, Wt, C, c);
                    real t01 = texel(tbb, iv1, iu0, Wt, C, c), t11 = texel(tbb, iv1, iu1, Wt, C, c);
                    if (iv0 >= 0 && iu0 >= 0) dtb[((size_t)iv0 * Wt + iu0) * C + c] += g[c] * ((real)1 - fu) * ((real)1 - fv);
                    if (iv0 >= 0 && iu1 >= 0) dtb[((size_t)iv0 * Wt + iu1) * C + c] += g[c] * fu * ((real)1 - fv);
                    if (iv1 >= 0 && iu0 >= 0) dtb[((size_t)iv1 * Wt + iu0) * C + c] += g[c] * ((real)1 - fu) * fv;
                    if (iv1 >= 0 && iu1 >= 0) dtb[((size_t)iv1 * Wt + iu1) * C + c] += g[c] * fu * fv;
                    gu += g[c] * ((t10 - t00) * ((real)1 - fv) + (t11 - t01) * fv);
                    gv += g[c] * ((t01 - t00) * ((real)1 - fu) + (t11 - t10) * fu);
                }
                duv[2 * o] += gu * Wt; duv[2 * o + 1] += gv * Ht;
            }
        }
    }
}

/* ------------------------------------------------------------------ mip-mapped texture
 * Restates the dependency's mip-mapped filter modes as its documentation and public source describe them (nvdiffrast 0.3.3
 * `texture(tex, uv, uv_da, mip_level_bias, mip, filter_mode='linear-mipmap-linear'|'linear-mipmap-nearest')`; reference call sites that
 * reach them through filter_mode='auto' + uv_da: Gen_3D_Modules/LGM/nerf_marching_cubes_converter.py:232,
 * Gen_3D_Modules/TRELLIS/trellis/utils/postprocessing_utils.py:384, Gen_3D_Modules/Stable3DGen/trellis/utils/_rasterization.py:88).
 *   pyramid: level l+1 = 2x2 box average of level l; an extent that has reached 1 stays 1 (2x1 / 1x2 averages); levels until 1x1 or
 *            max_mip_level.  `stack` holds levels 1..L back to back per batch item: [Bt][sum_l h_l*w_l][C].
 *   level:   with uv_da = (du/dX, du/dY, dv/dX, dv/dY): J = [[du/dX*Wt, du/dY*Wt],[dv/dX*Ht, dv/dY*Ht]], lambda = largest eigenvalue of
 *            J J^T (squared major axis of the pixel footprint in texels), level = log2(lambda)/2 (+ bias); without uv_da, level = bias.
 *            Clamped to [0, L]; NaN -> 0.  'linear-mipmap-linear' blends floor(level) and the next level by the fraction;
 *            'linear-mipmap-nearest' samples level floor(level + 0.5).  Each level is sampled bilinearly (texel centres at half integers).
 * Gradients: texels of both levels (dtex for level 0, dstack for levels >= 1; the pyramid's own backward folds dstack down), uv.
 * uv_da / bias receive no gradient (as for rast_db / out_da above: no consumer). */
#define MIP_MAX 16
typedef struct { int L; int w[MIP_MAX + 1], h[MIP_MAX + 1]; long long off[MIP_MAX + 1]; long long total; } mip_t;
/* returns 0, or -1 when an extent > 1 is odd at a level that still has to be halved */
static int mip_info(int Ht, int Wt, int max_level, mip_t *m) {
    m->L = 0; m->w[0] = Wt; m->h[0] = Ht; m->off[0] = 0; m->total = 0;
    int w = Wt, h = Ht;
    while ((w > 1 || h > 1) && m->L < MIP_MAX && (max_level < 0 || m->L < max_level)) {
        if ((w > 1 && (w & 1)) || (h > 1 && (h & 1))) return -1;
        if (w > 1) w >>= 1;
        if (h > 1) h >>= 1;
        m->L++; m->w[m->L] = w; m->h[m->L] = h; m->off[m->L] = m->total; m->total += (long long)w * h;
    }
    return 0;
}
/* levels_hw [2*(MIP_MAX+1)] receives (h, w) per level incl. the base; returns L (or -1), *stack_texels = texels of levels 1..L */
int mesh_mip_info(int Ht, int Wt, int max_level, int *levels_hw, long long *stack_texels) {
    mip_t m; if (mip_info(Ht, Wt, max_level, &m)) return -1;
    for (int l = 0; l <= m.L; l++) { levels_hw[2 * l] = m.h[l]; levels_hw[2 * l + 1] = m.w[l]; }
    *stack_texels = m.total;
    return m.L;
}
static const real *mip_level_c(const real *tex, const real *stack, const mip_t *m, int l, int bt, int C) {
    return l == 0 ? tex + (size_t)bt * m->h[0] * m->w[0] * C : stack + ((size_t)bt * m->total + m->off[l]) * C;
}
static real *mip_level(real *tex, real *stack, const mip_t *m, int l, int bt, int C) {
    return l == 0 ? tex + (size_t)bt * m->h[0] * m->w[0] * C : stack + ((size_t)bt * m->total + m->off[l]) * C;
}
int mesh_mip_build(const real *tex, int Bt, int Ht, int Wt, int C, int max_level, real *stack) {
    mip_t m; if (mip_info(Ht, Wt, max_level, &m)) return -1;
    for (int b = 0; b < Bt; b++)
        for (int l = 1; l <= m.L; l++) {
            const real *src = mip_level_c(tex, stack, &m, l - 1, b, C);
            real *dst = mip_level((real *)tex, stack, &m, l, b, C);
            int sx = m.w[l - 1] > 1 ? 2 : 1, sy = m.h[l - 1] > 1 ? 2 : 1, ws = m.w[l - 1];
            for (int y = 0; y < m.h[l]; y++) for (int x = 0; x < m.w[l]; x++) for (int c = 0; c < C; c++) {
                real a = 0;
                for (int j = 0; j < sy; j++) for (int i = 0; i < sx; i++) a += src[((size_t)(y * sy + j) * ws + (x * sx + i)) * C + c];
                dst[((size_t)y * m.w[l] + x) * C + c] = a / (real)(sx * sy);
            }
        }
    return 0;
}
/* dtex [Bt,Ht,Wt,C] += the pyramid's transposed filters applied to dstack (dstack is used as scratch and modified) */
int mesh_mip_build_bwd(real *dstack, int Bt, int Ht, int Wt, int C, int max_level, real *dtex) {
    mip_t m; if (mip_info(Ht, Wt, max_level, &m)) return -1;
    for (int b = 0; b < Bt; b++)
        for (int l = m.L; l >= 1; l--) {
            const real *src = mip_level_c(dtex, dstack, &m, l, b, C);
            real *dst = mip_level(dtex, dstack, &m, l - 1, b, C);
            int sx = m.w[l - 1] > 1 ? 2 : 1, sy = m.h[l - 1] > 1 ? 2 : 1, wd = m.w[l - 1];
            for (int y = 0; y < m.h[l - 1]; y++) for (int x = 0; x < wd; x++) for (int c = 0; c < C; c++)
                dst[((size_t)y * wd + x) * C + c] += src[((size_t)(y / sy) * m.w[l] + (x / sx)) * C + c] / (real)(sx * sy);
        }
    return 0;
}
/* level selection: returns level0, *level1, *frac */
static int mip_select(const real *da, const real *bias, int Ht, int Wt, int L, int filter, int *level1, real *frac) {
    real fl = 0;
    if (da) {
        real dsdx = da[0] * Wt, dsdy = da[1] * Wt, dtdx = da[2] * Ht, dtdy = da[3] * Ht;
        real A = dsdx * dsdx + dtdx * dtdx, Bq = dsdy * dsdy + dtdy * dtdy, Cq = dsdx * dsdy + dtdx * dtdy;
        real l2b = (real)0.5 * (A + Bq), l2n = (real)0.25 * (A - Bq) * (A - Bq) + Cq * Cq;
        real major = l2b + (real)sqrt((double)l2n);
        fl = (real)0.5 * (real)log2((double)major);
    }
    if (bias) fl += *bias;
    if (!(fl > 0)) fl = 0;                /* also NaN and -inf */
    if (fl > (real)L) fl = (real)L;
    int l0;
    if (filter == 2) { l0 = (int)floor((double)fl + 0.5); if (l0 > L) l0 = L; *level1 = l0; *frac = 0; return l0; }
    l0 = (int)floor((double)fl); if (l0 > L) l0 = L;
    *level1 = l0 + 1 > L ? L : l0 + 1;
    *frac = fl - (real)l0;
    return l0;
}
/* filter: 2 = linear-mipmap-nearest, 3 = linear-mipmap-linear.  uv_da [B,H,W,4] or NULL, bias [B,H,W] or NULL (not both NULL). */
int mesh_texture_mip_fwd(const real *tex, const real *stack, int Bt, const real *uv, const real *uv_da, const real *bias, int B, int H, int W,
                         int Ht, int Wt, int C, int filter, int boundary, int max_level, real *out) {
    if (boundary < 0 || boundary > 2) return -1;                     /* 2 = 'zero': a tap outside THIS level's texels reads 0 */
    mip_t m; if (mip_info(Ht, Wt, max_level, &m)) return -1;
    size_t P = (size_t)H * W;
    for (int b = 0; b < B; b++) for (size_t pid = 0; pid < P; pid++) {
        size_t o = (size_t)b * P + pid;
        int l1; real f;
        int l0 = mip_select(uv_da ? uv_da + 4 * o : NULL, bias ? bias + o : NULL, Ht, Wt, m.L, filter, &l1, &f);
        real *po = out + o * C;
        for (int c = 0; c < C; c++) po[c] = 0;
        for (int k = 0; k < 2; k++) {
            int l = k ? l1 : l0; real wl = k ? f : (real)1 - f;
            if (k && (l1 == l0 || f == 0)) break;
            const real *tb = mip_level_c(tex, stack, &m, l, Bt > 1 ? b : 0, C);
            int wl_ = m.w[l], hl_ = m.h[l];
            real u = uv[2 * o] * wl_ - (real)0.5, v = uv[2 * o + 1] * hl_ - (real)0.5;
            real fu0 = (real)floor((double)u), fv0 = (real)floor((double)v), fu = u - fu0, fv = v - fv0;
            int iu0 = wrapi((int)fu0, wl_, boundary), iu1 = wrapi((int)fu0 + 1, wl_, boundary);
            int iv0 = wrapi((int)fv0, hl_, boundary), iv1 = wrapi((int)fv0 + 1, hl_, boundary);
#define MIP_TX(iv, iu) (((iv) < 0 || (iu) < 0) ? (real)0 : tb[((size_t)(iv) * wl_ + (iu)) * C + c])
            for (int c = 0; c < C; c++) {
                real t00 = MIP_TX(iv0, iu0), t10 = MIP_TX(iv0, iu1), t01 = MIP_TX(iv1, iu0), t11 = MIP_TX(iv1, iu1);
                real top = t00 + fu * (t10 - t00), bot = t01 + fu * (t11 - t01);
                po[c] += wl * (top + fv * (bot - top));
            }
        }
    }
    return 0;
}
/* d(level)/d(uv_da) of mip_select where the level is strictly inside (0, L) before clamping: level = log2(major)/2 + bias with major the larger
 * eigenvalue of J J^T.  gda[4] receives dlevel times that derivative; returns 1 when the level is inside (the bias then receives dlevel itself). */
static int mip_level_grad(const real *da, const real *bias, int Ht, int Wt, int L, real dlevel, real *gda) {
    real fl = 0, major = 1, A = 0, Bq = 0, Cq = 0, s = 0, dsdx = 0, dsdy = 0, dtdx = 0, dtdy = 0;
    gda[0] = gda[1] = gda[2] = gda[3] = 0;
    if (da) {
        dsdx = da[0] * Wt; dsdy = da[1] * Wt; dtdx = da[2] * Ht; dtdy = da[3] * Ht;
        A = dsdx * dsdx + dtdx * dtdx; Bq = dsdy * dsdy + dtdy * dtdy; Cq = dsdx * dsdy + dtdx * dtdy;
        s = (real)sqrt((double)((real)0.25 * (A - Bq) * (A - Bq) + Cq * Cq));
        major = (real)0.5 * (A + Bq) + s;
        fl = (real)0.5 * (real)log2((double)major);
    }
    if (bias) fl += *bias;
    if (!(fl > 0) || !(fl < (real)L)) return 0;            /* clamped (or NaN): the level does not move */
    if (da) {
        real gm = dlevel * (real)0.5 / (major * (real)0.6931471805599453);       /* d level / d major */
        real t = s > 0 ? (real)0.25 * (A - Bq) / s : 0, gA = gm * ((real)0.5 + t), gB = gm * ((real)0.5 - t), gC = s > 0 ? gm * Cq / s : 0;
        gda[0] = ((real)2 * dsdx * gA + dsdy * gC) * Wt; gda[1] = ((real)2 * dsdy * gB + dsdx * gC) * Wt;
        gda[2] = ((real)2 * dtdx * gA + dtdy * gC) * Ht; gda[3] = ((real)2 * dtdy * gB + dtdx * gC) * Ht;
    }
    return 1;
}
/* dtex [Bt,Ht,Wt,C], dstack [Bt,total,C], duv [B,H,W,2]: zero-initialised by the caller.  dda [B,H,W,4] / dbias [B,H,W] (optional, written in full):
 * gradients w.r.t. uv_da and mip_level_bias -- 'linear-mipmap-linear' blends two levels by the fraction of the level, so d out / d level =
 * sample(level1) - sample(level0) wherever the two levels differ and the level is not clamped; zero for 'linear-mipmap-nearest'. */
int mesh_texture_mip_bwd(const real *tex, const real *stack, int Bt, const real *uv, const real *uv_da, const real *bias, const real *dy,
                         int B, int H, int W, int Ht, int Wt, int C, int filter, int boundary, int max_level, real *dtex, real *dstack, real *duv,
                         real *dda, real *dbias) {
    if (boundary < 0 || boundary > 2) return -1;
    mip_t m; if (mip_info(Ht, Wt, max_level, &m)) return -1;
    size_t P = (size_t)H * W;
    for (int b = 0; b < B; b++) for (size_t pid = 0; pid < P; pid++) {
        size_t o = (size_t)b * P + pid;
        int l1; real f;
        int l0 = mip_select(uv_da ? uv_da + 4 * o : NULL, bias ? bias + o : NULL, Ht, Wt, m.L, filter, &l1, &f);
        const real *g = dy + o * C;
        real sdot[2] = {0, 0};                                      /* dy . sample(level k) */
        int nlev = 1;
        for (int k = 0; k < 2; k++) {
            int l = k ? l1 : l0; real wl = k ? f : (real)1 - f;
            if (k && (l1 == l0 || f == 0)) break;
            nlev = k + 1;
            int bt = Bt > 1 ? b : 0;
            const real *tb = mip_level_c(tex, stack, &m, l, bt, C);
            real *db = mip_level(dtex, dstack, &m, l, bt, C);
            int wl_ = m.w[l], hl_ = m.h[l];
            real u = uv[2 * o] * wl_ - (real)0.5, v = uv[2 * o + 1] * hl_ - (real)0.5;
            real fu0 = (real)floor((double)u), fv0 = (real)floor((double)v), fu = u - fu0, fv = v - fv0;
            int iu0 = wrapi((int)fu0, wl_, boundary), iu1 = wrapi((int)fu0 + 1, wl_, boundary);
            int iv0 = wrapi((int)fv0, hl_, boundary), iv1 = wrapi((int)fv0 + 1, hl_, boundary);
            real gu = 0, gv = 0;
            int ok00 = iv0 >= 0 && iu0 >= 0, ok10 = iv0 >= 0 && iu1 >= 0, ok01 = iv1 >= 0 && iu0 >= 0, ok11 = iv1 >= 0 && iu1 >= 0;      /* 'zero': taps outside the level */
            for (int c = 0; c < C; c++) {
                size_t i00 = ((size_t)iv0 * wl_ + iu0) * C + c, i10 = ((size_t)iv0 * wl_ + iu1) * C + c;
                size_t i01 = ((size_t)iv1 * wl_ + iu0) * C + c, i11 = ((size_t)iv1 * wl_ + iu1) * C + c;
                real gc = g[c] * wl;
                if (ok00) db[i00] += gc * ((real)1 - fu) * ((real)1 - fv);
                if (ok10) db[i10] += gc * fu * ((real)1 - fv);
                if (ok01) db[i01] += gc * ((real)1 - fu) * fv;
                if (ok11) db[i11] += gc * fu * fv;
                real t00 = ok00 ? tb[i00] : 0, t10 = ok10 ? tb[i10] : 0, t01 = ok01 ? tb[i01] : 0, t11 = ok11 ? tb[i11] : 0;
                gu += gc * ((t10 - t00) * ((real)1 - fv) + (t11 - t01) * fv);
                gv += gc * ((t01 - t00) * ((real)1 - fu) + (t11 - t10) * fu);
                real top = t00 + fu * (t10 - t00), bot = t01 + fu * (t11 - t01);
                sdot[k] += g[c] * (top + fv * (bot - top));
            }
            duv[2 * o] += gu * wl_; duv[2 * o + 1] += gv * hl_;
        }
        if (dda || dbias) {
            real gda[4] = {0, 0, 0, 0};
            int inside = (filter == 3 && nlev == 2) ? mip_level_grad(uv_da ? uv_da + 4 * o : NULL, bias ? bias + o : NULL, Ht, Wt, m.L, sdot[1] - sdot[0], gda) : 0;
            if (dda) for (int i = 0; i < 4; i++) dda[4 * o + i] = inside ? gda[i] : 0;
            if (dbias) dbias[o] = inside ? sdot[1] - sdot[0] : 0;
        }
    }
    return 0;
}

/* ------------------------------------------------------------------ antialias */
typedef struct { int32_t va, vb, tri, opp; } edge_t;
static int edge_cmp(const void *x, const void *y) {
    const edge_t *a = (const edge_t *)x, *b = (const edge_t *)y;
    if (a->va != b->va) return a->va < b->va ? -1 : 1;
    if (a->vb != b->vb) return a->vb < b->vb ? -1 : 1;
    return a->tri < b->tri ? -1 : (a->tri > b->tri);
}
/* opposite vertex of the OTHER triangle on edge (va,vb) of triangle `tri`: -1 boundary, -2 more than two triangles */
static int other_opposite(const edge_t *edges, int ne, int va, int vb, int tri) {
    if (va > vb) { int t = va; va = vb; vb = t; }
    int lo = 0, hi = ne;
    while (lo < hi) { int mid = (lo + hi) / 2; if (edges[mid].va < va || (edges[mid].va == va && edges[mid].vb < vb)) lo = mid + 1; else hi = mid; }
    int cnt = 0, res = -1;
    for (int i = lo; i < ne && edges[i].va == va && edges[i].vb == vb; i++) { cnt++; if (edges[i].tri != tri && res == -1) res = edges[i].opp; }
    if (cnt > 2) return -2;
    return res;
}
static edge_t *build_edges(const int32_t *tri, int T, int *ne) {
    edge_t *e = (edge_t *)malloc(sizeof(edge_t) * 3 * (size_t)(T > 0 ? T : 1));
    int n = 0;
    for (int t = 0; t < T; t++)
        for (int k = 0; k < 3; k++) {
            int a = tri[3 * t + k], b = tri[3 * t + (k + 1) % 3], o = tri[3 * t + (k + 2) % 3];
            if (a == b) continue;
            e[n].va = a < b ? a : b; e[n].vb = a < b ? b : a; e[n].tri = t; e[n].opp = o; n++;
        }
    qsort(e, (size_t)n, sizeof(edge_t), edge_cmp);
    *ne = n;
    return e;
}

/* analysis of one pixel pair; returns 1 and fills (pa = pixel of the chosen triangle, pb = the other, va, vb, s) when a silhouette
 * edge of the nearer triangle crosses the centre-to-centre segment at parameter s in [0,1] (measured from pa towards pb) */
typedef struct { int ax, ay, bx, by, va, vb; real s; int d; real sgn; } aa_hit_t;
static int aa_analyze(const real *pb_, const int32_t *tri, const edge_t *edges, int ne, const real *rast, int H, int W, int b, int px, int py, int d, aa_hit_t *hit) {
    int qx = px + (d == 0), qy = py + (d == 1);
    if (qx >= W || qy >= H) return 0;
    size_t P = (size_t)H * W;
    size_t o0 = ((size_t)b * P + (size_t)py * W + px) * 4, o1 = ((size_t)b * P + (size_t)qy * W + qx) * 4;
    int id0 = (int)rast[o0 + 3], id1 = (int)rast[o1 + 3];
    if (id0 == id1) return 0;
    int t;
    int a_is_p;
    if (id0 > 0 && id1 > 0) a_is_p = rast[o0 + 2] < rast[o1 + 2];
    else a_is_p = id0 > 0;
    t = (a_is_p ? id0 : id1) - 1;
    int ax = a_is_p ? px : qx, ay = a_is_p ? py : qy, bx = a_is_p ? qx : px, by = a_is_p ? qy : py;
    real sgn = a_is_p ? (real)1 : (real)-1;
    int vi[3] = {tri[3 * t], tri[3 * t + 1], tri[3 * t + 2]};
    real nx[3], ny[3];
    for (int k = 0; k < 3; k++) {
        const real *p = pb_ + 4 * (size_t)vi[k];
        if (!(p[3] > 0)) return 0;
        nx[k] = p[0] / p[3]; ny[k] = p[1] / p[3];
    }
    real cx = ((real)ax + (real)0.5) * ((real)2 / W) - (real)1, cy = ((real)ay + (real)0.5) * ((real)2 / H) - (real)1;
    real h = d == 0 ? (real)2 / W : (real)2 / H;
    int found = 0;
    real best = 0;
    for (int k = 0; k < 3; k++) {
        int ia = k, ib = (k + 1) % 3, io = (k + 2) % 3;
        int opp = other_opposite(edges, ne, vi[ia], vi[ib], t);
        if (opp == -2) continue;
        real ex = nx[ib] - nx[ia], ey = ny[ib] - ny[ia];
        if (opp >= 0) {   /* interior edge: silhouette only if both triangles lie on the same side of it */
            const real *q = pb_ + 4 * (size_t)opp;
            if (!(q[3] > 0)) continue;
            real ox = q[0] / q[3], oy = q[1] / q[3];
            real s_this = ex * (ny[io] - ny[ia]) - ey * (nx[io] - nx[ia]);
            real s_other = ex * (oy - ny[ia]) - ey * (ox - nx[ia]);
            if (!(s_this * s_other > 0)) continue;
        }
        /* crossing of the edge with the axis-aligned segment from A's centre towards B's centre */
        real s;
        if (d == 0) {
            real da = ny[ia] - cy, db = ny[ib] - cy;
            if (!((da <= 0 && db > 0) || (db <= 0 && da > 0))) continue;
            real te = da / (da - db);
            real xc = nx[ia] + te * ex;
            s = sgn * (xc - cx) / h;
        } else {
            real da = nx[ia] - cx, db = nx[ib] - cx;
            if (!((da <= 0 && db > 0) || (db <= 0 && da > 0))) continue;
            real te = da / (da - db);
            real yc = ny[ia] + te * ey;
            s = sgn * (yc - cy) / h;
        }
        if (!(s >= 0 && s <= 1)) continue;
        if (!found || s < best) { found = 1; best = s; hit->va = vi[ia]; hit->vb = vi[ib]; }
    }
    if (!found) return 0;
    hit->ax = ax; hit->ay = ay; hit->bx = bx; hit->by = by; hit->s = best; hit->d = d; hit->sgn = sgn;
    return 1;
}

void mesh_antialias_fwd(const real *color, const real *rast, const real *pos, const int32_t *tri, int B, int V, int T, int H, int W, int C, real *out) {
    size_t P = (size_t)H * W;
    memcpy(out, color, sizeof(real) * (size_t)B * P * C);
    int ne; edge_t *edges = build_edges(tri, T, &ne);
    for (int b = 0; b < B; b++)
        for (int py = 0; py < H; py++)
            for (int px = 0; px < W; px++)
                for (int d = 0; d < 2; d++) {
                    aa_hit_t h;
                    if (!aa_analyze(pos + (size_t)b * V * 4, tri, edges, ne, rast, H, W, b, px, py, d, &h)) continue;
                    real alpha = h.s - (real)0.5;
                    const real *ca = color + ((size_t)b * P + (size_t)h.ay * W + h.ax) * C, *cb = color + ((size_t)b * P + (size_t)h.by * W + h.bx) * C;
                    if (alpha > 0) { real *o = out + ((size_t)b * P + (size_t)h.by * W + h.bx) * C; for (int c = 0; c < C; c++) o[c] += alpha * (ca[c] - cb[c]); }
                    else           { real *o = out + ((size_t)b * P + (size_t)h.ay * W + h.ax) * C; for (int c = 0; c < C; c++) o[c] += -alpha * (cb[c] - ca[c]); }
                }
    free(edges);
}
/* dcolor [B,H,W,C] and dpos [B,V,4]: dcolor is written in full; dpos must be zero-initialised */
void mesh_antialias_bwd(const real *color, const real *rast, const real *pos, const int32_t *tri, const real *dy, int B, int V, int T, int H, int W, int C,
                        real *dcolor, real *dpos) {
    size_t P = (size_t)H * W;
    memcpy(dcolor, dy, sizeof(real) * (size_t)B * P * C);
    int ne; edge_t *edges = build_edges(tri, T, &ne);
    for (int b = 0; b < B; b++)
        for (int py = 0; py < H; py++)
            for (int px = 0; px < W; px++)
                for (int d = 0; d < 2; d++) {
                    aa_hit_t h;
                    const real *pb_ = pos + (size_t)b * V * 4;
                    if (!aa_analyze(pb_, tri, edges, ne, rast, H, W, b, px, py, d, &h)) continue;
                    real alpha = h.s - (real)0.5;
                    size_t ia = ((size_t)b * P + (size_t)h.ay * W + h.ax) * C, ib = ((size_t)b * P + (size_t)h.by * W + h.bx) * C;
                    /* out[dst] += alpha (cA - cB) with dst = B if alpha > 0 else A  (same expression in both cases) */
                    size_t idst = alpha > 0 ? ib : ia;
                    real dalpha = 0;
                    for (int c = 0; c < C; c++) {
                        real g = dy[idst + c];
                        dcolor[ia + c] += alpha * g; dcolor[ib + c] -= alpha * g;
                        dalpha += g * (color[ia + c] - color[ib + c]);
                    }
                    /* alpha = s - 0.5;  s = sgn (xc - cx)/h (d=0) with xc = xa + te (xb - xa), te = da/(da - db), da = ya - cy, db = yb - cy */
                    const real *pa = pb_ + 4 * (size_t)h.va, *pbv = pb_ + 4 * (size_t)h.vb;
                    real xa = pa[0] / pa[3], ya = pa[1] / pa[3], xb = pbv[0] / pbv[3], yb = pbv[1] / pbv[3];
                    real cx = ((real)h.ax + (real)0.5) * ((real)2 / W) - (real)1, cy = ((real)h.ay + (real)0.5) * ((real)2 / H) - (real)1;
                    real hh = d == 0 ? (real)2 / W : (real)2 / H;
                    real gs = dalpha * h.sgn / hh;   /* dL/d(crossing coordinate) */
                    real gxa, gya, gxb, gyb;
                    if (d == 0) {
                        real da = ya - cy, db = yb - cy, den = da - db, te = da / den;
                        /* xc = xa + te (xb - xa) */
                        real gte = gs * (xb - xa);
                        gxa = gs * ((real)1 - te); gxb = gs * te;
                        /* te = da/(da-db): dte/dda = -db/den^2 ; dte/ddb = da/den^2 */
                        gya = gte * (-db / (den * den)); gyb = gte * (da / (den * den));
                    } else {
                        real da = xa - cx, db = xb - cx, den = da - db, te = da / den;
                        real gte = gs * (yb - ya);
                        gya = gs * ((real)1 - te); gyb = gs * te;
                        gxa = gte * (-db / (den * den)); gxb = gte * (da / (den * den));
                    }
                    /* n = p.xy / p.w */
                    real *dA = dpos + ((size_t)b * V + h.va) * 4, *dB = dpos + ((size_t)b * V + h.vb) * 4;
                    dA[0] += gxa / pa[3]; dA[1] += gya / pa[3]; dA[3] += -(gxa * xa + gya * ya) / pa[3];
                    dB[0] += gxb / pbv[3]; dB[1] += gyb / pbv[3]; dB[3] += -(gxb * xb + gyb * yb) / pbv[3];
                }
    free(edges);
}
