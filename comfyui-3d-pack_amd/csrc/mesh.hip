// mesh.hip -- differentiable triangle-mesh ops for gfx950: rasterize, interpolate, texture, antialias (fwd + bwd).
// C-ABI: include/c3d_mesh.h.  Boundary implemented: the `nvdiffrast.torch` calls of
// /root/reference/MVs_Algorithms/DiffRastMesh/diff_mesh_renderer.py:97-138.
//
// Rasterizer design (not the dependency's bin/coarse/fine software pipeline): the reference's meshes are dense
// (config 5: 5e5 triangles on 1024^2, a few pixels per triangle), so the fast path is one lane per triangle that
// walks its own small bounding box and resolves visibility with ONE 64-bit atomicMin per covered pixel on a packed
// (ordered z/w bits << 32 | triangle id) word -- nearest depth, ties to the lower id, order independent, deterministic.
// Triangles with a large bounding box are queued and rasterized by a whole workgroup each.  A resolve pass turns the
// winning id into perspective-correct barycentrics and their screen-space derivatives.
// Coverage is evaluated exactly: vertices snapped to 1/16 pixel, 64-bit integer edge functions, top-left style tie rule.
#include "../../include/c3d_mesh.h"
#include "c3d_common.h"

#define MESH_BIG_BBOX 256          // bounding boxes with more pixels than this go to the workgroup-per-triangle path
#define MESH_EMPTY_KEY 0xFFFFFFFFFFFFFFFFull

struct Frag { float b0, b1, zw, dudx, dudy, dvdx, dvdy; };

__device__ __forceinline__ Frag mesh_shade(const float4 p0, const float4 p1, const float4 p2, float fx, float fy, float xs, float ys) {
    const float p0x = p0.x - fx * p0.w, p0y = p0.y - fy * p0.w;
    const float p1x = p1.x - fx * p1.w, p1y = p1.y - fy * p1.w;
    const float p2x = p2.x - fx * p2.w, p2y = p2.y - fy * p2.w;
    const float a0 = p1x * p2y - p1y * p2x, a1 = p2x * p0y - p2y * p0x, a2 = p0x * p1y - p0y * p1x;
    const float iw = 1.f / (a0 + a1 + a2);
    Frag f;
    f.b0 = a0 * iw; f.b1 = a1 * iw;
    const float z = p0.z * a0 + p1.z * a1 + p2.z * a2, w = p0.w * a0 + p1.w * a1 + p2.w * a2;
    f.zw = z / w;
    const float dfxdx = xs * iw, dfydy = ys * iw;
    const float da0dx = p2.y * p1.w - p1.y * p2.w, da0dy = p1.x * p2.w - p2.x * p1.w;
    const float da1dx = p0.y * p2.w - p2.y * p0.w, da1dy = p2.x * p0.w - p0.x * p2.w;
    const float da2dx = p1.y * p0.w - p0.y * p1.w, da2dy = p0.x * p1.w - p1.x * p0.w;
    const float datdx = da0dx + da1dx + da2dx, datdy = da0dy + da1dy + da2dy;
    f.dudx = dfxdx * (f.b0 * datdx - da0dx); f.dudy = dfydy * (f.b0 * datdy - da0dy);
    f.dvdx = dfxdx * (f.b1 * datdx - da1dx); f.dvdy = dfydy * (f.b1 * datdy - da1dy);
    return f;
}
__device__ __forceinline__ uint32_t ordered_bits(float f) {   // monotone float -> uint; -0 and +0 coincide
    uint32_t u = __float_as_uint(f);
    if (u == 0x80000000u) u = 0u;
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ bool edge_owns_tie(long long dx, long long dy) { return dy > 0 || (dy == 0 && dx < 0); }

struct TriSetup { long long X[3], Y[3], s; int px0, px1, py0, py1; bool ok, clip; };

// A vertex at or behind the camera plane (w <= 0): the dependency clips such a triangle against the view volume; restated per pixel, the part in
// front of the near plane z = -w survives.  Pixel bounding box of the polygon clipped against z + w >= 0 (false: nothing survives).
__device__ __forceinline__ bool mesh_clip_bbox(const float4 p0, const float4 p1, const float4 p2, int W, int H, int& px0, int& px1, int& py0, int& py1) {
    const float4 pp[3] = {p0, p1, p2};
    float bx0 = 0.f, bx1 = 0.f, by0 = 0.f, by1 = 0.f;
    bool any = false, full = false;
#pragma unroll
    for (int k = 0; k < 3; k++) {
        const float4 a = pp[k], c = pp[(k + 1) % 3];
        const float da = a.z + a.w, dc = c.z + c.w;
        float4 cand[2];
        int nc = 0;
        if (da >= 0.f) cand[nc++] = a;
        if ((da >= 0.f) != (dc >= 0.f)) { const float tt = da / (da - dc); cand[nc++] = make_float4(a.x + tt * (c.x - a.x), a.y + tt * (c.y - a.y), a.z + tt * (c.z - a.z), a.w + tt * (c.w - a.w)); }
        for (int q = 0; q < nc; q++) {
            if (!(cand[q].w > 1e-12f)) { full = true; continue; }
            const float nx = cand[q].x / cand[q].w, ny = cand[q].y / cand[q].w;
            if (!any) { bx0 = bx1 = nx; by0 = by1 = ny; any = true; }
            else { bx0 = fminf(bx0, nx); bx1 = fmaxf(bx1, nx); by0 = fminf(by0, ny); by1 = fmaxf(by1, ny); }
        }
    }
    if (!any && !full) return false;
    px0 = 0; py0 = 0; px1 = W - 1; py1 = H - 1;
    if (!full) {
        const float fx0 = (bx0 + 1.f) * 0.5f * W - 1.5f, fx1 = (bx1 + 1.f) * 0.5f * W + 0.5f, fy0 = (by0 + 1.f) * 0.5f * H - 1.5f, fy1 = (by1 + 1.f) * 0.5f * H + 0.5f;
        if (fx0 > 0.f) px0 = fx0 > (float)W ? W : (int)fx0;
        if (fy0 > 0.f) py0 = fy0 > (float)H ? H : (int)fy0;
        if (fx1 < (float)(W - 1)) px1 = fx1 < -1.f ? -1 : (int)fx1;
        if (fy1 < (float)(H - 1)) py1 = fy1 < -1.f ? -1 : (int)fy1;
    }
    return px0 <= px1 && py0 <= py1;
}
// coverage of a clipped triangle: homogeneous barycentrics all >= 0 (inclusive) and interpolated w > 0
__device__ __forceinline__ bool mesh_covers_clip(const float4 p0, const float4 p1, const float4 p2, float fx, float fy) {
    const float p0x = p0.x - fx * p0.w, p0y = p0.y - fy * p0.w, p1x = p1.x - fx * p1.w, p1y = p1.y - fy * p1.w, p2x = p2.x - fx * p2.w, p2y = p2.y - fy * p2.w;
    const float a0 = p1x * p2y - p1y * p2x, a1 = p2x * p0y - p2y * p0x, a2 = p0x * p1y - p0y * p1x, sum = a0 + a1 + a2;
    if (sum == 0.f) return false;
    const float b0 = a0 / sum, b1 = a1 / sum, b2 = a2 / sum;
    if (!(b0 >= 0.f && b1 >= 0.f && b2 >= 0.f)) return false;
    return b0 * p0.w + b1 * p1.w + b2 * p2.w > 0.f;
}

__device__ __forceinline__ TriSetup mesh_setup(const float4 p0, const float4 p1, const float4 p2, int W, int H) {
    TriSetup t;
    t.ok = false; t.clip = false;
    if (!(p0.w > 0.f && p1.w > 0.f && p2.w > 0.f)) {
        t.s = 1;
        t.ok = t.clip = mesh_clip_bbox(p0, p1, p2, W, H, t.px0, t.px1, t.py0, t.py1);
        return t;
    }
    const float4 pp[3] = {p0, p1, p2};
#pragma unroll
    for (int k = 0; k < 3; k++) {
        const float nx = pp[k].x / pp[k].w, ny = pp[k].y / pp[k].w;
        if (!(fabsf(nx) < 1e4f && fabsf(ny) < 1e4f)) return t;
        t.X[k] = (long long)__float2int_rn(nx * (float)(W * 8)) + (long long)W * 8;
        t.Y[k] = (long long)__float2int_rn(ny * (float)(H * 8)) + (long long)H * 8;
    }
    const long long area = (t.X[1] - t.X[0]) * (t.Y[2] - t.Y[0]) - (t.Y[1] - t.Y[0]) * (t.X[2] - t.X[0]);
    if (area == 0) return t;
    t.s = area > 0 ? 1 : -1;
    const long long xmin = min(t.X[0], min(t.X[1], t.X[2])), xmax = max(t.X[0], max(t.X[1], t.X[2]));
    const long long ymin = min(t.Y[0], min(t.Y[1], t.Y[2])), ymax = max(t.Y[0], max(t.Y[1], t.Y[2]));
    long long px0 = (xmin - 8 + 15) >> 4, px1 = (xmax - 8) >> 4, py0 = (ymin - 8 + 15) >> 4, py1 = (ymax - 8) >> 4;
    t.px0 = (int)max(px0, 0ll); t.py0 = (int)max(py0, 0ll);
    t.px1 = (int)min(px1, (long long)W - 1); t.py1 = (int)min(py1, (long long)H - 1);
    t.ok = t.px0 <= t.px1 && t.py0 <= t.py1;
    return t;
}
__device__ __forceinline__ bool mesh_covers(const TriSetup& t, int px, int py) {
    const long long cx = 16ll * px + 8, cy = 16ll * py + 8;
#pragma unroll
    for (int k = 0; k < 3; k++) {
        const int a = k, c = (k + 1) % 3;
        const long long dx = (t.X[c] - t.X[a]) * t.s, dy = (t.Y[c] - t.Y[a]) * t.s;
        const long long e = dx * (cy - t.Y[a]) - dy * (cx - t.X[a]);
        if (e < 0 || (e == 0 && !edge_owns_tie(dx, dy))) return false;
    }
    return true;
}
__device__ __forceinline__ void mesh_plot(const float4 p0, const float4 p1, const float4 p2, int px, int py, int W, int H, float xs, float ys,
                                          uint32_t t, unsigned long long* zbuf, const unsigned long long* peel) {
    const Frag f = mesh_shade(p0, p1, p2, xs * ((float)px + 0.5f) - 1.f, ys * ((float)py + 0.5f) - 1.f, xs, ys);
    if (!(f.zw >= -1.f && f.zw <= 1.f)) return;
    const uint32_t zb = ordered_bits(f.zw);
    if (peel) {   // depth peeling: only what lies strictly behind the previous layer's surface, and only where that layer had one
        const unsigned long long pk = peel[(size_t)py * W + px];
        if (pk == MESH_EMPTY_KEY || zb <= (uint32_t)(pk >> 32)) return;
    }
    // Measured in round 4 and dropped (mesh bench, 8 views per launch, same box): a plain "is the stored key already nearer?" load in front of the atomic --
    // rasterize 0.298 -> 0.347 ms; an XCD-aware order (each XCD a contiguous eighth of a view's triangles / pixel tiles, so that the lines the atomics hit stay
    // in one L2) -- rasterize 0.298 -> 0.324, texture backward 0.247 -> 0.262: the compact parts of a view are unequal work (background bands, back faces).
    atomicMin(&zbuf[(size_t)py * W + px], ((unsigned long long)zb << 32) | t);
}

__global__ void __launch_bounds__(256) k_ras_tri(const float4* __restrict__ pos, const int3* __restrict__ tri, int B, int V, int T, int H, int W,
                                                  unsigned long long* __restrict__ zbuf, uint32_t* __restrict__ big_queue, uint32_t* __restrict__ big_count,
                                                  const unsigned long long* __restrict__ peel) {
    const long long gid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= (long long)B * T) return;
    const int b = (int)(gid / T), t = (int)(gid % T);
    const int3 vi = tri[t];
    if ((unsigned)vi.x >= (unsigned)V || (unsigned)vi.y >= (unsigned)V || (unsigned)vi.z >= (unsigned)V) return;
    const float4* pb = pos + (size_t)b * V;
    const float4 p0 = pb[vi.x], p1 = pb[vi.y], p2 = pb[vi.z];
    const TriSetup ts = mesh_setup(p0, p1, p2, W, H);
    if (!ts.ok) return;
    const long long area = (long long)(ts.px1 - ts.px0 + 1) * (ts.py1 - ts.py0 + 1);
    if (area > MESH_BIG_BBOX || ts.clip) { big_queue[atomicAdd(big_count, 1u)] = (uint32_t)gid; return; }   // near-plane clipped triangles: rare, a workgroup each
    const float xs = 2.f / W, ys = 2.f / H;
    unsigned long long* zb = zbuf + (size_t)b * H * W;
    const unsigned long long* pl = peel ? peel + (size_t)b * H * W : nullptr;
    for (int py = ts.py0; py <= ts.py1; py++)
        for (int px = ts.px0; px <= ts.px1; px++)
            if (mesh_covers(ts, px, py)) mesh_plot(p0, p1, p2, px, py, W, H, xs, ys, (uint32_t)t, zb, pl);
}
// one workgroup per queued large triangle (grid-stride over the queue; the queue length is read on the device)
__global__ void __launch_bounds__(256) k_ras_big(const float4* __restrict__ pos, const int3* __restrict__ tri, int V, int T, int H, int W,
                                                  unsigned long long* __restrict__ zbuf, const uint32_t* __restrict__ big_queue,
                                                  const uint32_t* __restrict__ big_count, const unsigned long long* __restrict__ peel) {
    const uint32_t n = *big_count;
    const float xs = 2.f / W, ys = 2.f / H;
    for (uint32_t q = blockIdx.x; q < n; q += gridDim.x) {
        const uint32_t gid = big_queue[q];
        const int b = (int)(gid / (uint32_t)T), t = (int)(gid % (uint32_t)T);
        const int3 vi = tri[t];
        const float4* pb = pos + (size_t)b * V;
        const float4 p0 = pb[vi.x], p1 = pb[vi.y], p2 = pb[vi.z];
        const TriSetup ts = mesh_setup(p0, p1, p2, W, H);
        const int bw = ts.px1 - ts.px0 + 1;
        const long long area = (long long)bw * (ts.py1 - ts.py0 + 1);
        unsigned long long* zb = zbuf + (size_t)b * H * W;
        const unsigned long long* pl = peel ? peel + (size_t)b * H * W : nullptr;
        for (long long i = threadIdx.x; i < area; i += blockDim.x) {
            const int px = ts.px0 + (int)(i % bw), py = ts.py0 + (int)(i / bw);
            const bool in = ts.clip ? mesh_covers_clip(p0, p1, p2, xs * ((float)px + 0.5f) - 1.f, ys * ((float)py + 0.5f) - 1.f) : mesh_covers(ts, px, py);
            if (in) mesh_plot(p0, p1, p2, px, py, W, H, xs, ys, (uint32_t)t, zb, pl);
        }
    }
}
__global__ void __launch_bounds__(256) k_ras_resolve(const float4* __restrict__ pos, const int3* __restrict__ tri, int B, int V, int H, int W,
                                                      const unsigned long long* __restrict__ zbuf, float4* __restrict__ rast, float4* __restrict__ rast_db) {
    const long long gid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long P = (long long)H * W;
    if (gid >= (long long)B * P) return;
    const unsigned long long key = zbuf[gid];
    if (key == MESH_EMPTY_KEY) {
        rast[gid] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (rast_db) rast_db[gid] = make_float4(0.f, 0.f, 0.f, 0.f);
        return;
    }
    const int b = (int)(gid / P), pid = (int)(gid % P), px = pid % W, py = pid / W;
    const uint32_t t = (uint32_t)(key & 0xFFFFFFFFull);
    const int3 vi = tri[t];
    const float4* pb = pos + (size_t)b * V;
    const float xs = 2.f / W, ys = 2.f / H;
    const Frag f = mesh_shade(pb[vi.x], pb[vi.y], pb[vi.z], xs * ((float)px + 0.5f) - 1.f, ys * ((float)py + 0.5f) - 1.f, xs, ys);
    rast[gid] = make_float4(fminf(fmaxf(f.b0, 0.f), 1.f), fminf(fmaxf(f.b1, 0.f), 1.f), fminf(fmaxf(f.zw, -1.f), 1.f), (float)(t + 1));
    if (rast_db) rast_db[gid] = make_float4(f.dudx, f.dudy, f.dvdx, f.dvdy);
}

// Gradient of one pixel w.r.t. the three clip-space corners: g0, g1 = d/d(u, v) (not through a clamped value), G = d/d rast_db = d/d(du/dX, du/dY, dv/dX, dv/dY)
// (round 3: the dependency's grad_db path -- what texture(uv_da) -> interpolate(rast_db) hands back; zero = absent).  With A_kx = d a_k / d fx, A_ky = d a_k / d fy
// (functions of the un-shifted x, y, w) and T = their sums: du/dX = xs iw (b0 Tx - A0x), ...; the chain rule runs through b0, b1, iw and the six A terms.
// Accumulates into ax / ay / aw (x, y, w components of the corners' gradients).
template <bool DB>
__device__ __forceinline__ void ras_bwd_corner_grads(const float4 p0, const float4 p1, const float4 p2, float fx, float fy, float xs, float ys, float g0, float g1,
                                                     const float4 G, float ax[3], float ay[3], float aw[3]) {
    const float p0x = p0.x - fx * p0.w, p0y = p0.y - fy * p0.w;
    const float p1x = p1.x - fx * p1.w, p1y = p1.y - fy * p1.w;
    const float p2x = p2.x - fx * p2.w, p2y = p2.y - fy * p2.w;
    const float a0 = p1x * p2y - p1y * p2x, a1 = p2x * p0y - p2y * p0x, a2 = p0x * p1y - p0y * p1x;
    const float iw = 1.f / (a0 + a1 + a2), b0 = a0 * iw, b1 = a1 * iw;
    if (b0 < 0.f || b0 > 1.f) g0 = 0.f;   // forward clamps u,v: no gradient through a clamped value
    if (b1 < 0.f || b1 > 1.f) g1 = 0.f;
    float da0 = (g0 * (1.f - b0) - g1 * b1) * iw, da1 = (-g0 * b0 + g1 * (1.f - b1)) * iw, da2 = (-g0 * b0 - g1 * b1) * iw;
    if (DB) {
        const float A0x = p1.w * p2.y - p1.y * p2.w, A0y = p1.x * p2.w - p2.x * p1.w;
        const float A1x = p0.y * p2.w - p2.y * p0.w, A1y = p2.x * p0.w - p0.x * p2.w;
        const float A2x = p1.y * p0.w - p0.y * p1.w, A2y = p0.x * p1.w - p1.x * p0.w;
        const float Tx = A0x + A1x + A2x, Ty = A0y + A1y + A2y, al = xs * iw, be = ys * iw;
        const float dudx = al * (b0 * Tx - A0x), dudy = be * (b0 * Ty - A0y), dvdx = al * (b1 * Tx - A1x), dvdy = be * (b1 * Ty - A1y);
        const float gb0 = G.x * al * Tx + G.y * be * Ty, gb1 = G.z * al * Tx + G.w * be * Ty;
        const float gTx = al * (G.x * b0 + G.z * b1), gTy = be * (G.y * b0 + G.w * b1);
        const float Lv = G.x * dudx + G.y * dudy + G.z * dvdx + G.w * dvdy;
        const float gi = Lv / iw + gb0 * a0 + gb1 * a1, gS = -iw * iw * gi;
        da0 += gb0 * iw + gS; da1 += gb1 * iw + gS; da2 += gS;
        const float gx[3] = {gTx - al * G.x, gTx - al * G.z, gTx}, gy[3] = {gTy - be * G.y, gTy - be * G.w, gTy};
        const float4 pp[3] = {p0, p1, p2};
#pragma unroll
        for (int k = 0; k < 3; k++) {      // A_kx = y_m w_l - y_l w_m, A_ky = x_l w_m - x_m w_l with (l, m) = (k + 1, k + 2) mod 3
            const int l = (k + 1) % 3, m = (k + 2) % 3;
            ay[m] += gx[k] * pp[l].w; aw[l] += gx[k] * pp[m].y; ay[l] -= gx[k] * pp[m].w; aw[m] -= gx[k] * pp[l].y;
            ax[l] += gy[k] * pp[m].w; aw[m] += gy[k] * pp[l].x; ax[m] -= gy[k] * pp[l].w; aw[l] -= gy[k] * pp[m].x;
        }
    }
    const float dx[3] = {da1 * (-p2y) + da2 * p1y, da0 * p2y + da2 * (-p0y), da0 * (-p1y) + da1 * p0y};
    const float dyv[3] = {da1 * p2x + da2 * (-p1x), da0 * (-p2x) + da2 * p0x, da0 * p1x + da1 * (-p0x)};
#pragma unroll
    for (int k = 0; k < 3; k++) { ax[k] += dx[k]; ay[k] += dyv[k]; aw[k] += -fx * dx[k] - fy * dyv[k]; }
}
__global__ void __launch_bounds__(256) k_ras_bwd(const float4* __restrict__ pos, const int3* __restrict__ tri, const float4* __restrict__ rast,
                                                  const float4* __restrict__ dy, const float4* __restrict__ ddb, int B, int V, int H, int W, float* __restrict__ dpos) {
    const long long gid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long P = (long long)H * W;
    if (gid >= (long long)B * P) return;
    const float4 r = rast[gid];
    const int t = (int)r.w - 1;
    if (t < 0) return;
    const float4 g = dy ? dy[gid] : make_float4(0.f, 0.f, 0.f, 0.f);
    const float4 G = ddb ? ddb[gid] : make_float4(0.f, 0.f, 0.f, 0.f);
    const bool with_db = G.x != 0.f || G.y != 0.f || G.z != 0.f || G.w != 0.f;
    if (g.x == 0.f && g.y == 0.f && !with_db) return;
    const int b = (int)(gid / P), pid = (int)(gid % P), px = pid % W, py = pid / W;
    const int3 vi = tri[t];
    const float4* pb = pos + (size_t)b * V;
    float ax[3] = {0.f, 0.f, 0.f}, ay[3] = {0.f, 0.f, 0.f}, aw[3] = {0.f, 0.f, 0.f};
    const float fx = (2.f / W) * ((float)px + 0.5f) - 1.f, fy = (2.f / H) * ((float)py + 0.5f) - 1.f;
    if (with_db) ras_bwd_corner_grads<true>(pb[vi.x], pb[vi.y], pb[vi.z], fx, fy, 2.f / W, 2.f / H, g.x, g.y, G, ax, ay, aw);
    else ras_bwd_corner_grads<false>(pb[vi.x], pb[vi.y], pb[vi.z], fx, fy, 2.f / W, 2.f / H, g.x, g.y, G, ax, ay, aw);
    const int vv[3] = {vi.x, vi.y, vi.z};
#pragma unroll
    for (int k = 0; k < 3; k++) {
        float* d = dpos + ((size_t)b * V + vv[k]) * 4;
        atomicAdd(d + 0, ax[k]); atomicAdd(d + 1, ay[k]); atomicAdd(d + 3, aw[k]);
    }
}

// ---- atomic-free, bit-reproducible variant --------------------------------------------------------------------------------------------
// Scatter-adds from pixels to vertices are the expensive part of every mesh backward op on this chip (device-scope float atomics execute
// beyond the XCD-private L2, and neighbouring pixels hit the same vertices).  The gather formulation needs none: (1) triangle-parallel, as in
// the forward pass: a lane re-walks its triangle's bounding box, keeps the pixels it OWNS (rast id == its id), and accumulates the gradient
// of its three corners in registers -> one 16-byte record per (triangle, corner); (2) vertex-parallel: a lane sums the records of its
// vertex through the vertex -> corner adjacency (CSR, built once per topology with the radix sort of the binning stage).  Fixed summation
// order on both levels: the result is deterministic.
struct VertexTopo { uint32_t* start; uint32_t* key[2]; uint32_t* val[2]; int* meta; void* tmp; size_t bytes; };
static void carve_vertex_topo(char* base, int V, int T, VertexTopo& t) {
    size_t off = 0;
    auto take = [&](size_t b) { char* p = base ? base + off : nullptr; off += c3d_align(b); return p; };
    const size_t n = 3 * (size_t)(T > 0 ? T : 1);
    t.start = (uint32_t*)take(4 * ((size_t)(V > 0 ? V : 0) + 2));
    t.key[0] = (uint32_t*)take(4 * n); t.key[1] = (uint32_t*)take(4 * n);
    t.val[0] = (uint32_t*)take(4 * n); t.val[1] = (uint32_t*)take(4 * n);
    t.meta = (int*)take(64);
    t.tmp = take(c3d_sort_tmp_bytes(n));
    t.bytes = off;
}
__global__ void __launch_bounds__(256) k_topo_keys(const int* __restrict__ tri, int n3, int V, uint32_t* __restrict__ key) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n3) { const int v = tri[i]; key[i] = (unsigned)v < (unsigned)V ? (uint32_t)v : (uint32_t)V; }   // out-of-range ids park behind the last vertex
}
__global__ void __launch_bounds__(256) k_topo_starts(const uint32_t* __restrict__ skey, int n3, int V, uint32_t* __restrict__ start, int* __restrict__ meta, int res) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i == 0) meta[0] = res;
    if (i > n3) return;
    const long long prev = i ? (long long)skey[i - 1] : -1ll, cur = i < n3 ? (long long)skey[i] : (long long)V + 1;
    for (long long v = prev + 1; v <= cur && v <= (long long)V + 1; v++) start[v] = (uint32_t)i;   // start[v] = first sorted slot with key >= v
}

// ---- the antialias pass of the fused view as gathers (round 3) ------------------------------------------------------------------------------
// The silhouette analysis (k_aa2_pairs) leaves per pixel pair (first pixel p, direction d: second = p + 1 | p + W) a flag byte -- bit 0 hit, bit 1 "the first pixel
// is pixel a" (the one that owns the nearer triangle), bits 2-3 the silhouette edge's first corner in a's triangle -- and the blend weight alpha = s - 0.5.
// A pixel belongs to four pairs: (p, 0), (p, 1), (p - 1, 0), (p - W, 1).  Forward blend, colour gradient and position gradient each GATHER over those four in that
// order instead of scattering with float atomics: no atomics, one summation order, the same bits every run.
struct AaPair { bool hit; int k; float alpha, sgn; size_t ia, ib, idst; };
__device__ __forceinline__ AaPair aa_pair_load(const uint8_t* __restrict__ hit, const float* __restrict__ pair_alpha, size_t first, int d, int W) {
    AaPair r;
    const size_t gid = 2 * first + (size_t)d;
    const uint32_t f = hit[gid];
    r.hit = (f & 1u) != 0;
    if (!r.hit) return r;
    const size_t second = first + (d == 0 ? (size_t)1 : (size_t)W);
    const bool a_first = (f & 2u) != 0;
    r.k = (int)((f >> 2) & 3u);
    r.sgn = a_first ? 1.f : -1.f;
    r.ia = a_first ? first : second; r.ib = a_first ? second : first;
    r.alpha = pair_alpha[gid];
    r.idst = r.alpha > 0.f ? r.ib : r.ia;
    return r;
}
// pair j = 0..3 of pixel (px, py): its first pixel, its direction, and whether it exists
__device__ __forceinline__ bool aa_pair_of(int j, int px, int py, int W, size_t pid, size_t& first, int& d) {
    d = j & 1;
    if (j < 2) { first = pid; return true; }
    if (j == 2) { first = pid - 1; return px > 0; }
    first = pid - (size_t)W; return py > 0;
}
// Bit t of view b's row: some pixel of the view carries triangle t's id (set by the fused view's pixel pass).  With ~2-pixel triangles more than half of a closed
// mesh's triangles own no pixel (back faces, hidden, off screen): their gradient record is zero, and the backward pass neither computes nor stores nor gathers it.
// bits == NULL (the stand-alone op, which has no forward state): every triangle counts as an owner.
struct TriOwned {
    const uint32_t* bits; int words;      // words per view
    __device__ __forceinline__ bool has(int b, int t) const { return !bits || ((bits[(size_t)b * words + (t >> 5)] >> (t & 31)) & 1u); }
};
// what the position gradient of the antialias needs besides the pair records: the colours that were blended and the gradients of the two antialias outputs
struct AaBwdIn { const uint8_t* hit; const uint8_t* pflag; const float* pair_alpha; const float* albedo0; const float* dy3; const float* dy1; };      // pflag: k_view_shade_fwd_g
// position gradient of the pairs whose pixel a is `pid` (owned by the triangle with corners p0, p1, p2) -- the algebra of k_aa_bwd (the standalone antialias backward), accumulated per corner
__device__ __forceinline__ void aa_bwd_pixel(const AaBwdIn& aa, const float4* __restrict__ rast, const float4 pc[3], int px, int py, size_t pid, int H, int W,
                                             float ax[3], float ay[3], float aw[3]) {
    const uint32_t pf = aa.pflag[pid] & 15u;      // bit j: pair j is a hit and this pixel is its pixel a (one byte per pixel; nearly always zero)
    if (!pf) return;
#pragma unroll
    for (int j = 0; j < 4; j++) {
        if (!(pf >> j & 1u)) continue;
        size_t first; int d;
        aa_pair_of(j, px, py, W, pid, first, d);
        const AaPair pr = aa_pair_load(aa.hit, aa.pair_alpha, first, d, W);
        float dalpha = 0.f;
#pragma unroll
        for (int c = 0; c < 3; c++) dalpha += aa.dy3[3 * pr.idst + c] * (aa.albedo0[3 * pr.ia + c] - aa.albedo0[3 * pr.ib + c]);
        const float cb = rast[pr.ib].w > 0.f ? 1.f : 0.f;
        dalpha += aa.dy1[pr.idst] * (1.f - cb);
        const int k0 = pr.k, k1 = pr.k == 2 ? 0 : pr.k + 1;
        const float4 pa = k0 == 0 ? pc[0] : (k0 == 1 ? pc[1] : pc[2]), pv = k1 == 0 ? pc[0] : (k1 == 1 ? pc[1] : pc[2]);
        const float xa = pa.x / pa.w, ya = pa.y / pa.w, xb = pv.x / pv.w, yb = pv.y / pv.w;
        const float cx = ((float)px + 0.5f) * (2.f / W) - 1.f, cy = ((float)py + 0.5f) * (2.f / H) - 1.f;
        const float hh = d == 0 ? 2.f / W : 2.f / H;
        const float gs = dalpha * pr.sgn / hh;
        float gxa, gya, gxb, gyb;
        if (d == 0) {
            const float da = ya - cy, db = yb - cy, den = da - db, te = da / den, gte = gs * (xb - xa);
            gxa = gs * (1.f - te); gxb = gs * te;
            gya = gte * (-db / (den * den)); gyb = gte * (da / (den * den));
        } else {
            const float da = xa - cx, db = xb - cx, den = da - db, te = da / den, gte = gs * (yb - ya);
            gya = gs * (1.f - te); gyb = gs * te;
            gxa = gte * (-db / (den * den)); gxb = gte * (da / (den * den));
        }
        const float a3[3] = {gxa / pa.w, gya / pa.w, -(gxa * xa + gya * ya) / pa.w}, b3[3] = {gxb / pv.w, gyb / pv.w, -(gxb * xb + gyb * yb) / pv.w};
#pragma unroll
        for (int k = 0; k < 3; k++) {
            if (k == k0) { ax[k] += a3[0]; ay[k] += a3[1]; aw[k] += a3[2]; }
            if (k == k1) { ax[k] += b3[0]; ay[k] += b3[1]; aw[k] += b3[2]; }
        }
    }
}
template <bool DB>
__global__ void __launch_bounds__(256) k_ras_bwd_tri(const float4* __restrict__ pos, const int3* __restrict__ tri, const float4* __restrict__ rast,
                                                      const float4* __restrict__ dy, const float4* __restrict__ ddb, int B, int V, int T, int H, int W,
                                                      float4* __restrict__ rec, uint32_t* __restrict__ big_queue, uint32_t* __restrict__ big_count, AaBwdIn aa, TriOwned own) {
    const long long gid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= (long long)B * T) return;
    const int b = (int)(gid / T), t = (int)(gid % T);
    if (!own.has(b, t)) return;      // no pixel carries this triangle's id: its record is all zero, it is neither written here nor read by the vertex gather
    float4* out = rec + (size_t)gid * 3;
    const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
    const int3 vi = tri[t];
    if ((unsigned)vi.x >= (unsigned)V || (unsigned)vi.y >= (unsigned)V || (unsigned)vi.z >= (unsigned)V) { out[0] = z4; out[1] = z4; out[2] = z4; return; }
    const float4* pb = pos + (size_t)b * V;
    const float4 p0 = pb[vi.x], p1 = pb[vi.y], p2 = pb[vi.z];
    const TriSetup ts = mesh_setup(p0, p1, p2, W, H);
    if (!ts.ok) { out[0] = z4; out[1] = z4; out[2] = z4; return; }
    const long long area = (long long)(ts.px1 - ts.px0 + 1) * (ts.py1 - ts.py0 + 1);
    if (area > MESH_BIG_BBOX || ts.clip) { big_queue[atomicAdd(big_count, 1u)] = (uint32_t)gid; return; }   // records written by k_ras_bwd_big
    const float xs = 2.f / W, ys = 2.f / H, idf = (float)(t + 1);
    const size_t pbase = (size_t)b * H * W;
    float ax[3] = {0.f, 0.f, 0.f}, ay[3] = {0.f, 0.f, 0.f}, aw[3] = {0.f, 0.f, 0.f};
    for (int py = ts.py0; py <= ts.py1; py++)
        for (int px = ts.px0; px <= ts.px1; px++) {
            const size_t pid = pbase + (size_t)py * W + px;
            if (rast[pid].w != idf) continue;
            const float4 g = dy ? dy[pid] : make_float4(0.f, 0.f, 0.f, 0.f);
            const float4 G = DB ? ddb[pid] : make_float4(0.f, 0.f, 0.f, 0.f);
            if (g.x != 0.f || g.y != 0.f || (DB && (G.x != 0.f || G.y != 0.f || G.z != 0.f || G.w != 0.f)))
                ras_bwd_corner_grads<DB>(p0, p1, p2, xs * ((float)px + 0.5f) - 1.f, ys * ((float)py + 0.5f) - 1.f, xs, ys, g.x, g.y, G, ax, ay, aw);
            if (aa.hit) { const float4 pc[3] = {p0, p1, p2}; aa_bwd_pixel(aa, rast, pc, px, py, pid, H, W, ax, ay, aw); }      // fused view only (B = 1)
        }
#pragma unroll
    for (int k = 0; k < 3; k++) out[k] = make_float4(ax[k], ay[k], 0.f, aw[k]);
}
template <bool DB>
__global__ void __launch_bounds__(256) k_ras_bwd_big(const float4* __restrict__ pos, const int3* __restrict__ tri, const float4* __restrict__ rast,
                                                      const float4* __restrict__ dy, const float4* __restrict__ ddb, int V, int T, int H, int W, float4* __restrict__ rec,
                                                      const uint32_t* __restrict__ big_queue, const uint32_t* __restrict__ big_count, AaBwdIn aa) {
    __shared__ float red[4][9];
    const uint32_t n = *big_count;
    const float xs = 2.f / W, ys = 2.f / H;
    for (uint32_t qi = blockIdx.x; qi < n; qi += gridDim.x) {
        const uint32_t gid = big_queue[qi];
        const int b = (int)(gid / (uint32_t)T), t = (int)(gid % (uint32_t)T);
        const int3 vi = tri[t];
        const float4* pb = pos + (size_t)b * V;
        const float4 p0 = pb[vi.x], p1 = pb[vi.y], p2 = pb[vi.z];
        const TriSetup ts = mesh_setup(p0, p1, p2, W, H);
        const int bw = ts.px1 - ts.px0 + 1;
        const long long area = (long long)bw * (ts.py1 - ts.py0 + 1);
        const size_t pbase = (size_t)b * H * W;
        const float idf = (float)(t + 1);
        float ax[3] = {0.f, 0.f, 0.f}, ay[3] = {0.f, 0.f, 0.f}, aw[3] = {0.f, 0.f, 0.f};
        for (long long i = threadIdx.x; i < area; i += blockDim.x) {
            const int px = ts.px0 + (int)(i % bw), py = ts.py0 + (int)(i / bw);
            const size_t pid = pbase + (size_t)py * W + px;
            if (rast[pid].w != idf) continue;
            const float4 g = dy ? dy[pid] : make_float4(0.f, 0.f, 0.f, 0.f);
            const float4 G = DB ? ddb[pid] : make_float4(0.f, 0.f, 0.f, 0.f);
            if (g.x != 0.f || g.y != 0.f || (DB && (G.x != 0.f || G.y != 0.f || G.z != 0.f || G.w != 0.f)))
                ras_bwd_corner_grads<DB>(p0, p1, p2, xs * ((float)px + 0.5f) - 1.f, ys * ((float)py + 0.5f) - 1.f, xs, ys, g.x, g.y, G, ax, ay, aw);
            if (aa.hit) { const float4 pc[3] = {p0, p1, p2}; aa_bwd_pixel(aa, rast, pc, px, py, pid, H, W, ax, ay, aw); }
        }
        float v9[9] = {ax[0], ay[0], aw[0], ax[1], ay[1], aw[1], ax[2], ay[2], aw[2]};
        __syncthreads();
#pragma unroll
        for (int q = 0; q < 9; q++) {
            const float sum = c3d_wave_sum(v9[q]);
            if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6][q] = sum;
        }
        __syncthreads();
        if (threadIdx.x < 3) {
            const int k = threadIdx.x;
            float r3[3];
#pragma unroll
            for (int c = 0; c < 3; c++) r3[c] = red[0][3 * k + c] + red[1][3 * k + c] + red[2][3 * k + c] + red[3][3 * k + c];
            rec[(size_t)gid * 3 + k] = make_float4(r3[0], r3[1], 0.f, r3[2]);
        }
    }
}
// out[b][v] = sum of the corner records of vertex v, in corner-id order (the adjacency is sorted stably).  Vertices with more than
// VG_HEAVY corners (the poles of a lat-long sphere have hundreds) are queued for a wave each: one lane walking such a list serially set the
// kernel's duration (0.14 ms for two pole vertices).
#define VG_HEAVY 32
__global__ void __launch_bounds__(256) k_vertex_gather4(const float4* __restrict__ rec, const uint32_t* __restrict__ start, const uint32_t* __restrict__ corner,
                                                         int B, int V, int T, float4* __restrict__ out, uint32_t* __restrict__ heavy_queue, uint32_t* __restrict__ heavy_count, TriOwned own) {
    const long long gid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= (long long)B * V) return;
    const int b = (int)(gid / V), v = (int)(gid % V);
    const uint32_t i0 = start[v], i1 = start[v + 1];
    if (i1 - i0 > VG_HEAVY) { heavy_queue[atomicAdd(heavy_count, 1u)] = (uint32_t)gid; return; }
    const float4* rb = rec + (size_t)b * T * 3;
    float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
    // eight corners at a time, three rounds of independent loads (corner ids, owner bits, records) instead of a chain of dependent pairs per corner; summed in list order
    for (uint32_t i = i0; i < i1; i += 8) {
        uint32_t c[8]; bool on[8]; float4 r[8];
#pragma unroll
        for (int k = 0; k < 8; k++) c[k] = i + k < i1 ? corner[i + k] : 0xFFFFFFFFu;
#pragma unroll
        for (int k = 0; k < 8; k++) on[k] = c[k] != 0xFFFFFFFFu && own.has(b, (int)(c[k] / 3u));
#pragma unroll
        for (int k = 0; k < 8; k++) r[k] = on[k] ? rb[c[k]] : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int k = 0; k < 8; k++) { a.x += r[k].x; a.y += r[k].y; a.z += r[k].z; a.w += r[k].w; }
    }
    out[gid] = a;
}
// one wave per queued vertex: lane l takes corners l, l + 64, ... (fixed assignment), then a fixed-shape wave reduction -> still deterministic
__global__ void __launch_bounds__(256) k_vertex_gather4_heavy(const float4* __restrict__ rec, const uint32_t* __restrict__ start, const uint32_t* __restrict__ corner,
                                                               int V, int T, float4* __restrict__ out, const uint32_t* __restrict__ heavy_queue,
                                                               const uint32_t* __restrict__ heavy_count, TriOwned own) {
    const uint32_t n = *heavy_count;
    const int lane = threadIdx.x & 63;
    for (uint32_t q = blockIdx.x * 4 + (threadIdx.x >> 6); q < n; q += gridDim.x * 4) {
        const uint32_t gid = heavy_queue[q];
        const int b = (int)(gid / (uint32_t)V), v = (int)(gid % (uint32_t)V);
        const float4* rb = rec + (size_t)b * T * 3;
        float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
        for (uint32_t i = start[v] + lane, e = start[v + 1]; i < e; i += 64) {
            const uint32_t c = corner[i];
            if (!own.has(b, (int)(c / 3u))) continue;
            const float4 r = rb[c];
            a.x += r.x; a.y += r.y; a.z += r.z; a.w += r.w;
        }
        a.x = c3d_wave_sum(a.x); a.y = c3d_wave_sum(a.y); a.z = c3d_wave_sum(a.z); a.w = c3d_wave_sum(a.w);
        if (lane == 0) out[gid] = a;
    }
}

// ------------------------------------------------------------------------------------------ interpolate
__global__ void __launch_bounds__(256) k_interp_fwd(const float* __restrict__ attr, int Ba, const float4* __restrict__ rast, const int3* __restrict__ tri,
                                                     const float4* __restrict__ rast_db, const int* __restrict__ diff, int nd, long long BP, long long P,
                                                     int V, int A, float* __restrict__ out, float* __restrict__ out_da) {
    const long long gid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= BP) return;
    const float4 r = rast[gid];
    const int t = (int)r.w - 1;
    float* po = out + gid * A;
    if (t < 0) {
        for (int a = 0; a < A; a++) po[a] = 0.f;
        for (int a = 0; a < 2 * nd; a++) out_da[gid * 2 * nd + a] = 0.f;
        return;
    }
    const int b = (int)(gid / P);
    const float* ab = attr + (size_t)(Ba > 1 ? b : 0) * V * A;
    const int3 vi = tri[t];
    const float *a0 = ab + (size_t)vi.x * A, *a1 = ab + (size_t)vi.y * A, *a2 = ab + (size_t)vi.z * A;
    const float u = r.x, v = r.y, w2 = 1.f - u - v;
    for (int a = 0; a < A; a++) po[a] = u * a0[a] + v * a1[a] + w2 * a2[a];
    if (nd) {
        const float4 db = rast_db[gid];
        for (int k = 0; k < nd; k++) {
            const int a = diff[k];
            const float dsdu = a0[a] - a2[a], dsdv = a1[a] - a2[a];
            out_da[(gid * nd + k) * 2] = dsdu * db.x + dsdv * db.z;
            out_da[(gid * nd + k) * 2 + 1] = dsdu * db.y + dsdv * db.w;
        }
    }
}
__global__ void __launch_bounds__(256) k_interp_bwd(const float* __restrict__ attr, int Ba, const float4* __restrict__ rast, const int3* __restrict__ tri,
                                                     const float* __restrict__ dy, long long BP, long long P, int V, int A, float* __restrict__ dattr,
                                                     float4* __restrict__ drast) {
    const long long gid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= BP) return;
    const float4 r = rast[gid];
    const int t = (int)r.w - 1;
    if (t < 0) { drast[gid] = make_float4(0.f, 0.f, 0.f, 0.f); return; }
    const int b = (int)(gid / P);
    const size_t ab = (size_t)(Ba > 1 ? b : 0) * V * A;
    const int3 vi = tri[t];
    const size_t i0 = ab + (size_t)vi.x * A, i1 = ab + (size_t)vi.y * A, i2 = ab + (size_t)vi.z * A;
    const float u = r.x, v = r.y, w2 = 1.f - u - v;
    float gu = 0.f, gv = 0.f;
    for (int a = 0; a < A; a++) {
        const float g = dy[gid * A + a];
        if (dattr && g != 0.f) { atomicAdd(&dattr[i0 + a], u * g); atomicAdd(&dattr[i1 + a], v * g); atomicAdd(&dattr[i2 + a], w2 * g); }
        gu += g * (attr[i0 + a] - attr[i2 + a]); gv += g * (attr[i1 + a] - attr[i2 + a]);
    }
    drast[gid] = make_float4(gu, gv, 0.f, 0.f);
}

// backward of the pixel differentials: out_da[k] = ((a0 - a2) db.x + (a1 - a2) db.z, (a0 - a2) db.y + (a1 - a2) db.w) for attribute diff[k]
__global__ void __launch_bounds__(256) k_interp_da_bwd(const float* __restrict__ attr, int Ba, const float4* __restrict__ rast, const int3* __restrict__ tri,
                                                        const float4* __restrict__ rast_db, const int* __restrict__ diff, int nd, const float* __restrict__ dout_da,
                                                        long long BP, long long P, int V, int A, float* __restrict__ dattr, float4* __restrict__ drast_db) {
    const long long gid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= BP) return;
    const int t = (int)rast[gid].w - 1;
    if (t < 0) { drast_db[gid] = make_float4(0.f, 0.f, 0.f, 0.f); return; }
    const size_t ab = (size_t)(Ba > 1 ? (gid / P) : 0) * V * A;
    const int3 vi = tri[t];
    const size_t i0 = ab + (size_t)vi.x * A, i1 = ab + (size_t)vi.y * A, i2 = ab + (size_t)vi.z * A;
    const float4 db = rast_db[gid];
    float4 g = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int k = 0; k < nd; k++) {
        const int a = diff[k];
        const float gx = dout_da[(gid * nd + k) * 2], gy = dout_da[(gid * nd + k) * 2 + 1];
        const float dsdu = attr[i0 + a] - attr[i2 + a], dsdv = attr[i1 + a] - attr[i2 + a];
        const float gu = gx * db.x + gy * db.y, gv = gx * db.z + gy * db.w;
        if (dattr && (gu != 0.f || gv != 0.f)) { atomicAdd(&dattr[i0 + a], gu); atomicAdd(&dattr[i1 + a], gv); atomicAdd(&dattr[i2 + a], -(gu + gv)); }
        g.x += gx * dsdu; g.y += gy * dsdu; g.z += gx * dsdv; g.w += gy * dsdv;
    }
    drast_db[gid] = g;
}

// ------------------------------------------------------------------------------------------ texture
__device__ __forceinline__ int wrapi(int i, int n, int boundary) {
    if (boundary == 0) { i %= n; if (i < 0) i += n; return i; }
    return min(max(i, 0), n - 1);
}
__global__ void __launch_bounds__(256) k_tex_fwd(const float* __restrict__ tex, int Bt, const float2* __restrict__ uv, long long BP, long long P,
                                                  int Ht, int Wt, int C, int filter, int boundary, float* __restrict__ out) {
    const long long gid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= BP) return;
    const float* tb = tex + (size_t)(Bt > 1 ? (gid / P) : 0) * Ht * Wt * C;
    const float2 q = uv[gid];
    float u = q.x * Wt, v = q.y * Ht;
    float* po = out + gid * C;
    if (filter == 0) {
        const int iu = wrapi((int)floorf(u), Wt, boundary), iv = wrapi((int)floorf(v), Ht, boundary);
        for (int c = 0; c < C; c++) po[c] = tb[((size_t)iv * Wt + iu) * C + c];
        return;
    }
    u -= 0.5f; v -= 0.5f;
    const float fu0 = floorf(u), fv0 = floorf(v), fu = u - fu0, fv = v - fv0;
    const int iu0 = wrapi((int)fu0, Wt, boundary), iu1 = wrapi((int)fu0 + 1, Wt, boundary);
    const int iv0 = wrapi((int)fv0, Ht, boundary), iv1 = wrapi((int)fv0 + 1, Ht, boundary);
    const float *t00 = tb + ((size_t)iv0 * Wt + iu0) * C, *t10 = tb + ((size_t)iv0 * Wt + iu1) * C;
    const float *t01 = tb + ((size_t)iv1 * Wt + iu0) * C, *t11 = tb + ((size_t)iv1 * Wt + iu1) * C;
    for (int c = 0; c < C; c++) {
        const float top = t00[c] + fu * (t10[c] - t00[c]), bot = t01[c] + fu * (t11[c] - t01[c]);
        po[c] = top + fv * (bot - top);
    }
}
__global__ void __launch_bounds__(256) k_tex_bwd(const float* __restrict__ tex, int Bt, const float2* __restrict__ uv, const float* __restrict__ dy,
                                                  long long BP, long long P, int Ht, int Wt, int C, int filter, int boundary, float* __restrict__ dtex,
                                                  float2* __restrict__ duv) {
    const long long gid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= BP) return;
    const size_t tbo = (size_t)(Bt > 1 ? (gid / P) : 0) * Ht * Wt * C;
    const float2 q = uv[gid];
    float u = q.x * Wt, v = q.y * Ht;
    const float* g = dy + gid * C;
    if (filter == 0) {
        const int iu = wrapi((int)floorf(u), Wt, boundary), iv = wrapi((int)floorf(v), Ht, boundary);
        for (int c = 0; c < C; c++) if (g[c] != 0.f) atomicAdd(&dtex[tbo + ((size_t)iv * Wt + iu) * C + c], g[c]);
        duv[gid] = make_float2(0.f, 0.f);
        return;
    }
    u -= 0.5f; v -= 0.5f;
    const float fu0 = floorf(u), fv0 = floorf(v), fu = u - fu0, fv = v - fv0;
    const int iu0 = wrapi((int)fu0, Wt, boundary), iu1 = wrapi((int)fu0 + 1, Wt, boundary);
    const int iv0 = wrapi((int)fv0, Ht, boundary), iv1 = wrapi((int)fv0 + 1, Ht, boundary);
    const size_t i00 = tbo + ((size_t)iv0 * Wt + iu0) * C, i10 = tbo + ((size_t)iv0 * Wt + iu1) * C;
    const size_t i01 = tbo + ((size_t)iv1 * Wt + iu0) * C, i11 = tbo + ((size_t)iv1 * Wt + iu1) * C;
    float gu = 0.f, gv = 0.f;
    for (int c = 0; c < C; c++) {
        const float gc = g[c];
        const float t00 = tex[i00 + c], t10 = tex[i10 + c], t01 = tex[i01 + c], t11 = tex[i11 + c];
        if (gc != 0.f) {
            atomicAdd(&dtex[i00 + c], gc * (1.f - fu) * (1.f - fv)); atomicAdd(&dtex[i10 + c], gc * fu * (1.f - fv));
            atomicAdd(&dtex[i01 + c], gc * (1.f - fu) * fv); atomicAdd(&dtex[i11 + c], gc * fu * fv);
        }
        gu += gc * ((t10 - t00) * (1.f - fv) + (t11 - t01) * fv);
        gv += gc * ((t01 - t00) * (1.f - fu) + (t11 - t10) * fu);
    }
    duv[gid] = make_float2(gu * Wt, gv * Ht);
}

// Linear-filter texture gradient with the scatter pre-combined per 16x16-pixel tile.  Device-scope float atomics are executed beyond the
// XCD-private L2 at a packet rate of ~10-20 G/s on this chip (profiles/microbench/xcd_atomics.hip), and the 12 atomics a pixel issues
// (4 taps x 3 channels) land on the same few texel lines as its neighbours': 0.27 ms per 1024^2 view, 0.02 ms of it arithmetic.  A tile
// first sums its taps per texel in an LDS hash table (LDS float atomics), then issues one global atomic per (distinct texel, channel).
#define TEXT_SLOTS 1024
#define TEXT_EMPTY 0xFFFFFFFFu
// Pre-combining pays only where neighbouring pixels hit the same texels (magnification, coarse mip levels).  Under minification every tap
// is its own texel: the table saturates (256 pixels x 4 taps = its 1024 slots) and the probing costs 3-4x the plain scatter (2048^2 texture
// at 3 texels / pixel: 0.64 ms hashed).  A wave decides for itself: does a lane share any of its four taps with its right-hand neighbour?
// Fewer than a quarter do -> scatter directly.  (Heuristic only: both routes add the same values.)
__device__ __forceinline__ bool tex_taps_shared(const uint32_t tk[4]) {
    uint32_t nk[4];
#pragma unroll
    for (int i = 0; i < 4; i++) nk[i] = (uint32_t)__shfl_down((int)tk[i], 1);
    bool share = false;
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
        for (int j = 0; j < 4; j++) share = share || nk[i] == tk[j];
    return 4 * __popcll(__ballot(share)) >= __popcll(__ballot(true));
}
template <int C>
__global__ void __launch_bounds__(256) k_tex_bwd_tiled(const float* __restrict__ tex, int Bt, const float2* __restrict__ uv, const float* __restrict__ dy,
                                                        int H, int W, int Ht, int Wt, int boundary, float* __restrict__ dtex, float2* __restrict__ duv) {
    __shared__ uint32_t keys[TEXT_SLOTS];
    __shared__ float vals[TEXT_SLOTS][C];
    for (int i = threadIdx.x; i < TEXT_SLOTS; i += 256) {
        keys[i] = TEXT_EMPTY;
#pragma unroll
        for (int c = 0; c < C; c++) vals[i][c] = 0.f;
    }
    __syncthreads();
    const int b = blockIdx.z;
    const int px = blockIdx.x * 16 + (threadIdx.x & 15), py = blockIdx.y * 16 + (threadIdx.x >> 4);
    const size_t tbo = (size_t)(Bt > 1 ? b : 0) * Ht * Wt * C;
    if (px < W && py < H) {
        const size_t gid = ((size_t)b * H + py) * W + px;
        const float2 q = uv[gid];
        const float u = q.x * Wt - 0.5f, v = q.y * Ht - 0.5f;
        const float fu0 = floorf(u), fv0 = floorf(v), fu = u - fu0, fv = v - fv0;
        const int iu0 = wrapi((int)fu0, Wt, boundary), iu1 = wrapi((int)fu0 + 1, Wt, boundary);
        const int iv0 = wrapi((int)fv0, Ht, boundary), iv1 = wrapi((int)fv0 + 1, Ht, boundary);
        const uint32_t tk[4] = {(uint32_t)(iv0 * Wt + iu0), (uint32_t)(iv0 * Wt + iu1), (uint32_t)(iv1 * Wt + iu0), (uint32_t)(iv1 * Wt + iu1)};
        const float tw[4] = {(1.f - fu) * (1.f - fv), fu * (1.f - fv), (1.f - fu) * fv, fu * fv};
        float g[C];
        bool any = false;
#pragma unroll
        for (int c = 0; c < C; c++) {
            g[c] = dy[gid * C + c];
            any = any || g[c] != 0.f;
        }
        float gu = 0.f, gv = 0.f;
#pragma unroll
        for (int c = 0; c < C; c++) {
            const float t00 = tex[tbo + (size_t)tk[0] * C + c], t10 = tex[tbo + (size_t)tk[1] * C + c];
            const float t01 = tex[tbo + (size_t)tk[2] * C + c], t11 = tex[tbo + (size_t)tk[3] * C + c];
            gu += g[c] * ((t10 - t00) * (1.f - fv) + (t11 - t01) * fv);
            gv += g[c] * ((t01 - t00) * (1.f - fu) + (t11 - t10) * fu);
        }
        duv[gid] = make_float2(gu * Wt, gv * Ht);
        const bool use_hash = tex_taps_shared(tk);
        if (any) {
#pragma unroll
            for (int t = 0; t < 4; t++) {
                uint32_t h = (tk[t] * 2654435761u) >> 22;           // 10 bits
                int slot = -1;
                for (int probe = 0; use_hash && probe < 32; probe++) {   // bounded: a full table falls back to the direct scatter
                    const uint32_t old = atomicCAS(&keys[h], TEXT_EMPTY, tk[t]);
                    if (old == TEXT_EMPTY || old == tk[t]) { slot = (int)h; break; }
                    h = (h + 1) & (TEXT_SLOTS - 1);
                }
#pragma unroll
                for (int c = 0; c < C; c++) {
                    const float w = g[c] * tw[t];
                    if (w == 0.f) continue;
                    if (slot >= 0) atomicAdd(&vals[slot][c], w);
                    else atomicAdd(&dtex[tbo + (size_t)tk[t] * C + c], w);
                }
            }
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < TEXT_SLOTS; i += 256) {
        const uint32_t k = keys[i];
        if (k == TEXT_EMPTY) continue;
#pragma unroll
        for (int c = 0; c < C; c++) {
            const float v = vals[i][c];
            if (v != 0.f) atomicAdd(&dtex[tbo + (size_t)k * C + c], v);
        }
    }
}

// ------------------------------------------------------------------------------------------ antialias
// ------------------------------------------------------------------------------------------ mip-mapped texture
// 'linear-mipmap-linear' / 'linear-mipmap-nearest' (what `dr.texture(tex, uv, uv_da)` selects under filter_mode='auto').  The pyramid is the
// base texture plus one packed buffer `stack` = levels 1..L back to back per batch item; level l is max(Ht >> l, 1) x max(Wt >> l, 1) (the
// host refuses pyramids that would have to halve an odd extent), so kernels derive every level's shape and offset from (Ht, Wt) alone --
// no per-lane table lookups.  A texel of the pyramid is named by one 32-bit "virtual" index: [0, Ht*Wt) = base, Ht*Wt + off_l + ... = stack.
#define MIP_MAX 16
struct MipShape { int L; long long total; int w[MIP_MAX + 1], h[MIP_MAX + 1]; long long off[MIP_MAX + 1]; };
static int mip_shape(int Ht, int Wt, int max_level, MipShape& m) {
    m.L = 0; m.total = 0; m.w[0] = Wt; m.h[0] = Ht; m.off[0] = 0;
    int w = Wt, h = Ht;
    while ((w > 1 || h > 1) && m.L < MIP_MAX && (max_level < 0 || m.L < max_level)) {
        if ((w > 1 && (w & 1)) || (h > 1 && (h & 1))) return -1;
        if (w > 1) w >>= 1;
        if (h > 1) h >>= 1;
        m.L++; m.w[m.L] = w; m.h[m.L] = h; m.off[m.L] = m.total; m.total += (long long)w * h;
    }
    return 0;
}
__device__ __forceinline__ int mip_w(int Wt, int l) { return max(Wt >> l, 1); }
// texels of levels 1 .. l-1 (offset of level l inside the stack)
__device__ __forceinline__ uint32_t mip_off(int Ht, int Wt, int l) {
    uint32_t o = 0;
    for (int k = 1; k < l; k++) o += (uint32_t)mip_w(Wt, k) * (uint32_t)mip_w(Ht, k);
    return o;
}
// level selection from the pixel footprint: (du/dX, du/dY, dv/dX, dv/dY) scaled to texels, level = log2(major axis); + bias; clamp to [0, L]
__device__ __forceinline__ int mip_select(const float4* __restrict__ da, const float* __restrict__ bias, long long gid, int Ht, int Wt, int L, int filter,
                                          int& l1, float& f) {
    float fl = 0.f;
    if (da) {
        const float4 d = da[gid];
        const float dsdx = d.x * Wt, dsdy = d.y * Wt, dtdx = d.z * Ht, dtdy = d.w * Ht;
        const float A = dsdx * dsdx + dtdx * dtdx, Bq = dsdy * dsdy + dtdy * dtdy, Cq = dsdx * dsdy + dtdx * dtdy;
        const float l2b = 0.5f * (A + Bq), l2n = 0.25f * (A - Bq) * (A - Bq) + Cq * Cq;
        fl = 0.5f * log2f(l2b + sqrtf(l2n));
    }
    if (bias) fl += bias[gid];
    if (!(fl > 0.f)) fl = 0.f;                     // also NaN, -inf
    fl = fminf(fl, (float)L);
    if (filter == 2) { const int l0 = min((int)floorf(fl + 0.5f), L); l1 = l0; f = 0.f; return l0; }
    const int l0 = min((int)floorf(fl), L);
    l1 = min(l0 + 1, L);
    f = fl - (float)l0;
    return l0;
}
// d(level) / d(uv_da) where the level lies strictly inside (0, L) before clamping (mirror of mip_select).
// -> true when inside: gda = dlevel * d level / d uv_da, and the bias receives dlevel itself.
__device__ __forceinline__ bool mip_level_grad(const float4* __restrict__ da, const float* __restrict__ bias, long long gid, int Ht, int Wt, int L, float dlevel, float4& gda) {
    float fl = 0.f, major = 1.f, A = 0.f, Bq = 0.f, Cq = 0.f, sq = 0.f, dsdx = 0.f, dsdy = 0.f, dtdx = 0.f, dtdy = 0.f;
    gda = make_float4(0.f, 0.f, 0.f, 0.f);
    if (da) {
        const float4 d = da[gid];
        dsdx = d.x * Wt; dsdy = d.y * Wt; dtdx = d.z * Ht; dtdy = d.w * Ht;
        A = dsdx * dsdx + dtdx * dtdx; Bq = dsdy * dsdy + dtdy * dtdy; Cq = dsdx * dsdy + dtdx * dtdy;
        sq = sqrtf(0.25f * (A - Bq) * (A - Bq) + Cq * Cq);
        major = 0.5f * (A + Bq) + sq;
        fl = 0.5f * log2f(major);
    }
    if (bias) fl += bias[gid];
    if (!(fl > 0.f) || !(fl < (float)L)) return false;
    if (da) {
        const float gm = dlevel * 0.5f / (major * 0.6931471805599453f);
        const float t = sq > 0.f ? 0.25f * (A - Bq) / sq : 0.f, gA = gm * (0.5f + t), gB = gm * (0.5f - t), gC = sq > 0.f ? gm * Cq / sq : 0.f;
        gda = make_float4((2.f * dsdx * gA + dsdy * gC) * Wt, (2.f * dsdy * gB + dsdx * gC) * Wt, (2.f * dtdx * gA + dtdy * gC) * Ht, (2.f * dtdy * gB + dtdx * gC) * Ht);
    }
    return true;
}
// the four bilinear taps of one level as virtual texel indices + weights
// ok: bit k = tap k lies inside the level (always under wrap / clamp; boundary 2 = 'zero': a tap outside reads 0 and receives no gradient -- its index is parked
// on the level's first texel)
struct MipTaps { uint32_t k[4]; float fu, fv; int w, h; uint32_t ok; };
__device__ __forceinline__ MipTaps mip_taps(float2 q, int Ht, int Wt, int l, int boundary) {
    MipTaps t;
    t.w = mip_w(Wt, l); t.h = mip_w(Ht, l);
    const uint32_t base = l == 0 ? 0u : (uint32_t)Ht * (uint32_t)Wt + mip_off(Ht, Wt, l);
    const float u = q.x * t.w - 0.5f, v = q.y * t.h - 0.5f;
    const float fu0 = floorf(u), fv0 = floorf(v);
    t.fu = u - fu0; t.fv = v - fv0;
    int iu0, iu1, iv0, iv1;
    t.ok = 15u;
    if (boundary == 2) {
        iu0 = (int)fu0; iu1 = iu0 + 1; iv0 = (int)fv0; iv1 = iv0 + 1;
        const bool x0 = iu0 >= 0 && iu0 < t.w, x1 = iu1 >= 0 && iu1 < t.w, y0 = iv0 >= 0 && iv0 < t.h, y1 = iv1 >= 0 && iv1 < t.h;
        t.ok = (x0 && y0 ? 1u : 0u) | (x1 && y0 ? 2u : 0u) | (x0 && y1 ? 4u : 0u) | (x1 && y1 ? 8u : 0u);
        if (!x0) iu0 = 0; if (!x1) iu1 = 0; if (!y0) iv0 = 0; if (!y1) iv1 = 0;
    } else {
        iu0 = wrapi((int)fu0, t.w, boundary); iu1 = wrapi((int)fu0 + 1, t.w, boundary);
        iv0 = wrapi((int)fv0, t.h, boundary); iv1 = wrapi((int)fv0 + 1, t.h, boundary);
    }
    t.k[0] = base + (uint32_t)(iv0 * t.w + iu0); t.k[1] = base + (uint32_t)(iv0 * t.w + iu1);
    t.k[2] = base + (uint32_t)(iv1 * t.w + iu0); t.k[3] = base + (uint32_t)(iv1 * t.w + iu1);
    return t;
}
#define MIP_V(t, k, p, c) (((t).ok >> (k)) & 1u ? (p)[c] : 0.f)
// address of channel 0 of virtual texel k: tb = this batch item's base texture, sb = its stack, HW = Ht*Wt
template <typename T>
__device__ __forceinline__ T* mip_texel(T* tb, T* sb, uint32_t HW, uint32_t k, int C) {
    return k < HW ? tb + (size_t)k * C : sb + (size_t)(k - HW) * C;
}

__global__ void __launch_bounds__(256) k_mip_down(const float* __restrict__ src, float* __restrict__ dst, int Bt, int hs, int ws, int hd, int wd, int C,
                                                   size_t src_bstride, size_t dst_bstride) {
    const long long n = (long long)Bt * hd * wd * C, gid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= n) return;
    const int c = (int)(gid % C); long long r = gid / C;
    const int x = (int)(r % wd); r /= wd;
    const int y = (int)(r % hd); const int b = (int)(r / hd);
    const int sx = ws > 1 ? 2 : 1, sy = hs > 1 ? 2 : 1;
    const float* sp = src + (size_t)b * src_bstride;
    float a = 0.f;
    for (int j = 0; j < sy; j++) for (int i = 0; i < sx; i++) a += sp[((size_t)(y * sy + j) * ws + (x * sx + i)) * C + c];
    dst[(size_t)b * dst_bstride + ((size_t)y * wd + x) * C + c] = a / (float)(sx * sy);
}
// transpose of k_mip_down: fine[b][y][x][c] += coarse[b][y / sy][x / sx][c] / (sx * sy)
__global__ void __launch_bounds__(256) k_mip_fold(const float* __restrict__ coarse, float* __restrict__ fine, int Bt, int hf, int wf, int hc, int wc, int C,
                                                   size_t coarse_bstride, size_t fine_bstride) {
    (void)hc;
    const long long n = (long long)Bt * hf * wf * C, gid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= n) return;
    const int c = (int)(gid % C); long long r = gid / C;
    const int x = (int)(r % wf); r /= wf;
    const int y = (int)(r % hf); const int b = (int)(r / hf);
    const int sx = wf > 1 ? 2 : 1, sy = hf > 1 ? 2 : 1;
    fine[(size_t)b * fine_bstride + ((size_t)y * wf + x) * C + c] +=
        coarse[(size_t)b * coarse_bstride + ((size_t)(y / sy) * wc + (x / sx)) * C + c] / (float)(sx * sy);
}

__global__ void __launch_bounds__(256) k_tex_mip_fwd(const float* __restrict__ tex, const float* __restrict__ stack, int Bt, const float2* __restrict__ uv,
                                                      const float4* __restrict__ da, const float* __restrict__ bias, long long BP, long long P, int Ht, int Wt,
                                                      int C, int L, long long total, int filter, int boundary, float* __restrict__ out) {
    const long long gid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= BP) return;
    const size_t bt = Bt > 1 ? (size_t)(gid / P) : 0;
    const uint32_t HW = (uint32_t)Ht * (uint32_t)Wt;
    const float* tb = tex + bt * HW * C;
    const float* sb = stack + bt * (size_t)total * C;
    int l1; float f;
    const int l0 = mip_select(da, bias, gid, Ht, Wt, L, filter, l1, f);
    const float2 q = uv[gid];
    const MipTaps a = mip_taps(q, Ht, Wt, l0, boundary);
    const float *a00 = mip_texel(tb, sb, HW, a.k[0], C), *a10 = mip_texel(tb, sb, HW, a.k[1], C);
    const float *a01 = mip_texel(tb, sb, HW, a.k[2], C), *a11 = mip_texel(tb, sb, HW, a.k[3], C);
    float* po = out + gid * C;
    const bool two = l1 != l0 && f != 0.f;
    if (!two) {
        for (int c = 0; c < C; c++) {
            const float v00 = MIP_V(a, 0, a00, c), v10 = MIP_V(a, 1, a10, c), v01 = MIP_V(a, 2, a01, c), v11 = MIP_V(a, 3, a11, c);
            const float top = v00 + a.fu * (v10 - v00), bot = v01 + a.fu * (v11 - v01);
            po[c] = top + a.fv * (bot - top);
        }
        return;
    }
    const MipTaps b = mip_taps(q, Ht, Wt, l1, boundary);
    const float *b00 = mip_texel(tb, sb, HW, b.k[0], C), *b10 = mip_texel(tb, sb, HW, b.k[1], C);
    const float *b01 = mip_texel(tb, sb, HW, b.k[2], C), *b11 = mip_texel(tb, sb, HW, b.k[3], C);
    for (int c = 0; c < C; c++) {
        const float v00 = MIP_V(a, 0, a00, c), v10 = MIP_V(a, 1, a10, c), v01 = MIP_V(a, 2, a01, c), v11 = MIP_V(a, 3, a11, c);
        const float w00 = MIP_V(b, 0, b00, c), w10 = MIP_V(b, 1, b10, c), w01 = MIP_V(b, 2, b01, c), w11 = MIP_V(b, 3, b11, c);
        const float ta = v00 + a.fu * (v10 - v00), ba = v01 + a.fu * (v11 - v01);
        const float tb_ = w00 + b.fu * (w10 - w00), bb = w01 + b.fu * (w11 - w01);
        po[c] = (1.f - f) * (ta + a.fv * (ba - ta)) + f * (tb_ + b.fv * (bb - tb_));
    }
}

// Gradient of the mip-mapped fetch.  Same structure as k_tex_bwd_tiled: a 16x16-pixel tile sums its (up to 8) taps per virtual texel in an
// LDS hash table and issues one global atomic per distinct (texel, channel); coarse levels, where a whole tile lands on a handful of
// texels, are exactly where the pre-combination pays most.  CT = 0: any channel count, direct global atomics.
#define MIPT_SLOTS 2048
template <int CT>
__global__ void __launch_bounds__(256) k_tex_mip_bwd(const float* __restrict__ tex, const float* __restrict__ stack, int Bt, const float2* __restrict__ uv,
                                                      const float4* __restrict__ da, const float* __restrict__ bias, const float* __restrict__ dy, int H, int W,
                                                      int Ht, int Wt, int Crt, int L, long long total, int filter, int boundary, float* __restrict__ dtex,
                                                      float* __restrict__ dstack, float2* __restrict__ duv, float4* __restrict__ dda, float* __restrict__ dbias) {
    constexpr int CS = CT > 0 ? CT : 1;
    __shared__ uint32_t keys[CT > 0 ? MIPT_SLOTS : 1];
    __shared__ float vals[CT > 0 ? MIPT_SLOTS : 1][CS];
    const int C = CT > 0 ? CT : Crt;
    if (CT > 0) {
        for (int i = threadIdx.x; i < MIPT_SLOTS; i += 256) {
            keys[i] = TEXT_EMPTY;
#pragma unroll
            for (int c = 0; c < CS; c++) vals[i][c] = 0.f;
        }
        __syncthreads();
    }
    const int b = blockIdx.z;
    const int px = blockIdx.x * 16 + (threadIdx.x & 15), py = blockIdx.y * 16 + (threadIdx.x >> 4);
    const size_t bt = Bt > 1 ? (size_t)b : 0;
    const uint32_t HW = (uint32_t)Ht * (uint32_t)Wt;
    const float* tb = tex + bt * HW * C;
    const float* sb = stack + bt * (size_t)total * C;
    float* dtb = dtex + bt * HW * C;
    float* dsb = dstack + bt * (size_t)total * C;
    if (px < W && py < H) {
        const long long gid = ((long long)b * H + py) * W + px;
        int l1; float f;
        const int l0 = mip_select(da, bias, gid, Ht, Wt, L, filter, l1, f);
        const float2 q = uv[gid];
        const float* g = dy + gid * C;
        float gu = 0.f, gv = 0.f;
        float sdot[2] = {0.f, 0.f};
        const int nl = (l1 != l0 && f != 0.f) ? 2 : 1;
        const MipTaps t0 = mip_taps(q, Ht, Wt, l0, boundary);
        const bool use_hash = CT > 0 && tex_taps_shared(t0.k);        // decided on the finer level: the coarser one shares at least as much
        for (int lev = 0; lev < nl; lev++) {
            const MipTaps t = lev ? mip_taps(q, Ht, Wt, l1, boundary) : t0;
            const float wl = nl == 1 ? 1.f : (lev ? f : 1.f - f);
            const float tw[4] = {(1.f - t.fu) * (1.f - t.fv), t.fu * (1.f - t.fv), (1.f - t.fu) * t.fv, t.fu * t.fv};
            const float *p00 = mip_texel(tb, sb, HW, t.k[0], C), *p10 = mip_texel(tb, sb, HW, t.k[1], C);
            const float *p01 = mip_texel(tb, sb, HW, t.k[2], C), *p11 = mip_texel(tb, sb, HW, t.k[3], C);
            float lu = 0.f, lv = 0.f, sd = 0.f;
            bool any = false;
            for (int c = 0; c < C; c++) {
                const float gc = g[c] * wl;
                any = any || gc != 0.f;
                const float v00 = MIP_V(t, 0, p00, c), v10 = MIP_V(t, 1, p10, c), v01 = MIP_V(t, 2, p01, c), v11 = MIP_V(t, 3, p11, c);
                lu += gc * ((v10 - v00) * (1.f - t.fv) + (v11 - v01) * t.fv);
                lv += gc * ((v01 - v00) * (1.f - t.fu) + (v11 - v10) * t.fu);
                if (dda || dbias) {                                   // dy . sample(level): the level gradient is the difference of the two levels' dots
                    const float top = v00 + t.fu * (v10 - v00), bot = v01 + t.fu * (v11 - v01);
                    sd += g[c] * (top + t.fv * (bot - top));
                }
            }
            sdot[lev] = sd;
            gu += lu * t.w; gv += lv * t.h;
            if (!any) continue;
#pragma unroll
            for (int k = 0; k < 4; k++) {
                if (!((t.ok >> k) & 1u)) continue;                   // 'zero': nothing behind this tap
                int slot = -1;
                if (use_hash) {
                    uint32_t h = (t.k[k] * 2654435761u) >> 21;       // 11 bits
                    for (int probe = 0; probe < 32; probe++) {       // bounded: a full table falls back to the direct scatter
                        const uint32_t old = atomicCAS(&keys[h], TEXT_EMPTY, t.k[k]);
                        if (old == TEXT_EMPTY || old == t.k[k]) { slot = (int)h; break; }
                        h = (h + 1) & (MIPT_SLOTS - 1);
                    }
                }
                float* dp = mip_texel(dtb, dsb, HW, t.k[k], C);
                for (int c = 0; c < C; c++) {
                    const float w = g[c] * wl * tw[k];
                    if (w == 0.f) continue;
                    if (slot >= 0) atomicAdd(&vals[slot][CT > 0 ? c : 0], w);
                    else atomicAdd(&dp[c], w);
                }
            }
        }
        duv[gid] = make_float2(gu, gv);
        if (dda || dbias) {      // gradients w.r.t. uv_da / mip_level_bias: only 'linear-mipmap-linear' (3) with two distinct levels and an unclamped level
            float4 gda;
            const float dl = sdot[1] - sdot[0];
            const bool in = filter == 3 && nl == 2 && mip_level_grad(da, bias, gid, Ht, Wt, L, dl, gda);
            if (dda) dda[gid] = in ? gda : make_float4(0.f, 0.f, 0.f, 0.f);
            if (dbias) dbias[gid] = in ? dl : 0.f;
        }
    }
    if (CT > 0) {
        __syncthreads();
        for (int i = threadIdx.x; i < MIPT_SLOTS; i += 256) {
            const uint32_t k = keys[i];
            if (k == TEXT_EMPTY) continue;
            float* dp = mip_texel(dtb, dsb, HW, k, C);
#pragma unroll
            for (int c = 0; c < CS; c++) {
                const float v = vals[i][c];
                if (v != 0.f) atomicAdd(&dp[c], v);
            }
        }
    }
}

// Topology hash: open addressing on the 64-bit (min vertex, max vertex) key; each slot records up to two incident
// (triangle, opposite vertex) pairs and the total count.
struct EdgeSlot { unsigned long long key; uint32_t count; int32_t tri0, opp0, tri1, opp1; uint32_t pad; };   // 32 bytes
__device__ __forceinline__ uint32_t edge_hash(unsigned long long k, uint32_t mask) {
    k ^= k >> 33; k *= 0xff51afd7ed558ccdull; k ^= k >> 33; k *= 0xc4ceb9fe1a85ec53ull; k ^= k >> 33;
    return (uint32_t)k & mask;
}
static inline uint32_t edge_table_size(int T) {
    uint32_t need = (uint32_t)(T > 0 ? T : 1) * 3u, n = 1024;   // <= 1.5 T distinct edges on a manifold mesh: load factor <= 0.5
    while (n < need) n <<= 1;
    return n;
}
__global__ void __launch_bounds__(256) k_aa_hash_init(EdgeSlot* __restrict__ table, uint32_t n) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) { table[i].key = MESH_EMPTY_KEY; table[i].count = 0u; }
}
__global__ void __launch_bounds__(256) k_aa_hash_build(const int3* __restrict__ tri, int T, EdgeSlot* __restrict__ table, uint32_t mask) {
    const int gid = blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= 3 * T) return;
    const int t = gid / 3, k = gid % 3;
    const int3 vi = tri[t];
    const int vv[3] = {vi.x, vi.y, vi.z};
    const int a = vv[k], b = vv[(k + 1) % 3], o = vv[(k + 2) % 3];
    if (a == b) return;
    const unsigned long long key = ((unsigned long long)(uint32_t)min(a, b) << 32) | (uint32_t)max(a, b);
    uint32_t h = edge_hash(key, mask);
    for (uint32_t probe = 0; probe <= mask; probe++, h = (h + 1) & mask) {
        const unsigned long long old = atomicCAS(&table[h].key, MESH_EMPTY_KEY, key);
        if (old == MESH_EMPTY_KEY || old == key) {
            const uint32_t i = atomicAdd(&table[h].count, 1u);
            if (i == 0) { table[h].tri0 = t; table[h].opp0 = o; }
            else if (i == 1) { table[h].tri1 = t; table[h].opp1 = o; }
            return;
        }
    }
}
// opposite vertex of the other triangle on edge (a,b) of `t`: -1 boundary, -2 non-manifold (> 2 triangles)
__device__ __forceinline__ int other_opposite(const EdgeSlot* __restrict__ table, uint32_t mask, int a, int b, int t) {
    const unsigned long long key = ((unsigned long long)(uint32_t)min(a, b) << 32) | (uint32_t)max(a, b);
    uint32_t h = edge_hash(key, mask);
    for (uint32_t probe = 0; probe <= mask; probe++, h = (h + 1) & mask) {
        const unsigned long long kk = table[h].key;
        if (kk == key) {
            const uint32_t c = table[h].count;
            if (c > 2) return -2;
            if (c < 2) return -1;
            // with two entries: the one that is not `t`; if the lower-index one is t, the other; order-independent
            const int t0 = table[h].tri0, t1 = table[h].tri1;
            if (t0 != t && t1 != t) return (t0 < t1) ? table[h].opp0 : table[h].opp1;
            return (t0 != t) ? table[h].opp0 : ((t1 != t) ? table[h].opp1 : -1);
        }
        if (kk == MESH_EMPTY_KEY) return -1;
    }
    return -1;
}

// Round 3: the answer of other_opposite() for every (triangle, edge), computed once per topology behind the hash table (int32 [T][3]): the per-view silhouette
// analysis reads one coalesced triple per triangle instead of probing the hash (a random 32-byte gather + key compare per crossing edge).
__global__ void __launch_bounds__(256) k_aa_adj_build(const int3* __restrict__ tri, int T, const EdgeSlot* __restrict__ table, uint32_t mask, int* __restrict__ adj) {
    const int gid = blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= 3 * T) return;
    const int t = gid / 3, k = gid % 3;
    const int3 vi = tri[t];
    const int vv[3] = {vi.x, vi.y, vi.z};
    adj[gid] = other_opposite(table, mask, vv[k], vv[(k + 1) % 3], t);
}
__device__ __forceinline__ const int* aa_adjacency(const EdgeSlot* table, uint32_t mask) { return (const int*)(table + (size_t)mask + 1); }
struct AaHit { int ax, ay, bx, by, va, vb, ek; float s, sgn; };      // ek: the silhouette edge runs from corner ek to corner (ek + 1) % 3 of pixel a's triangle
// sil_b (optional, the fused view): per triangle of THIS view, bit k = edge k is a silhouette edge (k_aa_sil_bits below, one lane per triangle, once per view).  On a mesh
// of ~2-pixel triangles nearly every pixel pair has two different ids and ~1 % of them straddle a silhouette: with the bits a pair whose nearer triangle has no
// silhouette edge costs one byte gather instead of its index triple, three vertex positions and the crossing tests.  Same decisions, same bits.
__device__ __forceinline__ bool aa_analyze(const float4* __restrict__ pb, const int3* __restrict__ tri, const EdgeSlot* __restrict__ table, uint32_t mask,
                                           const float4* __restrict__ rast_b, int H, int W, int px, int py, int d, AaHit& hit, const uint8_t* __restrict__ sil_b = nullptr) {
    const int qx = px + (d == 0), qy = py + (d == 1);
    if (qx >= W || qy >= H) return false;
    // only (z/w, id) of the two pixels: the upper half of each 16-byte rast entry
    const float2 r0 = reinterpret_cast<const float2*>(rast_b + ((size_t)py * W + px))[1], r1 = reinterpret_cast<const float2*>(rast_b + ((size_t)qy * W + qx))[1];
    const int id0 = (int)r0.y, id1 = (int)r1.y;
    if (id0 == id1) return false;
    bool a_is_p;
    if (id0 > 0 && id1 > 0) a_is_p = r0.x < r1.x;
    else a_is_p = id0 > 0;
    const int t = (a_is_p ? id0 : id1) - 1;
    uint32_t silb = 7u;
    if (sil_b) { silb = sil_b[t]; if (!silb) return false; }
    const int ax = a_is_p ? px : qx, ay = a_is_p ? py : qy;
    const float sgn = a_is_p ? 1.f : -1.f;
    const int3 vi3 = tri[t];
    const int vi[3] = {vi3.x, vi3.y, vi3.z};
    float nx[3], ny[3];
#pragma unroll
    for (int k = 0; k < 3; k++) {
        const float4 p = pb[vi[k]];
        if (!(p.w > 0.f)) return false;
        nx[k] = p.x / p.w; ny[k] = p.y / p.w;
    }
    const float cx = ((float)ax + 0.5f) * (2.f / W) - 1.f, cy = ((float)ay + 0.5f) * (2.f / H) - 1.f;
    const float h = d == 0 ? 2.f / W : 2.f / H;
    bool found = false;
    float best = 0.f;
#pragma unroll
    for (int k = 0; k < 3; k++) {
        const int ia = k, ib = (k + 1) % 3, io = (k + 2) % 3;
        if (!((silb >> k) & 1u)) continue;
        const float ex = nx[ib] - nx[ia], ey = ny[ib] - ny[ia];
        // cheap geometric test first, hash lookup only for edges that cross the centre-to-centre segment
        float s;
        if (d == 0) {
            const float da = ny[ia] - cy, db = ny[ib] - cy;
            if (!((da <= 0.f && db > 0.f) || (db <= 0.f && da > 0.f))) continue;
            const float te = da / (da - db);
            s = sgn * ((nx[ia] + te * ex) - cx) / h;
        } else {
            const float da = nx[ia] - cx, db = nx[ib] - cx;
            if (!((da <= 0.f && db > 0.f) || (db <= 0.f && da > 0.f))) continue;
            const float te = da / (da - db);
            s = sgn * ((ny[ia] + te * ey) - cy) / h;
        }
        if (!(s >= 0.f && s <= 1.f)) continue;
        const int opp = sil_b ? -1 : aa_adjacency(table, mask)[3 * t + k];      // = other_opposite(table, mask, vi[ia], vi[ib], t); with the bits the edge is already known to be a silhouette
        if (opp == -2) continue;
        if (opp >= 0) {   // interior edge: silhouette only if both triangles lie on the same side of it
            const float4 q = pb[opp];
            if (!(q.w > 0.f)) continue;
            const float ox = q.x / q.w, oy = q.y / q.w;
            const float s_this = ex * (ny[io] - ny[ia]) - ey * (nx[io] - nx[ia]);
            const float s_other = ex * (oy - ny[ia]) - ey * (ox - nx[ia]);
            if (!(s_this * s_other > 0.f)) continue;
        }
        if (!found || s < best) { found = true; best = s; hit.va = vi[ia]; hit.vb = vi[ib]; hit.ek = k; }
    }
    if (!found) return false;
    hit.ax = ax; hit.ay = ay; hit.bx = a_is_p ? qx : px; hit.by = a_is_p ? qy : py; hit.s = best; hit.sgn = sgn;
    return true;
}

// silhouette bits of one view's triangles (blockIdx.y = view): the per-edge part of aa_analyze -- all three vertices in front of the camera, the neighbour across the
// edge absent (boundary) or on the same side of it -- evaluated once per triangle instead of once per pixel pair.  The expressions are aa_analyze's, term by term.
__global__ void __launch_bounds__(256) k_aa_sil_bits(const float4* __restrict__ pos, const int3* __restrict__ tri, const EdgeSlot* __restrict__ table, uint32_t mask, int V, int T,
                                                      uint8_t* __restrict__ sil) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= T) return;
    const float4* __restrict__ pb = pos + (size_t)blockIdx.y * V;
    const int3 vi3 = tri[t];
    const int vi[3] = {vi3.x, vi3.y, vi3.z};
    float nx[3], ny[3];
    bool front = true;
#pragma unroll
    for (int k = 0; k < 3; k++) {
        const float4 p = pb[vi[k]];
        front = front && (p.w > 0.f);
        nx[k] = p.x / p.w; ny[k] = p.y / p.w;
    }
    uint32_t bits = 0;
    if (front) {
#pragma unroll
        for (int k = 0; k < 3; k++) {
            const int ia = k, ib = (k + 1) % 3, io = (k + 2) % 3;
            const float ex = nx[ib] - nx[ia], ey = ny[ib] - ny[ia];
            const int opp = aa_adjacency(table, mask)[3 * t + k];
            if (opp == -2) continue;
            if (opp >= 0) {
                const float4 q = pb[opp];
                if (!(q.w > 0.f)) continue;
                const float ox = q.x / q.w, oy = q.y / q.w;
                const float s_this = ex * (ny[io] - ny[ia]) - ey * (nx[io] - nx[ia]);
                const float s_other = ex * (oy - ny[ia]) - ey * (ox - nx[ia]);
                if (!(s_this * s_other > 0.f)) continue;
            }
            bits |= 1u << k;
        }
    }
    sil[(size_t)blockIdx.y * T + t] = (uint8_t)bits;
}

__global__ void __launch_bounds__(256) k_aa_fwd(const float* __restrict__ color, const float4* __restrict__ rast, const float4* __restrict__ pos,
                                                 const int3* __restrict__ tri, const EdgeSlot* __restrict__ table, uint32_t mask, int B, int V, int H, int W,
                                                 int C, float* __restrict__ out) {
    const long long gid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long P = (long long)H * W;
    if (gid >= (long long)B * P * 2) return;
    const int d = (int)(gid & 1);
    const long long pg = gid >> 1;
    const int b = (int)(pg / P), pid = (int)(pg % P), px = pid % W, py = pid / W;
    AaHit h;
    if (!aa_analyze(pos + (size_t)b * V, tri, table, mask, rast + (size_t)b * P, H, W, px, py, d, h)) return;
    const float alpha = h.s - 0.5f;
    const float* ca = color + ((size_t)b * P + (size_t)h.ay * W + h.ax) * C;
    const float* cb = color + ((size_t)b * P + (size_t)h.by * W + h.bx) * C;
    float* o = out + ((size_t)b * P + (alpha > 0.f ? ((size_t)h.by * W + h.bx) : ((size_t)h.ay * W + h.ax))) * C;
    for (int c = 0; c < C; c++) atomicAdd(&o[c], alpha * (ca[c] - cb[c]));
}
__global__ void __launch_bounds__(256) k_aa_bwd(const float* __restrict__ color, const float4* __restrict__ rast, const float4* __restrict__ pos,
                                                 const int3* __restrict__ tri, const EdgeSlot* __restrict__ table, uint32_t mask, const float* __restrict__ dy,
                                                 int B, int V, int H, int W, int C, float* __restrict__ dcolor, float* __restrict__ dpos) {
    const long long gid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long P = (long long)H * W;
    if (gid >= (long long)B * P * 2) return;
    const int d = (int)(gid & 1);
    const long long pg = gid >> 1;
    const int b = (int)(pg / P), pid = (int)(pg % P), px = pid % W, py = pid / W;
    const float4* pb = pos + (size_t)b * V;
    AaHit h;
    if (!aa_analyze(pb, tri, table, mask, rast + (size_t)b * P, H, W, px, py, d, h)) return;
    const float alpha = h.s - 0.5f;
    const size_t ia = ((size_t)b * P + (size_t)h.ay * W + h.ax) * C, ib = ((size_t)b * P + (size_t)h.by * W + h.bx) * C;
    const size_t idst = alpha > 0.f ? ib : ia;
    float dalpha = 0.f;
    for (int c = 0; c < C; c++) {
        const float g = dy[idst + c];
        atomicAdd(&dcolor[ia + c], alpha * g); atomicAdd(&dcolor[ib + c], -alpha * g);
        dalpha += g * (color[ia + c] - color[ib + c]);
    }
    const float4 pa = pb[h.va], pv = pb[h.vb];
    const float xa = pa.x / pa.w, ya = pa.y / pa.w, xb = pv.x / pv.w, yb = pv.y / pv.w;
    const float cx = ((float)h.ax + 0.5f) * (2.f / W) - 1.f, cy = ((float)h.ay + 0.5f) * (2.f / H) - 1.f;
    const float hh = d == 0 ? 2.f / W : 2.f / H;
    const float gs = dalpha * h.sgn / hh;
    float gxa, gya, gxb, gyb;
    if (d == 0) {
        const float da = ya - cy, db = yb - cy, den = da - db, te = da / den, gte = gs * (xb - xa);
        gxa = gs * (1.f - te); gxb = gs * te;
        gya = gte * (-db / (den * den)); gyb = gte * (da / (den * den));
    } else {
        const float da = xa - cx, db = xb - cx, den = da - db, te = da / den, gte = gs * (yb - ya);
        gya = gs * (1.f - te); gyb = gs * te;
        gxa = gte * (-db / (den * den)); gxb = gte * (da / (den * den));
    }
    float* dA = dpos + ((size_t)b * V + h.va) * 4;
    float* dB = dpos + ((size_t)b * V + h.vb) * 4;
    atomicAdd(dA + 0, gxa / pa.w); atomicAdd(dA + 1, gya / pa.w); atomicAdd(dA + 3, -(gxa * xa + gya * ya) / pa.w);
    atomicAdd(dB + 0, gxb / pv.w); atomicAdd(dB + 1, gyb / pv.w); atomicAdd(dB + 3, -(gxb * xb + gyb * yb) / pv.w);
}

// ------------------------------------------------------------------------------------------ renderer glue
// The elementwise chain around the ops in DiffRastRenderer.render (diff_mesh_renderer.py:94-96, 139-151) as two small kernels each way:
// vertex transform (pad + two 4x4 GEMMs in the reference) and the final shade (clamp alpha, composite over the background, clamp).
__global__ void __launch_bounds__(256) k_transform_fwd(const float* __restrict__ v, const float* __restrict__ M, int V, float4* __restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= V) return;
    const float x = v[3 * i], y = v[3 * i + 1], z = v[3 * i + 2];
    out[i] = make_float4(M[0] * x + M[1] * y + M[2] * z + M[3], M[4] * x + M[5] * y + M[6] * z + M[7],
                         M[8] * x + M[9] * y + M[10] * z + M[11], M[12] * x + M[13] * y + M[14] * z + M[15]);
}
__global__ void __launch_bounds__(256) k_transform_bwd(const float* __restrict__ M, const float4* __restrict__ dout, int V, float* __restrict__ dv) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= V) return;
    const float4 g = dout[i];
    dv[3 * i] = M[0] * g.x + M[4] * g.y + M[8] * g.z + M[12] * g.w;
    dv[3 * i + 1] = M[1] * g.x + M[5] * g.y + M[9] * g.z + M[13] * g.w;
    dv[3 * i + 2] = M[2] * g.x + M[6] * g.y + M[10] * g.z + M[14] * g.w;
}
// image = clamp(a * albedo + (1 - a) * bg, 0, 1), a = clamp(alpha, 0, 1)
__global__ void __launch_bounds__(256) k_shade_fwd(const float* __restrict__ albedo, const float* __restrict__ alpha, const float* __restrict__ bg, long long P,
                                                    float* __restrict__ image, float* __restrict__ alpha_out) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= P) return;
    const float a = fminf(fmaxf(alpha[i], 0.f), 1.f);
    alpha_out[i] = a;
#pragma unroll
    for (int c = 0; c < 3; c++) image[3 * i + c] = fminf(fmaxf(a * albedo[3 * i + c] + (1.f - a) * bg[c], 0.f), 1.f);
}
__global__ void __launch_bounds__(256) k_shade_bwd(const float* __restrict__ albedo, const float* __restrict__ alpha, const float* __restrict__ bg, long long P,
                                                    const float* __restrict__ dimage, const float* __restrict__ dalpha_out, float* __restrict__ dalbedo,
                                                    float* __restrict__ dalpha) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= P) return;
    const float ar = alpha[i], a = fminf(fmaxf(ar, 0.f), 1.f);
    float da = dalpha_out ? dalpha_out[i] : 0.f;
#pragma unroll
    for (int c = 0; c < 3; c++) {
        const float al = albedo[3 * i + c], val = a * al + (1.f - a) * bg[c];
        const float g = (val >= 0.f && val <= 1.f) ? (dimage ? dimage[3 * i + c] : 0.f) : 0.f;     // torch.clamp passes the gradient on the closed interval
        dalbedo[3 * i + c] = g * a;
        da += g * (al - bg[c]);
    }
    dalpha[i] = (ar >= 0.f && ar <= 1.f) ? da : 0.f;
}

// ------------------------------------------------------------------------------------------ C-ABI
#define MESH_REQUIRE(cond, msg) do { if (!(cond)) { c3d_set_error("c3d_mesh: " msg); return -1; } } while (0)
static inline size_t zbuf_bytes(int B, int H, int W) { return c3d_align(8 * (size_t)B * H * W); }

extern "C" {

size_t c3d_mesh_raster_scratch_bytes(int32_t B, int32_t H, int32_t W, int32_t T) {
    return zbuf_bytes(B, H, W) + c3d_align(4 * ((size_t)B * (size_t)(T > 0 ? T : 1))) + 256;
}

int c3d_mesh_rasterize_fwd(const float* pos, const int32_t* tri, int32_t B, int32_t V, int32_t T, int32_t H, int32_t W, void* scratch, float* rast,
                           float* rast_db, c3d_stream_t stream) {
    return c3d_mesh_rasterize_peel_fwd(pos, tri, B, V, T, H, W, nullptr, scratch, rast, rast_db, stream);
}
// `resolve` = false: stop after the depth | id buffer is complete (the fused view resolves inside its own pixel pass)
static int mesh_rasterize_impl(const float* pos, const int32_t* tri, int32_t B, int32_t V, int32_t T, int32_t H, int32_t W, const void* prev_scratch,
                               void* scratch, float* rast, float* rast_db, c3d_stream_t stream, bool resolve);
int c3d_mesh_rasterize_peel_fwd(const float* pos, const int32_t* tri, int32_t B, int32_t V, int32_t T, int32_t H, int32_t W, const void* prev_scratch,
                                void* scratch, float* rast, float* rast_db, c3d_stream_t stream) {
    return mesh_rasterize_impl(pos, tri, B, V, T, H, W, prev_scratch, scratch, rast, rast_db, stream, true);
}
static int mesh_rasterize_impl(const float* pos, const int32_t* tri, int32_t B, int32_t V, int32_t T, int32_t H, int32_t W, const void* prev_scratch,
                               void* scratch, float* rast, float* rast_db, c3d_stream_t stream, bool resolve) {
    hipStream_t s = (hipStream_t)stream;
    const unsigned long long* peel = (const unsigned long long*)prev_scratch;   // the previous layer's depth|id words lead its scratch
    MESH_REQUIRE(prev_scratch != scratch || !scratch, "depth peeling needs two scratch buffers");
    MESH_REQUIRE(B >= 0 && V >= 0 && T >= 0 && H >= 0 && W >= 0, "negative size");
    MESH_REQUIRE(H <= 32768 && W <= 32768, "resolution above 32768 is not supported");
    const long long BP = (long long)B * H * W;
    if (BP == 0) return 0;
    MESH_REQUIRE(rast && scratch, "NULL buffer");
    MESH_REQUIRE((long long)B * T < (1ll << 32), "B*T must fit 32 bits");
    C3dProfScope ps(C3D_P_MESH_RASTERIZE, s);
    unsigned long long* zbuf = (unsigned long long*)scratch;
    uint32_t* queue = (uint32_t*)((char*)scratch + zbuf_bytes(B, H, W));
    uint32_t* count = (uint32_t*)((char*)queue + c3d_align(4 * ((size_t)B * (size_t)(T > 0 ? T : 1))));
    C3D_CHECK(hipMemsetAsync(zbuf, 0xFF, 8 * (size_t)BP, s));
    C3D_CHECK(hipMemsetAsync(count, 0, 4, s));
    if (T > 0 && V > 0) {
        MESH_REQUIRE(pos && tri, "NULL geometry");
        { C3dProfScope pt(C3D_P_MESH_RAS_TRI, s);
          hipLaunchKernelGGL(k_ras_tri, dim3(c3d_cdiv((long long)B * T, 256)), dim3(256), 0, s, (const float4*)pos, (const int3*)tri, B, V, T, H, W, zbuf, queue, count, peel); }
        hipLaunchKernelGGL(k_ras_big, dim3(2048), dim3(256), 0, s, (const float4*)pos, (const int3*)tri, V, T, H, W, zbuf, queue, count, peel);
    }
    if (resolve) hipLaunchKernelGGL(k_ras_resolve, dim3(c3d_cdiv(BP, 256)), dim3(256), 0, s, (const float4*)pos, (const int3*)tri, B, V, H, W, zbuf, (float4*)rast, (float4*)rast_db);
    C3D_LAUNCH_CHECK();
    return 0;
}

int c3d_mesh_rasterize_bwd(const float* pos, const int32_t* tri, const float* rast, const float* dy, const float* ddb, int32_t B, int32_t V, int32_t T, int32_t H,
                           int32_t W, float* dpos, c3d_stream_t stream) {
    (void)T;
    hipStream_t s = (hipStream_t)stream;
    if ((long long)B * V == 0) return 0;
    MESH_REQUIRE(dpos, "NULL dpos");
    C3dProfScope ps(C3D_P_MESH_RASTERIZE_BWD, s);
    C3D_CHECK(hipMemsetAsync(dpos, 0, sizeof(float) * 4 * (size_t)B * V, s));
    const long long BP = (long long)B * H * W;
    if (BP == 0) return 0;
    MESH_REQUIRE(pos && tri && rast && (dy || ddb), "NULL pointer");
    hipLaunchKernelGGL(k_ras_bwd, dim3(c3d_cdiv(BP, 256)), dim3(256), 0, s, (const float4*)pos, (const int3*)tri, (const float4*)rast, (const float4*)dy, (const float4*)ddb, B, V, H, W, dpos);
    C3D_LAUNCH_CHECK();
    return 0;
}

size_t c3d_mesh_vertex_topology_bytes(int32_t V, int32_t T) { VertexTopo t; carve_vertex_topo(nullptr, V, T, t); return t.bytes; }
int c3d_mesh_build_vertex_topology(const int32_t* tri, int32_t V, int32_t T, void* topology, c3d_stream_t stream) {
    hipStream_t s = (hipStream_t)stream;
    MESH_REQUIRE(topology, "NULL topology buffer");
    VertexTopo t; carve_vertex_topo((char*)topology, V, T, t);
    const int n3 = 3 * T;
    if (V <= 0 || T <= 0) { C3D_CHECK(hipMemsetAsync(t.start, 0, 4 * ((size_t)(V > 0 ? V : 0) + 2), s)); C3D_CHECK(hipMemsetAsync(t.meta, 0, 64, s)); return 0; }
    MESH_REQUIRE(tri, "NULL tri");
    hipLaunchKernelGGL(k_topo_keys, dim3(c3d_cdiv(n3, 256)), dim3(256), 0, s, (const int*)tri, n3, V, t.key[0]);
    int bits = 1;
    while ((1ll << bits) <= (long long)V) bits++;
    int res = 0, rc;
    if ((rc = c3d_sort_pairs_u32(t.key[0], t.key[1], t.val[0], t.val[1], true, (size_t)n3, bits, t.tmp, &res, s))) return rc;
    hipLaunchKernelGGL(k_topo_starts, dim3(c3d_cdiv(n3 + 1, 256)), dim3(256), 0, s, t.key[res], n3, V, t.start, t.meta, res);
    C3D_LAUNCH_CHECK();
    return 0;
}
size_t c3d_mesh_rasterize_bwd_scratch_bytes(int32_t B, int32_t T) {
    const size_t bt = (size_t)(B > 0 ? B : 1) * (size_t)(T > 0 ? T : 1);
    return c3d_align(sizeof(float4) * 3 * bt) + c3d_align(4 * bt) + c3d_align(64);
}
static int mesh_rasterize_bwd_gather(const float* pos, const int32_t* tri, const float* rast, const float* dy, const float* ddb, int32_t B, int32_t V, int32_t T, int32_t H,
                                     int32_t W, const void* topology, void* scratch, float* dpos, c3d_stream_t stream, AaBwdIn aa, TriOwned own);
int c3d_mesh_rasterize_bwd_gather(const float* pos, const int32_t* tri, const float* rast, const float* dy, const float* ddb, int32_t B, int32_t V, int32_t T, int32_t H,
                                  int32_t W, const void* topology, void* scratch, float* dpos, c3d_stream_t stream) {
    return mesh_rasterize_bwd_gather(pos, tri, rast, dy, ddb, B, V, T, H, W, topology, scratch, dpos, stream, AaBwdIn{nullptr, nullptr, nullptr, nullptr, nullptr, nullptr}, TriOwned{nullptr, 0});
}
// aa.hit != NULL (fused views): the antialias pass's position gradient is accumulated into the same per-corner records; own: see TriOwned
static int mesh_rasterize_bwd_gather(const float* pos, const int32_t* tri, const float* rast, const float* dy, const float* ddb, int32_t B, int32_t V, int32_t T, int32_t H,
                                     int32_t W, const void* topology, void* scratch, float* dpos, c3d_stream_t stream, AaBwdIn aa, TriOwned own) {
    hipStream_t s = (hipStream_t)stream;
    if ((long long)B * V == 0) return 0;
    MESH_REQUIRE(dpos, "NULL dpos");
    C3dProfScope ps(C3D_P_MESH_RASTERIZE_BWD, s);
    const long long BP = (long long)B * H * W;
    if (BP == 0 || T == 0) { C3D_CHECK(hipMemsetAsync(dpos, 0, sizeof(float) * 4 * (size_t)B * V, s)); return 0; }
    MESH_REQUIRE(pos && tri && rast && (dy || ddb) && topology && scratch, "NULL pointer");
    VertexTopo t; carve_vertex_topo((char*)topology, V, T, t);
    const size_t bt = (size_t)B * T;
    float4* rec = (float4*)scratch;
    uint32_t* queue = (uint32_t*)((char*)scratch + c3d_align(sizeof(float4) * 3 * bt));
    uint32_t* count = (uint32_t*)((char*)queue + c3d_align(4 * bt));
    C3D_CHECK(hipMemsetAsync(count, 0, 8, s));
#define RAS_BWD_LAUNCH(DB_)                                                                                                                                          \
    hipLaunchKernelGGL((k_ras_bwd_tri<DB_>), dim3(c3d_cdiv((long long)bt, 256)), dim3(256), 0, s, (const float4*)pos, (const int3*)tri, (const float4*)rast, (const float4*)dy, \
                       (const float4*)ddb, B, V, T, H, W, rec, queue, count, aa, own);                                                                                       \
    hipLaunchKernelGGL((k_ras_bwd_big<DB_>), dim3(1024), dim3(256), 0, s, (const float4*)pos, (const int3*)tri, (const float4*)rast, (const float4*)dy, (const float4*)ddb, V, T, H, W, \
                       rec, queue, count, aa)
    if (ddb) { RAS_BWD_LAUNCH(true); } else { RAS_BWD_LAUNCH(false); }
#undef RAS_BWD_LAUNCH
    // the sorted corner list sits in val[res]; res is a host constant of the build (same digit count): recompute it instead of reading meta
    int bits = 1;
    while ((1ll << bits) <= (long long)V) bits++;
    const int res = ((bits + 7) / 8) & 1;
    // the large-triangle queue has been consumed: its storage and the second counter word serve the heavy-vertex queue
    hipLaunchKernelGGL(k_vertex_gather4, dim3(c3d_cdiv((long long)B * V, 256)), dim3(256), 0, s, (const float4*)rec, t.start, t.val[res], B, V, T, (float4*)dpos, queue, count + 1, own);
    hipLaunchKernelGGL(k_vertex_gather4_heavy, dim3(64), dim3(256), 0, s, (const float4*)rec, t.start, t.val[res], V, T, (float4*)dpos, queue, count + 1, own);
    C3D_LAUNCH_CHECK();
    return 0;
}

int c3d_mesh_interpolate_fwd(const float* attr, int32_t Ba, const float* rast, const int32_t* tri, const float* rast_db, const int32_t* diff, int32_t nd,
                             int32_t B, int32_t V, int32_t A, int32_t H, int32_t W, float* out, float* out_da, c3d_stream_t stream) {
    hipStream_t s = (hipStream_t)stream;
    const long long P = (long long)H * W, BP = P * B;
    if (BP == 0 || A == 0) return 0;
    MESH_REQUIRE(attr && rast && tri && out, "NULL pointer");
    MESH_REQUIRE(Ba == 1 || Ba == B, "attribute batch must be 1 or B");
    MESH_REQUIRE(nd == 0 || (rast_db && diff && out_da), "diff_attrs needs rast_db");
    C3dProfScope ps(C3D_P_MESH_INTERPOLATE, s);
    hipLaunchKernelGGL(k_interp_fwd, dim3(c3d_cdiv(BP, 256)), dim3(256), 0, s, attr, Ba, (const float4*)rast, (const int3*)tri, (const float4*)rast_db, diff, nd, BP, P, V, A, out, out_da);
    C3D_LAUNCH_CHECK();
    return 0;
}
int c3d_mesh_interpolate_bwd(const float* attr, int32_t Ba, const float* rast, const int32_t* tri, const float* dy, int32_t B, int32_t V, int32_t A,
                             int32_t H, int32_t W, float* dattr, float* drast, c3d_stream_t stream) {
    hipStream_t s = (hipStream_t)stream;
    const long long P = (long long)H * W, BP = P * B;
    MESH_REQUIRE(Ba == 1 || Ba == B, "attribute batch must be 1 or B");
    C3dProfScope ps(C3D_P_MESH_INTERPOLATE_BWD, s);
    if (dattr && (long long)Ba * V * A > 0) C3D_CHECK(hipMemsetAsync(dattr, 0, sizeof(float) * (size_t)Ba * V * A, s));   // NULL: attribute gradient not wanted
    if (BP == 0) return 0;
    MESH_REQUIRE(attr && rast && tri && dy && drast, "NULL pointer");
    hipLaunchKernelGGL(k_interp_bwd, dim3(c3d_cdiv(BP, 256)), dim3(256), 0, s, attr, Ba, (const float4*)rast, (const int3*)tri, dy, BP, P, V, A, dattr, (float4*)drast);
    C3D_LAUNCH_CHECK();
    return 0;
}

int c3d_mesh_interpolate_da_bwd(const float* attr, int32_t Ba, const float* rast, const int32_t* tri, const float* rast_db, const int32_t* diff_attrs, int32_t n_diff,
                                const float* dout_da, int32_t B, int32_t V, int32_t A, int32_t H, int32_t W, float* dattr, float* drast_db, c3d_stream_t stream) {
    hipStream_t s = (hipStream_t)stream;
    const long long P = (long long)H * W, BP = P * B;
    MESH_REQUIRE(Ba == 1 || Ba == B, "attribute batch must be 1 or B");
    if (BP == 0 || n_diff == 0) return 0;
    MESH_REQUIRE(attr && rast && tri && rast_db && diff_attrs && dout_da && drast_db, "NULL pointer");
    C3dProfScope ps(C3D_P_MESH_INTERPOLATE_BWD, s);
    hipLaunchKernelGGL(k_interp_da_bwd, dim3(c3d_cdiv(BP, 256)), dim3(256), 0, s, attr, Ba, (const float4*)rast, (const int3*)tri, (const float4*)rast_db, diff_attrs, n_diff,
                       dout_da, BP, P, V, A, dattr, (float4*)drast_db);
    C3D_LAUNCH_CHECK();
    return 0;
}

int c3d_mesh_texture_fwd(const float* tex, int32_t Bt, const float* uv, int32_t B, int32_t H, int32_t W, int32_t Ht, int32_t Wt, int32_t C, int32_t filter,
                         int32_t boundary, float* out, c3d_stream_t stream) {
    hipStream_t s = (hipStream_t)stream;
    const long long P = (long long)H * W, BP = P * B;
    if (BP == 0 || C == 0) return 0;
    MESH_REQUIRE(tex && uv && out && Ht > 0 && Wt > 0, "NULL pointer / empty texture");
    MESH_REQUIRE(Bt == 1 || Bt == B, "texture batch must be 1 or B");
    MESH_REQUIRE((filter == 0 || filter == 1) && (boundary == 0 || boundary == 1), "unsupported filter/boundary mode");
    C3dProfScope ps(C3D_P_MESH_TEXTURE, s);
    hipLaunchKernelGGL(k_tex_fwd, dim3(c3d_cdiv(BP, 256)), dim3(256), 0, s, tex, Bt, (const float2*)uv, BP, P, Ht, Wt, C, filter, boundary, out);
    C3D_LAUNCH_CHECK();
    return 0;
}
static int mesh_texture_bwd(const float* tex, int32_t Bt, const float* uv, const float* dy, int32_t B, int32_t H, int32_t W, int32_t Ht, int32_t Wt, int32_t C,
                            int32_t filter, int32_t boundary, float* dtex, float* duv, c3d_stream_t stream, bool zero_dtex);
int c3d_mesh_texture_bwd(const float* tex, int32_t Bt, const float* uv, const float* dy, int32_t B, int32_t H, int32_t W, int32_t Ht, int32_t Wt, int32_t C,
                         int32_t filter, int32_t boundary, float* dtex, float* duv, c3d_stream_t stream) {
    return mesh_texture_bwd(tex, Bt, uv, dy, B, H, W, Ht, Wt, C, filter, boundary, dtex, duv, stream, true);
}
// zero_dtex = false: the texel gradients are ADDED to what dtex holds (the multi-view step accumulates the views of a lane in one buffer)
static int mesh_texture_bwd(const float* tex, int32_t Bt, const float* uv, const float* dy, int32_t B, int32_t H, int32_t W, int32_t Ht, int32_t Wt, int32_t C,
                            int32_t filter, int32_t boundary, float* dtex, float* duv, c3d_stream_t stream, bool zero_dtex) {
    hipStream_t s = (hipStream_t)stream;
    const long long P = (long long)H * W, BP = P * B;
    MESH_REQUIRE(Bt == 1 || Bt == B, "texture batch must be 1 or B");
    MESH_REQUIRE((filter == 0 || filter == 1) && (boundary == 0 || boundary == 1), "unsupported filter/boundary mode");
    C3dProfScope ps(C3D_P_MESH_TEXTURE_BWD, s);
    if ((long long)Bt * Ht * Wt * C > 0) { MESH_REQUIRE(dtex, "NULL dtex"); if (zero_dtex) C3D_CHECK(hipMemsetAsync(dtex, 0, sizeof(float) * (size_t)Bt * Ht * Wt * C, s)); }
    if (BP == 0 || C == 0) return 0;
    MESH_REQUIRE(tex && uv && dy && duv, "NULL pointer");
    if (filter == 1 && (C == 1 || C == 3 || C == 4) && (long long)Ht * Wt < 0xFFFFFFFFll) {
        const dim3 grid(c3d_cdiv(W, 16), c3d_cdiv(H, 16), B);
        if (C == 1) hipLaunchKernelGGL((k_tex_bwd_tiled<1>), grid, dim3(256), 0, s, tex, Bt, (const float2*)uv, dy, H, W, Ht, Wt, boundary, dtex, (float2*)duv);
        else if (C == 3) hipLaunchKernelGGL((k_tex_bwd_tiled<3>), grid, dim3(256), 0, s, tex, Bt, (const float2*)uv, dy, H, W, Ht, Wt, boundary, dtex, (float2*)duv);
        else hipLaunchKernelGGL((k_tex_bwd_tiled<4>), grid, dim3(256), 0, s, tex, Bt, (const float2*)uv, dy, H, W, Ht, Wt, boundary, dtex, (float2*)duv);
    } else {
        hipLaunchKernelGGL(k_tex_bwd, dim3(c3d_cdiv(BP, 256)), dim3(256), 0, s, tex, Bt, (const float2*)uv, dy, BP, P, Ht, Wt, C, filter, boundary, dtex, (float2*)duv);
    }
    C3D_LAUNCH_CHECK();
    return 0;
}

int32_t c3d_mesh_mip_info(int32_t Ht, int32_t Wt, int32_t max_mip_level, int32_t* levels_hw, int64_t* stack_texels) {
    MipShape m;
    if (Ht <= 0 || Wt <= 0) { c3d_set_error("c3d_mesh_mip_info: empty texture"); return -1; }
    if (mip_shape(Ht, Wt, max_mip_level, m)) {
        c3d_set_error("mip pyramid: an odd extent > 1 would have to be halved (limit the depth with max_mip_level)");
        return -1;
    }
    if (levels_hw) for (int l = 0; l <= m.L; l++) { levels_hw[2 * l] = m.h[l]; levels_hw[2 * l + 1] = m.w[l]; }
    if (stack_texels) *stack_texels = m.total;
    return m.L;
}
#define MIP_SHAPE(m)                                                                                              \
    MipShape m;                                                                                                   \
    MESH_REQUIRE(Ht > 0 && Wt > 0, "empty texture");                                                              \
    MESH_REQUIRE(mip_shape(Ht, Wt, max_mip_level, m) == 0, "mip pyramid: an odd extent > 1 would have to be halved"); \
    MESH_REQUIRE((long long)Ht * Wt + m.total < 0xFFFFFFFFll, "texture too large for 32-bit texel indices")
int c3d_mesh_mip_build(const float* tex, int32_t Bt, int32_t Ht, int32_t Wt, int32_t C, int32_t max_mip_level, float* stack, c3d_stream_t stream) {
    hipStream_t s = (hipStream_t)stream;
    MIP_SHAPE(m);
    if (m.L == 0 || (long long)Bt * C == 0) return 0;
    MESH_REQUIRE(tex && stack, "NULL pointer");
    C3dProfScope ps(C3D_P_MESH_TEXTURE, s);
    for (int l = 1; l <= m.L; l++) {
        const float* src = l == 1 ? tex : stack + (size_t)m.off[l - 1] * C;
        const size_t sstride = l == 1 ? (size_t)Ht * Wt * C : (size_t)m.total * C;
        const long long n = (long long)Bt * m.h[l] * m.w[l] * C;
        hipLaunchKernelGGL(k_mip_down, dim3(c3d_cdiv(n, 256)), dim3(256), 0, s, src, stack + (size_t)m.off[l] * C, Bt, m.h[l - 1], m.w[l - 1], m.h[l], m.w[l], C,
                           sstride, (size_t)m.total * C);
    }
    C3D_LAUNCH_CHECK();
    return 0;
}
int c3d_mesh_mip_build_bwd(float* dstack, int32_t Bt, int32_t Ht, int32_t Wt, int32_t C, int32_t max_mip_level, float* dtex, c3d_stream_t stream) {
    hipStream_t s = (hipStream_t)stream;
    MIP_SHAPE(m);
    if ((long long)Bt * C == 0) return 0;
    MESH_REQUIRE(dtex, "NULL dtex");
    C3dProfScope ps(C3D_P_MESH_TEXTURE_BWD, s);
    C3D_CHECK(hipMemsetAsync(dtex, 0, sizeof(float) * (size_t)Bt * Ht * Wt * C, s));
    if (m.L == 0) return 0;
    MESH_REQUIRE(dstack, "NULL dstack");
    for (int l = m.L; l >= 1; l--) {
        float* fine = l == 1 ? dtex : dstack + (size_t)m.off[l - 1] * C;
        const size_t fstride = l == 1 ? (size_t)Ht * Wt * C : (size_t)m.total * C;
        const long long n = (long long)Bt * m.h[l - 1] * m.w[l - 1] * C;
        hipLaunchKernelGGL(k_mip_fold, dim3(c3d_cdiv(n, 256)), dim3(256), 0, s, dstack + (size_t)m.off[l] * C, fine, Bt, m.h[l - 1], m.w[l - 1], m.h[l], m.w[l], C,
                           (size_t)m.total * C, fstride);
    }
    C3D_LAUNCH_CHECK();
    return 0;
}
int c3d_mesh_texture_mip_fwd(const float* tex, const float* stack, int32_t Bt, const float* uv, const float* uv_da, const float* mip_level_bias, int32_t B,
                             int32_t H, int32_t W, int32_t Ht, int32_t Wt, int32_t C, int32_t filter, int32_t boundary, int32_t max_mip_level, float* out,
                             c3d_stream_t stream) {
    hipStream_t s = (hipStream_t)stream;
    const long long P = (long long)H * W, BP = P * B;
    if (BP == 0 || C == 0) return 0;
    MIP_SHAPE(m);
    MESH_REQUIRE(tex && uv && out && (stack || m.L == 0), "NULL pointer");
    MESH_REQUIRE(uv_da || mip_level_bias, "mip-mapped filter modes need uv_da or mip_level_bias");
    MESH_REQUIRE(Bt == 1 || Bt == B, "texture batch must be 1 or B");
    MESH_REQUIRE((filter == 2 || filter == 3) && (boundary >= 0 && boundary <= 2), "unsupported filter/boundary mode");
    C3dProfScope ps(C3D_P_MESH_TEXTURE, s);
    hipLaunchKernelGGL(k_tex_mip_fwd, dim3(c3d_cdiv(BP, 256)), dim3(256), 0, s, tex, stack ? stack : tex, Bt, (const float2*)uv, (const float4*)uv_da, mip_level_bias,
                       BP, P, Ht, Wt, C, m.L, m.total, filter, boundary, out);
    C3D_LAUNCH_CHECK();
    return 0;
}
int c3d_mesh_texture_mip_bwd(const float* tex, const float* stack, int32_t Bt, const float* uv, const float* uv_da, const float* mip_level_bias, const float* dy,
                             int32_t B, int32_t H, int32_t W, int32_t Ht, int32_t Wt, int32_t C, int32_t filter, int32_t boundary, int32_t max_mip_level,
                             float* dtex, float* dstack, float* duv, float* d_uv_da, float* d_bias, c3d_stream_t stream) {
    hipStream_t s = (hipStream_t)stream;
    const long long P = (long long)H * W, BP = P * B;
    MIP_SHAPE(m);
    MESH_REQUIRE(Bt == 1 || Bt == B, "texture batch must be 1 or B");
    MESH_REQUIRE((filter == 2 || filter == 3) && (boundary >= 0 && boundary <= 2), "unsupported filter/boundary mode");
    MESH_REQUIRE((!d_uv_da || uv_da) && (!d_bias || mip_level_bias), "a gradient was asked for an input that is absent");
    C3dProfScope ps(C3D_P_MESH_TEXTURE_BWD, s);
    if ((long long)Bt * C > 0) {
        MESH_REQUIRE(dtex && (dstack || m.L == 0), "NULL dtex / dstack");
        C3D_CHECK(hipMemsetAsync(dtex, 0, sizeof(float) * (size_t)Bt * Ht * Wt * C, s));
        if (m.total) C3D_CHECK(hipMemsetAsync(dstack, 0, sizeof(float) * (size_t)Bt * m.total * C, s));
    }
    if (BP == 0 || C == 0) return 0;
    MESH_REQUIRE(tex && uv && dy && duv && (stack || m.L == 0), "NULL pointer");
    MESH_REQUIRE(uv_da || mip_level_bias, "mip-mapped filter modes need uv_da or mip_level_bias");
    const dim3 grid(c3d_cdiv(W, 16), c3d_cdiv(H, 16), B);
    const float* st = stack ? stack : tex;
    float* dst = dstack ? dstack : dtex;
#define MIP_BWD(CT) hipLaunchKernelGGL((k_tex_mip_bwd<CT>), grid, dim3(256), 0, s, tex, st, Bt, (const float2*)uv, (const float4*)uv_da, mip_level_bias, dy, H, W, \
                                       Ht, Wt, C, m.L, m.total, filter, boundary, dtex, dst, (float2*)duv, (float4*)d_uv_da, d_bias)
    if (C == 1) MIP_BWD(1); else if (C == 3) MIP_BWD(3); else if (C == 4) MIP_BWD(4); else MIP_BWD(0);
#undef MIP_BWD
    C3D_LAUNCH_CHECK();
    return 0;
}

int c3d_mesh_transform_fwd(const float* v, const float* M, int32_t V, float* out, c3d_stream_t stream) {
    if (V <= 0) return 0;
    MESH_REQUIRE(v && M && out, "NULL pointer");
    hipLaunchKernelGGL(k_transform_fwd, dim3(c3d_cdiv(V, 256)), dim3(256), 0, (hipStream_t)stream, v, M, V, (float4*)out);
    C3D_LAUNCH_CHECK();
    return 0;
}
int c3d_mesh_transform_bwd(const float* M, const float* dout, int32_t V, float* dv, c3d_stream_t stream) {
    if (V <= 0) return 0;
    MESH_REQUIRE(M && dout && dv, "NULL pointer");
    hipLaunchKernelGGL(k_transform_bwd, dim3(c3d_cdiv(V, 256)), dim3(256), 0, (hipStream_t)stream, M, (const float4*)dout, V, dv);
    C3D_LAUNCH_CHECK();
    return 0;
}
int c3d_mesh_shade_fwd(const float* albedo, const float* alpha, const float* bg, int64_t P, float* image, float* alpha_out, c3d_stream_t stream) {
    if (P <= 0) return 0;
    MESH_REQUIRE(albedo && alpha && bg && image && alpha_out, "NULL pointer");
    hipLaunchKernelGGL(k_shade_fwd, dim3(c3d_cdiv(P, 256)), dim3(256), 0, (hipStream_t)stream, albedo, alpha, bg, (long long)P, image, alpha_out);
    C3D_LAUNCH_CHECK();
    return 0;
}
int c3d_mesh_shade_bwd(const float* albedo, const float* alpha, const float* bg, int64_t P, const float* dimage, const float* dalpha_out, float* dalbedo,
                       float* dalpha, c3d_stream_t stream) {
    if (P <= 0) return 0;
    MESH_REQUIRE(albedo && alpha && bg && dalbedo && dalpha, "NULL pointer");
    hipLaunchKernelGGL(k_shade_bwd, dim3(c3d_cdiv(P, 256)), dim3(256), 0, (hipStream_t)stream, albedo, alpha, bg, (long long)P, dimage, dalpha_out, dalbedo, dalpha);
    C3D_LAUNCH_CHECK();
    return 0;
}

size_t c3d_mesh_antialias_scratch_bytes(int32_t T) { return c3d_align(sizeof(EdgeSlot) * (size_t)edge_table_size(T)) + c3d_align(sizeof(int) * 3 * (size_t)(T > 0 ? T : 1)); }

int c3d_mesh_antialias_build_topology(const int32_t* tri, int32_t T, void* scratch, c3d_stream_t stream) {
    hipStream_t s = (hipStream_t)stream;
    MESH_REQUIRE(scratch, "NULL scratch");
    C3dProfScope ps(C3D_P_MESH_ANTIALIAS, s);
    const uint32_t n = edge_table_size(T);
    EdgeSlot* table = (EdgeSlot*)scratch;
    hipLaunchKernelGGL(k_aa_hash_init, dim3(c3d_cdiv((long long)n, 256)), dim3(256), 0, s, table, n);
    if (T > 0) {
        MESH_REQUIRE(tri, "NULL tri");
        hipLaunchKernelGGL(k_aa_hash_build, dim3(c3d_cdiv(3ll * T, 256)), dim3(256), 0, s, (const int3*)tri, T, table, n - 1);
        hipLaunchKernelGGL(k_aa_adj_build, dim3(c3d_cdiv(3ll * T, 256)), dim3(256), 0, s, (const int3*)tri, T, (const EdgeSlot*)table, n - 1, (int*)(table + n));
    }
    C3D_LAUNCH_CHECK();
    return 0;
}

int c3d_mesh_antialias_fwd(const float* color, const float* rast, const float* pos, const int32_t* tri, int32_t B, int32_t V, int32_t T, int32_t H, int32_t W,
                           int32_t C, const void* scratch, float* out, c3d_stream_t stream) {
    hipStream_t s = (hipStream_t)stream;
    const long long P = (long long)H * W, BP = P * B;
    if (BP == 0 || C == 0) return 0;
    MESH_REQUIRE(color && rast && out && scratch, "NULL pointer");
    C3dProfScope ps(C3D_P_MESH_ANTIALIAS, s);
    C3D_CHECK(hipMemcpyAsync(out, color, sizeof(float) * (size_t)BP * C, hipMemcpyDeviceToDevice, s));
    if (T > 0) {
        MESH_REQUIRE(pos && tri, "NULL geometry");
        const uint32_t n = edge_table_size(T);
        hipLaunchKernelGGL(k_aa_fwd, dim3(c3d_cdiv(BP * 2, 256)), dim3(256), 0, s, color, (const float4*)rast, (const float4*)pos, (const int3*)tri,
                           (const EdgeSlot*)scratch, n - 1, B, V, H, W, C, out);
    }
    C3D_LAUNCH_CHECK();
    return 0;
}
int c3d_mesh_antialias_bwd(const float* color, const float* rast, const float* pos, const int32_t* tri, const float* dy, int32_t B, int32_t V, int32_t T,
                           int32_t H, int32_t W, int32_t C, const void* scratch, float* dcolor, float* dpos, c3d_stream_t stream) {
    hipStream_t s = (hipStream_t)stream;
    const long long P = (long long)H * W, BP = P * B;
    C3dProfScope ps(C3D_P_MESH_ANTIALIAS_BWD, s);
    if ((long long)B * V > 0) { MESH_REQUIRE(dpos, "NULL dpos"); C3D_CHECK(hipMemsetAsync(dpos, 0, sizeof(float) * 4 * (size_t)B * V, s)); }
    if (BP == 0 || C == 0) return 0;
    MESH_REQUIRE(color && rast && dy && dcolor && scratch, "NULL pointer");
    C3D_CHECK(hipMemcpyAsync(dcolor, dy, sizeof(float) * (size_t)BP * C, hipMemcpyDeviceToDevice, s));
    if (T > 0) {
        MESH_REQUIRE(pos && tri, "NULL geometry");
        const uint32_t n = edge_table_size(T);
        hipLaunchKernelGGL(k_aa_bwd, dim3(c3d_cdiv(BP * 2, 256)), dim3(256), 0, s, color, (const float4*)rast, (const float4*)pos, (const int3*)tri,
                           (const EdgeSlot*)scratch, n - 1, dy, B, V, H, W, C, dcolor, dpos);
    }
    C3D_LAUNCH_CHECK();
    return 0;
}

}  // extern "C"

// ================================================================================================ fused view (round 2)
// DiffRastRenderer.render (reference diff_mesh_renderer.py:72-159) for one view as ONE library call each way.  Round 1 issued the seven
// nvdiffrast ops + ~15 torch elementwise ops of a view as separate autograd nodes: 1.07 ms per view of which 0.5 ms were kernels
// (profiles/r01k_bench_mesh_n1.json).  Here the op sequence -- transform, rasterize, interpolate(uv), texture(linear), sigmoid, antialias,
// composite over the background, clamps -- is enqueued from C without returning to Python, the camera matrix and background travel as kernel
// arguments (no upload), and the two antialias calls of the reference (coverage, :105; albedo, :138) share ONE silhouette analysis per pixel
// pair, forward and backward: the coverage "colour" is just (triangle id > 0) and is never materialised.
// Round 4: every kernel of the fused view takes a VIEW dimension (blockIdx.y, or the high part of a flat pixel index): the views of a training step live in
// batched arrays [B, ...] and go through each stage in ONE launch (the rasterizer's kernels were batch-capable already: B = views).  The step was host-bound --
// ~15 launches per view, 0.5 ms of enqueue time per 1.8 ms 8-view step; per-view camera matrices, backgrounds, targets travel as tables in the kernel arguments.
#define MESH_MAX_VIEWS 16      // views per launch (kernel-argument tables)
struct ViewMat { float m[16]; };
struct ViewMats { ViewMat v[MESH_MAX_VIEWS]; };
__global__ void __launch_bounds__(256) k_view_transform_fwd(const float* __restrict__ v, const float* __restrict__ voff, ViewMats Ms, int V, float4* __restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= V) return;
    const ViewMat& M = Ms.v[blockIdx.y];
    out += (size_t)blockIdx.y * V;
    float x = v[3 * i], y = v[3 * i + 1], z = v[3 * i + 2];
    if (voff) { x += voff[3 * i]; y += voff[3 * i + 1]; z += voff[3 * i + 2]; }
    out[i] = make_float4(M.m[0] * x + M.m[1] * y + M.m[2] * z + M.m[3], M.m[4] * x + M.m[5] * y + M.m[6] * z + M.m[7],
                         M.m[8] * x + M.m[9] * y + M.m[10] * z + M.m[11], M.m[12] * x + M.m[13] * y + M.m[14] * z + M.m[15]);
}
// dv += M^T (d_a + d_b): the two position gradients (antialias: scattered; rasterize: gathered) are summed on the way
// view b: gradients at da / db + b * V, result at dv + b * dv_stride floats
__global__ void __launch_bounds__(256) k_view_transform_bwd(ViewMats Ms, const float4* __restrict__ da, const float4* __restrict__ db, int V, float* __restrict__ dv, size_t dv_stride) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= V) return;
    const ViewMat& M = Ms.v[blockIdx.y];
    if (da) da += (size_t)blockIdx.y * V;
    if (db) db += (size_t)blockIdx.y * V;
    dv += (size_t)blockIdx.y * dv_stride;
    float4 g = da ? da[i] : make_float4(0.f, 0.f, 0.f, 0.f);
    if (db) { const float4 h = db[i]; g.x += h.x; g.y += h.y; g.z += h.z; g.w += h.w; }
    dv[3 * i] = M.m[0] * g.x + M.m[4] * g.y + M.m[8] * g.z + M.m[12] * g.w;
    dv[3 * i + 1] = M.m[1] * g.x + M.m[5] * g.y + M.m[9] * g.z + M.m[13] * g.w;
    dv[3 * i + 2] = M.m[2] * g.x + M.m[6] * g.y + M.m[10] * g.z + M.m[14] * g.w;
}
// Round 3: ONE pixel pass for rasterize's resolve -> interpolate(uv) -> texture('linear', wrap) -> sigmoid -> antialias seeds, which were five passes through HBM
// (k_ras_resolve 12.6 us + k_interp_fwd 11.6 + k_tex_fwd 12.1 + k_view_sigmoid_seed 10.5 per 1024^2 view, profiles/r02z_mesh_kernel_stats.csv).  Statement by
// statement the arithmetic of those kernels (an empty pixel interpolates uv = (0, 0) and fetches there, exactly as the op sequence does: the antialias blend
// across a silhouette reads it).
__global__ void __launch_bounds__(256) k_view_pixel_fwd(const float4* __restrict__ pos, const int3* __restrict__ tri, const float2* __restrict__ vt, const int3* __restrict__ ft,
                                                         const float* __restrict__ tex, int B, int V, int H, int W, int Ht, int Wt, const unsigned long long* __restrict__ zbuf,
                                                         float4* __restrict__ rast, float2* __restrict__ texc, float* __restrict__ albedo0, uint32_t* __restrict__ owned, int owned_words) {
    const long long gid = (long long)blockIdx.x * blockDim.x + threadIdx.x;      // view * H * W + pixel: all arrays are [B, H * W, ..]
    const long long Pv = (long long)H * W;
    if (gid >= (long long)B * Pv) return;
    const int bview = (int)(gid / Pv);
    const long long lp = gid - (long long)bview * Pv;
    pos += (size_t)bview * V;
    const unsigned long long key = zbuf[gid];
    float2 q = make_float2(0.f, 0.f);
    const uint32_t left_id = __shfl_up(key == MESH_EMPTY_KEY ? 0xFFFFFFFFu : (uint32_t)(key & 0xFFFFFFFFull), 1);      // by all lanes, before the branch
    if (key == MESH_EMPTY_KEY) {
        rast[gid] = make_float4(0.f, 0.f, 0.f, 0.f);
    } else {
        const int px = (int)(lp % W), py = (int)(lp / W);
        const uint32_t t = (uint32_t)(key & 0xFFFFFFFFull);
        const int3 vi = tri[t];
        const float xs = 2.f / W, ys = 2.f / H;
        const Frag f = mesh_shade(pos[vi.x], pos[vi.y], pos[vi.z], xs * ((float)px + 0.5f) - 1.f, ys * ((float)py + 0.5f) - 1.f, xs, ys);
        const float u = fminf(fmaxf(f.b0, 0.f), 1.f), v = fminf(fmaxf(f.b1, 0.f), 1.f), w2 = 1.f - u - v;
        // TriOwned: one atomicOr per run of equal ids in the wave's row of pixels (neighbours mostly share the triangle)
        if ((threadIdx.x & 63) == 0 || left_id != t) atomicOr(&owned[(size_t)bview * owned_words + (t >> 5)], 1u << (t & 31));
        rast[gid] = make_float4(u, v, fminf(fmaxf(f.zw, -1.f), 1.f), (float)(t + 1));      // (no rast_db: the linear texture filter has no use for pixel differentials, round 4 stopped writing them)
        const int3 ti = ft[t];
        const float2 a0 = vt[ti.x], a1 = vt[ti.y], a2 = vt[ti.z];
        q = make_float2(u * a0.x + v * a1.x + w2 * a2.x, u * a0.y + v * a1.y + w2 * a2.y);
    }
    texc[gid] = q;
    const float uu = q.x * Wt - 0.5f, vv = q.y * Ht - 0.5f;
    const float fu0 = floorf(uu), fv0 = floorf(vv), fu = uu - fu0, fv = vv - fv0;
    const int iu0 = wrapi((int)fu0, Wt, 0), iu1 = wrapi((int)fu0 + 1, Wt, 0), iv0 = wrapi((int)fv0, Ht, 0), iv1 = wrapi((int)fv0 + 1, Ht, 0);
    const float *t00 = tex + ((size_t)iv0 * Wt + iu0) * 3, *t10 = tex + ((size_t)iv0 * Wt + iu1) * 3, *t01 = tex + ((size_t)iv1 * Wt + iu0) * 3, *t11 = tex + ((size_t)iv1 * Wt + iu1) * 3;
#pragma unroll
    for (int c = 0; c < 3; c++) {
        const float top = t00[c] + fu * (t10[c] - t00[c]), bot = t01[c] + fu * (t11[c] - t01[c]);
        const float sg = 1.f / (1.f + __expf(-(top + fv * (bot - top))));
        albedo0[3 * gid + c] = sg;
    }
}
// ---- round 3: the same pass without atomics (aa_pair_load above) ----
// silhouette analysis only: one flag byte + one blend weight per pixel pair
__global__ void __launch_bounds__(256) k_aa2_pairs(const float4* __restrict__ rast, const float4* __restrict__ pos, const int3* __restrict__ tri,
                                                    const EdgeSlot* __restrict__ table, uint32_t mask, int B, int V, int H, int W, float* __restrict__ pair_alpha, uint8_t* __restrict__ hit,
                                                    const uint8_t* __restrict__ sil, int T) {
    const long long gid = (long long)blockIdx.x * blockDim.x + threadIdx.x;      // view * 2 P + pair
    const long long P = (long long)H * W;
    if (gid >= (long long)B * P * 2) return;
    const int bview = (int)(gid / (2 * P));
    const long long lg = gid - (long long)bview * 2 * P;
    const int d = (int)(lg & 1), pid = (int)(lg >> 1), px = pid % W, py = pid / W;
    AaHit h;
    const bool found = aa_analyze(pos + (size_t)bview * V, tri, table, mask, rast + (size_t)bview * P, H, W, px, py, d, h, sil ? sil + (size_t)bview * T : nullptr);
    hit[gid] = found ? (uint8_t)(1 | (h.sgn > 0.f ? 2 : 0) | (h.ek << 2)) : (uint8_t)0;
    if (found) pair_alpha[gid] = h.s - 0.5f;
}
struct ViewBg { float c[3]; };
struct ViewBgs { ViewBg v[MESH_MAX_VIEWS]; };
// both antialias outputs of a pixel gathered from its four pairs, then the shade (k_view_shade_fwd's statements): albedo_aa / cov_aa are written once, by their owner
__global__ void __launch_bounds__(256) k_view_shade_fwd_g(const float* __restrict__ albedo0, const float4* __restrict__ rast, const uint8_t* __restrict__ hit,
                                                          const float* __restrict__ pair_alpha, ViewBgs bgs, int B, int H, int W, float* __restrict__ albedo_aa,
                                                          float* __restrict__ cov_aa, float* __restrict__ image, float* __restrict__ alpha_out, uint8_t* __restrict__ pflag) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;        // view * P + pixel; a pixel's pairs and neighbours lie inside its view's block of every array
    const long long Pv = (long long)H * W;
    if (i >= (long long)B * Pv) return;
    const int bview = (int)(i / Pv);
    const ViewBg bg = bgs.v[bview];
    const long long lp = i - (long long)bview * Pv;
    const int px = (int)(lp % W), py = (int)(lp / W);
    float acc[3] = {albedo0[3 * i], albedo0[3 * i + 1], albedo0[3 * i + 2]};
    float cov = rast[i].w > 0.f ? 1.f : 0.f;
    uint32_t pf = 0;        // for the backward passes: bit j = pair j is a hit with this pixel as its pixel a, bit 4 + j = pair j is a hit at all
#pragma unroll
    for (int j = 0; j < 4; j++) {
        size_t first; int d;
        if (!aa_pair_of(j, px, py, W, (size_t)i, first, d)) continue;
        const AaPair pr = aa_pair_load(hit, pair_alpha, first, d, W);
        if (!pr.hit) continue;
        pf |= (16u | (pr.ia == (size_t)i ? 1u : 0u)) << j;
        if (pr.idst != (size_t)i) continue;
#pragma unroll
        for (int c = 0; c < 3; c++) acc[c] += pr.alpha * (albedo0[3 * pr.ia + c] - albedo0[3 * pr.ib + c]);
        const float cb = rast[pr.ib].w > 0.f ? 1.f : 0.f;         // pixel a owns the nearer triangle: its coverage is 1
        if (cb != 1.f) cov += pr.alpha * (1.f - cb);
    }
    cov_aa[i] = cov;
    pflag[i] = (uint8_t)pf;
    const float a = fminf(fmaxf(cov, 0.f), 1.f);
    alpha_out[i] = a;
#pragma unroll
    for (int c = 0; c < 3; c++) {
        albedo_aa[3 * i + c] = acc[c];
        image[3 * i + c] = fminf(fmaxf(a * acc[c] + (1.f - a) * bg.c[c], 0.f), 1.f);
    }
}
// ---- texel gradients of the fused view without float atomics ----
// Float adds do not commute in rounding, so any scatter with float atomics gives run-to-run different bits.  Integer adds do: a contribution w is split into
// hi = rint(w 2^16) (an integer count of 2^-16, exact) and lo = w - hi 2^-16 (|lo| <= 2^-17, exact), and lo is added as rint(lo 2^56) (resolution 1.4e-17, 2^24
// such terms fit 63 bits) to a 64-bit word per texel channel, hi -- zero for every |w| < 7.6e-6, i.e. for all of a mean-reduced image loss's texel gradients --
// to a second word.  Integer atomics are exact whatever their order, so ALL views and lanes of a step add into ONE pair of planes and the result has the same
// bits every run; k_tex_acc_finalize turns the planes into floats once.  Not finite / |w| >= 2^46: a flag, and the finalize pass writes NaN.
#define TEX_ACC_LO 72057594037927936.0      // 2^56
#define TEX_ACC_HI 65536.0                  // 2^16
struct TexAcc { long long* lo; long long* hi; uint32_t* bad; };
__device__ __forceinline__ void tex_acc_split(float w, long long& hi, long long& lo, bool& bad) {
    if (!(fabsf(w) < 7.0e13f)) { bad = true; hi = 0; lo = 0; return; }
    const double wd = (double)w;
    hi = __double2ll_rn(wd * TEX_ACC_HI);
    lo = __double2ll_rn((wd - (double)hi * (1.0 / TEX_ACC_HI)) * TEX_ACC_LO);
}
__global__ void __launch_bounds__(256) k_tex_acc_finalize(TexAcc acc, long long n, int accumulate, float* __restrict__ out) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double v = (double)acc.hi[i] * (1.0 / TEX_ACC_HI) + (double)acc.lo[i] * (1.0 / TEX_ACC_LO);
    const float r = *acc.bad ? __int_as_float(0x7fc00000) : (float)v;
    out[i] = accumulate ? out[i] + r : r;
}
// k_tex_bwd_tiled<3> ('linear', wrap) for the fused view: dy is gathered on load -- the antialias pass's colour gradient (own dalbedo_aa + the four pairs' blends)
// times the sigmoid's derivative --, the tile's taps are pre-combined in an LDS table of 64-bit words, one integer atomic per distinct (texel, channel) goes out,
// and interpolate's backward for the texture coordinates is the epilogue (drast).
struct ViewTexBwd { const float* G; const uint8_t* hit; const uint8_t* pflag; const float* pair_alpha; const float* sig; const float4* rast; const float2* vt; const int3* ft; float4* drast; TexAcc acc; };
#define VTB_BITS 10
#define VTB_SLOTS (1 << VTB_BITS)      // slots of the workgroup's texel table (24 B of sums each + 4 B of key)
// The workgroup's 16 x 16 pixels add their texel gradients in LDS first, then flush one global 64-bit atomic per touched texel channel.  Round 4 (profiles/r04:
// elimination runs, 8 views 1024^2): of the kernel's 0.397 ms the flush was 0.22 and the hash probing (CAS loop) 0.06.  The flush walked the table in HASH order --
// every lane of an atomic instruction in a different 64-B line, and agent-scope atomics execute at the memory side, a line per request.  So: the texels a
// tile touches are nearly always a small rectangle of the texture.  WINDOW mode: the workgroup reduces the bounding box of its (unwrapped) taps; if it fits the table,
// slot = row-major position in the box (no keys, no probing) and the flush runs in texel order, three lanes per texel: neighbouring lanes, neighbouring addresses.
// A box that does not fit (minified or scattered uv) takes the hashed route as before.  Both routes add the same integers: the planes come out bit-identical.
__device__ __forceinline__ int wave_min_i(int v) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v = min(v, __shfl_xor(v, d, 64));
    return v;
}
__device__ __forceinline__ int wave_max_i(int v) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v = max(v, __shfl_xor(v, d, 64));
    return v;
}
__global__ void __launch_bounds__(256) k_view_tex_bwd(const float* __restrict__ tex, const float2* __restrict__ uv, int H, int W, int Ht, int Wt, ViewTexBwd z) {
    __shared__ uint32_t keys[VTB_SLOTS];
    __shared__ unsigned long long vals[VTB_SLOTS][3];
    __shared__ int box[4];      // min u, min v, max u, max v of the base taps (unwrapped texel coordinates) of the pixels that contribute
    for (int i = threadIdx.x; i < VTB_SLOTS; i += 256) { keys[i] = TEXT_EMPTY; vals[i][0] = 0ull; vals[i][1] = 0ull; vals[i][2] = 0ull; }
    if (threadIdx.x == 0) { box[0] = INT_MAX; box[1] = INT_MAX; box[2] = INT_MIN; box[3] = INT_MIN; }
    __syncthreads();
    const int px = blockIdx.x * 16 + (threadIdx.x & 15), py = blockIdx.y * 16 + (threadIdx.x >> 4);
    {   // blockIdx.z = view: the view's block of every pixel array (the texel planes are shared by all views)
        const size_t o = (size_t)blockIdx.z * H * W;
        uv += o; z.G += 3 * o; z.hit += 2 * o; z.pflag += o; z.pair_alpha += 2 * o; z.sig += 3 * o;
        if (z.rast) z.rast += o;
        if (z.drast) z.drast += o;
    }
    bool bad = false, any = false;
    int bu = 0, bv = 0;
    uint32_t tk[4] = {0u, 0u, 0u, 0u};
    float tw[4] = {0.f, 0.f, 0.f, 0.f}, g[3] = {0.f, 0.f, 0.f};
    if (px < W && py < H) {
        const size_t gid = (size_t)py * W + px;
        const float2 q = uv[gid];
        const float u = q.x * Wt - 0.5f, v = q.y * Ht - 0.5f;
        const float fu0 = floorf(u), fv0 = floorf(v), fu = u - fu0, fv = v - fv0;
        bu = (int)fu0; bv = (int)fv0;
        const int iu0 = wrapi(bu, Wt, 0), iu1 = wrapi(bu + 1, Wt, 0), iv0 = wrapi(bv, Ht, 0), iv1 = wrapi(bv + 1, Ht, 0);
        tk[0] = (uint32_t)(iv0 * Wt + iu0); tk[1] = (uint32_t)(iv0 * Wt + iu1); tk[2] = (uint32_t)(iv1 * Wt + iu0); tk[3] = (uint32_t)(iv1 * Wt + iu1);
        tw[0] = (1.f - fu) * (1.f - fv); tw[1] = fu * (1.f - fv); tw[2] = (1.f - fu) * fv; tw[3] = fu * fv;
        g[0] = z.G[3 * gid]; g[1] = z.G[3 * gid + 1]; g[2] = z.G[3 * gid + 2];
        const uint32_t pf = (uint32_t)z.pflag[gid] >> 4;
#pragma unroll
        for (int j = 0; j < 4; j++) {      // d albedo0 of this pixel: + alpha dy[dst] where it is pixel a of the pair, - alpha dy[dst] where it is pixel b
            if (!(pf >> j & 1u)) continue;
            size_t first; int d;
            aa_pair_of(j, px, py, W, gid, first, d);
            const AaPair pr = aa_pair_load(z.hit, z.pair_alpha, first, d, W);
            const float sa = pr.ia == gid ? pr.alpha : -pr.alpha;
#pragma unroll
            for (int c = 0; c < 3; c++) g[c] += sa * z.G[3 * pr.idst + c];
        }
#pragma unroll
        for (int c = 0; c < 3; c++) { const float sv = z.sig[3 * gid + c]; g[c] *= sv * (1.f - sv); any = any || g[c] != 0.f; }
        float gu = 0.f, gv = 0.f;
#pragma unroll
        for (int c = 0; c < 3; c++) {
            const float t00 = tex[(size_t)tk[0] * 3 + c], t10 = tex[(size_t)tk[1] * 3 + c], t01 = tex[(size_t)tk[2] * 3 + c], t11 = tex[(size_t)tk[3] * 3 + c];
            gu += g[c] * ((t10 - t00) * (1.f - fv) + (t11 - t01) * fv);
            gv += g[c] * ((t01 - t00) * (1.f - fu) + (t11 - t10) * fu);
        }
        if (z.drast) {      // k_interp_bwd for the two texture-coordinate attributes, same statements
            const int t = (int)z.rast[gid].w - 1;
            float4 dr = make_float4(0.f, 0.f, 0.f, 0.f);
            if (t >= 0) {
                const int3 vi = z.ft[t];
                const float2 a0 = z.vt[vi.x], a1 = z.vt[vi.y], a2 = z.vt[vi.z];
                const float g0 = gu * Wt, g1 = gv * Ht;
                dr.x = g0 * (a0.x - a2.x) + g1 * (a0.y - a2.y);
                dr.y = g0 * (a1.x - a2.x) + g1 * (a1.y - a2.y);
            }
            z.drast[gid] = dr;
        }
    }
    {   // the box of the contributing pixels' base taps, reduced per wave, then four LDS atomics per wave
        const int lo_u = wave_min_i(any ? bu : INT_MAX), lo_v = wave_min_i(any ? bv : INT_MAX), hi_u = wave_max_i(any ? bu : INT_MIN), hi_v = wave_max_i(any ? bv : INT_MIN);
        if ((threadIdx.x & 63) == 0 && lo_u != INT_MAX) { atomicMin(&box[0], lo_u); atomicMin(&box[1], lo_v); atomicMax(&box[2], hi_u); atomicMax(&box[3], hi_v); }
    }
    const bool use_hash = tex_taps_shared(tk);      // (hashed route only: does a lane share taps with its neighbour at all?)
    __syncthreads();
    const int u0 = box[0], v0 = box[1];
    const long long bw = (long long)box[2] + 2 - u0, bh = (long long)box[3] + 2 - v0;      // + the second tap of the last column / row
    const bool window = u0 != INT_MAX && bw * bh <= VTB_SLOTS;
    if (any) {
#pragma unroll
        for (int t = 0; t < 4; t++) {
            int slot = -1;
            if (window) slot = (bv + (t >> 1) - v0) * (int)bw + (bu + (t & 1) - u0);
            else {
                uint32_t h = (tk[t] * 2654435761u) >> (32 - VTB_BITS);
                for (int probe = 0; use_hash && probe < 32; probe++) {   // bounded: a full table falls back to the direct scatter
                    const uint32_t old = atomicCAS(&keys[h], TEXT_EMPTY, tk[t]);
                    if (old == TEXT_EMPTY || old == tk[t]) { slot = (int)h; break; }
                    h = (h + 1) & (VTB_SLOTS - 1);
                }
            }
#pragma unroll
            for (int c = 0; c < 3; c++) {
                const float w = g[c] * tw[t];
                if (w == 0.f) continue;
                long long hi, lo;
                tex_acc_split(w, hi, lo, bad);
                if (hi) atomicAdd((unsigned long long*)&z.acc.hi[(size_t)tk[t] * 3 + c], (unsigned long long)hi);
                if (!lo) continue;
                if (slot >= 0) atomicAdd(&vals[slot][c], (unsigned long long)lo);
                else atomicAdd((unsigned long long*)&z.acc.lo[(size_t)tk[t] * 3 + c], (unsigned long long)lo);
            }
        }
    }
    if (bad) *z.acc.bad = 1u;
    __syncthreads();
    if (window) {      // texel order, lane = (texel, channel): the atomics of one instruction fall into a few 64-B lines
        const int n = (int)(bw * bh) * 3;
        const unsigned long long* flat = &vals[0][0];
        for (int e = threadIdx.x; e < n; e += 256) {
            const unsigned long long v = flat[e];
            if (!v) continue;
            const int slot = e / 3, c = e - 3 * slot, ty = slot / (int)bw, tx = slot - ty * (int)bw;
            const size_t texel = (size_t)wrapi(v0 + ty, Ht, 0) * Wt + wrapi(u0 + tx, Wt, 0);
            atomicAdd((unsigned long long*)&z.acc.lo[texel * 3 + c], v);
        }
        return;
    }
    for (int i = threadIdx.x; i < VTB_SLOTS; i += 256) {
        const uint32_t k = keys[i];
        if (k == TEXT_EMPTY) continue;
#pragma unroll
        for (int c = 0; c < 3; c++) {
            const unsigned long long v = vals[i][c];
            if (v) atomicAdd((unsigned long long*)&z.acc.lo[(size_t)k * 3 + c], v);
        }
    }
}
// shade backward for caller-supplied image gradients (c3d_mesh_view_bwd; the training step uses k_view_loss_shade_bwd)
__global__ void __launch_bounds__(256) k_view_shade_bwd(const float* __restrict__ albedo, const float* __restrict__ alpha, ViewBg bg, long long P,   /* single view */
                                                        const float* __restrict__ dimage, const float* __restrict__ dalpha_out, float* __restrict__ dalbedo_aa,
                                                        float* __restrict__ dalbedo0, float* __restrict__ dalpha) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= P) return;
    const float ar = alpha[i], a = fminf(fmaxf(ar, 0.f), 1.f);
    float da = dalpha_out ? dalpha_out[i] : 0.f;
#pragma unroll
    for (int c = 0; c < 3; c++) {
        const float al = albedo[3 * i + c], val = a * al + (1.f - a) * bg.c[c];
        const float g = (val >= 0.f && val <= 1.f) ? (dimage ? dimage[3 * i + c] : 0.f) : 0.f;     // torch.clamp passes the gradient on the closed interval
        dalbedo_aa[3 * i + c] = g * a;
        if (dalbedo0) dalbedo0[3 * i + c] = g * a;        // scattering antialias backward: dcolor starts as a copy of dy, the blends are added on top
        da += g * (al - bg.c[c]);
    }
    dalpha[i] = (ar >= 0.f && ar <= 1.f) ? da : 0.f;
}
// Round 3: the training step's pixel loss (k_mesh_pixel_loss) and the shade backward in ONE pass: d/dimage never goes through HBM.
// L_v = w * mean_{c,p} ((image - target) m)^2 (+ the MS-SSIM gradient planes, when there are any); one loss partial per workgroup.
// per view (blockIdx.y): image / dssim_chw / loss_part are the batch arrays' bases (view b at + b * 3 P, + b * 3 P, + b * loss_stride), targets and masks per-view pointers
struct ViewLossIn { const float* image; const float* target_chw[MESH_MAX_VIEWS]; const float* mask[MESH_MAX_VIEWS]; const float* dssim_chw; float w; float* loss_part; size_t loss_stride; };
__global__ void __launch_bounds__(256) k_view_loss_shade_bwd(const float* __restrict__ albedo, const float* __restrict__ alpha, ViewBgs bgs, long long P, ViewLossIn lin,
                                                              float* __restrict__ dalbedo_aa, float* __restrict__ dalbedo0, float* __restrict__ dalpha) {
    __shared__ float red[4];
    const int bview = blockIdx.y;
    const ViewBg bg = bgs.v[bview];
    struct { const float* image; const float* target_chw; const float* mask; const float* dssim_chw; float w; float* loss_part; } li =
        {lin.image + 3 * (size_t)bview * P, lin.target_chw[bview], lin.mask[bview], lin.dssim_chw ? lin.dssim_chw + 3 * (size_t)bview * P : nullptr, lin.w,
         lin.loss_part ? lin.loss_part + (size_t)bview * lin.loss_stride : nullptr};
    albedo += 3 * (size_t)bview * P; alpha += (size_t)bview * P; dalbedo_aa += 3 * (size_t)bview * P; dalpha += (size_t)bview * P;
    if (dalbedo0) dalbedo0 += 3 * (size_t)bview * P;
    float l = 0.f;
    const float inv = 1.f / (3.f * (float)P);
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < P; i += (long long)gridDim.x * blockDim.x) {
        const float m = li.mask ? li.mask[i] : 1.f;
        const float ar = alpha[i], a = fminf(fmaxf(ar, 0.f), 1.f);
        float da = 0.f;
#pragma unroll
        for (int c = 0; c < 3; c++) {
            const float d = (li.image[3 * i + c] - li.target_chw[c * P + i]) * m;
            l += d * d;
            const float dimg = 2.f * li.w * inv * d * m + (li.dssim_chw ? li.dssim_chw[c * P + i] : 0.f);
            const float al = albedo[3 * i + c], val = a * al + (1.f - a) * bg.c[c];
            const float g = (val >= 0.f && val <= 1.f) ? dimg : 0.f;
            dalbedo_aa[3 * i + c] = g * a;
            if (dalbedo0) dalbedo0[3 * i + c] = g * a;
            da += g * (al - bg.c[c]);
        }
        dalpha[i] = (ar >= 0.f && ar <= 1.f) ? da : 0.f;
    }
    l = c3d_wave_sum(l * li.w * inv);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = l;
    __syncthreads();
    if (threadIdx.x == 0 && li.loss_part) li.loss_part[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}
// The fused view has ONE form (round 3's): fused pixel passes (k_view_pixel_fwd; the pixel loss in the shade backward; sigmoid', the antialias colour gradient and
// interpolate's backward inside the texture backward) and NO float atomics (gathering antialias, integer texel-gradient planes): bit-reproducible.  Round 2's
// one-launch-per-op variant with scattering atomics (C3D_MESH_PIXEL_FUSED=0) is gone; profiles/r03 holds the comparison.
namespace {
// state of B views, every array [B, ...] (B = 1: the layout c3d_hip/mesh_fused.py reads rast / v_clip from)
struct ViewState { float* vclip; float* rast; float* texc; float* albedo0; float* albedo_aa; float* cov_aa; uint8_t* hit; float* pair_alpha; uint8_t* pflag; uint32_t* owned; int owned_words; uint8_t* sil; size_t bytes; };
void carve_view_state(char* base, int B, int V, int T, int H, int W, ViewState& st) {
    size_t off = 0;
    const size_t P = (size_t)B * H * W;
    auto take = [&](size_t b) { char* p = base ? base + off : nullptr; off += c3d_align(b); return (float*)p; };
    st.vclip = take(16 * (size_t)B * (size_t)(V > 0 ? V : 1)); st.rast = take(16 * P); st.texc = take(8 * P);
    st.albedo0 = take(12 * P); st.albedo_aa = take(12 * P); st.cov_aa = take(4 * P);
    st.hit = (uint8_t*)take(2 * P);
    st.pair_alpha = take(8 * P);
    st.pflag = (uint8_t*)take(P);
    st.owned_words = ((T > 0 ? T : 1) + 31) / 32;
    st.owned = (uint32_t*)take(4 * (size_t)B * st.owned_words);      // TriOwned
    st.sil = (uint8_t*)take((size_t)B * (size_t)(T > 0 ? T : 1));    // silhouette bits per view and triangle (k_aa_sil_bits); last: the offsets c3d_hip/mesh_fused.py reads rast / v_clip at stay
    st.bytes = off;
}
struct ViewBwdScratch { float* dalbedo_aa; float* dcov; float* drast; float* dpos_r; void* ras; TexAcc acc; size_t bytes; };
// Ht, Wt > 0: the scratch also holds the integer planes of the texel gradient (a single view's backward; the multi-view step owns ONE pair for all its views)
void carve_view_bwd(char* base, int B, int V, int T, int H, int W, int Ht, int Wt, ViewBwdScratch& sc) {
    size_t off = 0;
    const size_t P = (size_t)B * H * W, v = (size_t)B * (size_t)(V > 0 ? V : 1);
    auto take = [&](size_t b) { char* p = base ? base + off : nullptr; off += c3d_align(b); return p; };
    sc.dalbedo_aa = (float*)take(12 * P); sc.dcov = (float*)take(4 * P); sc.drast = (float*)take(16 * P);
    sc.dpos_r = (float*)take(16 * v);
    sc.ras = take(c3d_mesh_rasterize_bwd_scratch_bytes(B, T));
    const size_t ntex = 3 * (size_t)(Ht > 0 ? Ht : 0) * (size_t)(Wt > 0 ? Wt : 0);
    sc.acc.lo = sc.acc.hi = nullptr; sc.acc.bad = nullptr;
    if (ntex) { sc.acc.lo = (long long*)take(8 * ntex + 64); sc.acc.hi = (long long*)take(8 * ntex); sc.acc.bad = base ? (uint32_t*)(sc.acc.lo + ntex) : nullptr; }   // the flag word rides behind the first plane: one memset covers both
    sc.bytes = off;
}

// forward of B <= MESH_MAX_VIEWS views of one mesh: 6 launches + 2 clears whatever B is.  image [B, H, W, 3], alpha [B, H, W, 1].
int mesh_views_fwd(const c3d_mesh_view* views, int B, const float* v, const float* v_offsets, const int32_t* f, const float* vt, const int32_t* ft, const float* raw_albedo,
                   const void* aa_topology, void* raster_scratch, void* state, float* image, float* alpha, hipStream_t s) {
    const c3d_mesh_view* d = &views[0];
    const int V = d->V, T = d->T, H = d->H, W = d->W;
    const long long P = (long long)H * W, BP = P * B;
    ViewState st; carve_view_state((char*)state, B, V, T, H, W, st);
    ViewMats Ms; ViewBgs bgs;
    for (int b = 0; b < B; b++) { for (int i = 0; i < 16; i++) Ms.v[b].m[i] = views[b].clip_from_world[i]; for (int i = 0; i < 3; i++) bgs.v[b].c[i] = views[b].bg[i]; }
    int rc;
    hipLaunchKernelGGL(k_view_transform_fwd, dim3(c3d_cdiv(V, 256), B), dim3(256), 0, s, v, v_offsets, Ms, V, (float4*)st.vclip);
    // depth | id buffer of all views (the rasterizer's kernels take a batch), then ONE pixel pass: resolve -> interpolate(uv) -> texture -> sigmoid (k_view_pixel_fwd).
    // texture(..., filter_mode='linear') ignores uv_da (diff_mesh_renderer.py:110 passes it all the same): no pixel differentials of uv are produced
    C3D_CHECK(hipMemsetAsync(st.owned, 0, 4 * (size_t)B * st.owned_words, s));
    if ((rc = mesh_rasterize_impl(st.vclip, f, B, V, T, H, W, nullptr, raster_scratch, st.rast, nullptr, (c3d_stream_t)s, false))) return rc;
    { C3dProfScope ps(C3D_P_MESH_INTERPOLATE, s);
      hipLaunchKernelGGL(k_view_pixel_fwd, dim3(c3d_cdiv(BP, 256)), dim3(256), 0, s, (const float4*)st.vclip, (const int3*)f, (const float2*)vt, (const int3*)ft, raw_albedo, B, V, H, W,
                         d->Ht, d->Wt, (const unsigned long long*)raster_scratch, (float4*)st.rast, (float2*)st.texc, st.albedo0, st.owned, st.owned_words); }
    {   // silhouette analysis per pair, then every pixel gathers its blends and shades: no atomics (k_view_shade_fwd_g)
        C3dProfScope ps(C3D_P_MESH_ANTIALIAS, s);
        const uint8_t* sil = st.sil;
        hipLaunchKernelGGL(k_aa_sil_bits, dim3(c3d_cdiv(T, 256), B), dim3(256), 0, s, (const float4*)st.vclip, (const int3*)f, (const EdgeSlot*)aa_topology, edge_table_size(T) - 1, V, T, st.sil);
        hipLaunchKernelGGL(k_aa2_pairs, dim3(c3d_cdiv(BP * 2, 256)), dim3(256), 0, s, (const float4*)st.rast, (const float4*)st.vclip, (const int3*)f,
                           (const EdgeSlot*)aa_topology, edge_table_size(T) - 1, B, V, H, W, st.pair_alpha, st.hit, sil, T);
        hipLaunchKernelGGL(k_view_shade_fwd_g, dim3(c3d_cdiv(BP, 256)), dim3(256), 0, s, st.albedo0, (const float4*)st.rast, st.hit, st.pair_alpha, bgs, B, H, W,
                           st.albedo_aa, st.cov_aa, image, alpha, st.pflag);
    }
    C3D_LAUNCH_CHECK();
    return 0;
}

// backward of the same B views.  li (training step): the pixel loss of every view is evaluated in the first kernel; otherwise B = 1 and dimage / dalpha come from the caller.
// d_v (optional): view b's vertex gradient at d_v + b * d_v_stride floats.  step_acc: the step's shared integer texel planes (the caller finalizes them); NULL: this
// call owns a pair in its scratch, clears and finalizes it into d_raw_albedo.
int mesh_views_bwd(const c3d_mesh_view* views, int B, const int32_t* f, const float* vt, const int32_t* ft, const float* raw_albedo, const void* vertex_topology, void* scratch,
                   const void* state, const float* dimage, const float* dalpha, float* d_raw_albedo, float* d_v, size_t d_v_stride, hipStream_t s, bool zero_dtex,
                   const ViewLossIn* li, int loss_blocks, const TexAcc* step_acc) {
    const c3d_mesh_view* d = &views[0];
    const int V = d->V, T = d->T, H = d->H, W = d->W;
    const long long P = (long long)H * W;
    ViewState st; carve_view_state((char*)state, B, V, T, H, W, st);
    ViewBwdScratch sc; carve_view_bwd((char*)scratch, B, V, T, H, W, step_acc ? 0 : d->Ht, step_acc ? 0 : d->Wt, sc);
    ViewMats Ms; ViewBgs bgs;
    for (int b = 0; b < B; b++) { for (int i = 0; i < 16; i++) Ms.v[b].m[i] = views[b].clip_from_world[i]; for (int i = 0; i < 3; i++) bgs.v[b].c[i] = views[b].bg[i]; }
    int rc;
    {
        C3dProfScope ps(C3D_P_OTHER, s);
        if (li) hipLaunchKernelGGL(k_view_loss_shade_bwd, dim3((unsigned)loss_blocks, B), dim3(256), 0, s, st.albedo_aa, st.cov_aa, bgs, P, *li, sc.dalbedo_aa, (float*)nullptr, sc.dcov);
        else hipLaunchKernelGGL(k_view_shade_bwd, dim3(c3d_cdiv(P, 256)), dim3(256), 0, s, st.albedo_aa, st.cov_aa, bgs.v[0], P, dimage, dalpha, sc.dalbedo_aa, (float*)nullptr, sc.dcov);
    }
    // Gathers all the way (round 3): the texture backward gathers the antialias pass's colour gradient on load (x sigmoid'), adds the texel gradients into
    // integer planes and leaves drast; the per-triangle pass of the rasterizer's backward also gathers the antialias pass's position gradient.
    const size_t ntex = 3 * (size_t)d->Ht * d->Wt;
    const TexAcc acc = step_acc ? *step_acc : sc.acc;
    {
        C3dProfScope ps(C3D_P_MESH_TEXTURE_BWD, s);
        if (!step_acc) { C3D_CHECK(hipMemsetAsync(acc.lo, 0, 8 * ntex + 64, s)); C3D_CHECK(hipMemsetAsync(acc.hi, 0, 8 * ntex, s)); }
        const ViewTexBwd z{sc.dalbedo_aa, st.hit, st.pflag, st.pair_alpha, st.albedo0, d_v ? (const float4*)st.rast : nullptr, d_v ? (const float2*)vt : nullptr,
                           d_v ? (const int3*)ft : nullptr, d_v ? (float4*)sc.drast : nullptr, acc};
        hipLaunchKernelGGL(k_view_tex_bwd, dim3(c3d_cdiv(W, 16), c3d_cdiv(H, 16), B), dim3(256), 0, s, raw_albedo, (const float2*)st.texc, H, W, d->Ht, d->Wt, z);
        if (!step_acc) hipLaunchKernelGGL(k_tex_acc_finalize, dim3(c3d_cdiv((long long)ntex, 256)), dim3(256), 0, s, acc, (long long)ntex, zero_dtex ? 0 : 1, d_raw_albedo);
    }
    if (d_v) {
        const AaBwdIn aa{st.hit, st.pflag, st.pair_alpha, st.albedo0, sc.dalbedo_aa, sc.dcov};
        if ((rc = mesh_rasterize_bwd_gather(st.vclip, f, st.rast, sc.drast, nullptr, B, V, T, H, W, vertex_topology, sc.ras, sc.dpos_r, (c3d_stream_t)s, aa, TriOwned{st.owned, st.owned_words}))) return rc;
        hipLaunchKernelGGL(k_view_transform_bwd, dim3(c3d_cdiv(V, 256), B), dim3(256), 0, s, Ms, (const float4*)nullptr, (const float4*)sc.dpos_r, V, d_v, d_v_stride);
    }
    C3D_LAUNCH_CHECK();
    return 0;
}
}  // namespace

extern "C" {

size_t c3d_mesh_view_state_bytes(int32_t V, int32_t T, int32_t H, int32_t W) { ViewState st; carve_view_state(nullptr, 1, V, T, H, W, st); return st.bytes; }
size_t c3d_mesh_view_bwd_scratch_bytes(int32_t V, int32_t T, int32_t H, int32_t W, int32_t Ht, int32_t Wt) { ViewBwdScratch sc; carve_view_bwd(nullptr, 1, V, T, H, W, Ht, Wt, sc); return sc.bytes; }

int c3d_mesh_view_fwd(const c3d_mesh_view* d, const float* v, const float* v_offsets, const int32_t* f, const float* vt, const int32_t* ft, const float* raw_albedo,
                      const void* aa_topology, void* raster_scratch, void* state, float* image, float* alpha, c3d_stream_t stream) {
    MESH_REQUIRE(d && v && f && vt && ft && raw_albedo && aa_topology && raster_scratch && state && image && alpha, "c3d_mesh_view_fwd: NULL pointer");
    MESH_REQUIRE(d->V > 0 && d->T > 0 && d->H > 0 && d->W > 0 && d->Ht > 0 && d->Wt > 0 && d->Vt > 0, "c3d_mesh_view_fwd: empty mesh / image / texture");
    return mesh_views_fwd(d, 1, v, v_offsets, f, vt, ft, raw_albedo, aa_topology, raster_scratch, state, image, alpha, (hipStream_t)stream);
}

int c3d_mesh_view_bwd(const c3d_mesh_view* d, const float* v, const float* v_offsets, const int32_t* f, const float* vt, const int32_t* ft, const float* raw_albedo,
                      const void* aa_topology, const void* vertex_topology, void* scratch, const void* state, const float* dimage, const float* dalpha,
                      float* d_raw_albedo, float* d_v, c3d_stream_t stream) {
    (void)v; (void)v_offsets;
    MESH_REQUIRE(d && f && vt && ft && raw_albedo && aa_topology && scratch && state && d_raw_albedo, "c3d_mesh_view_bwd: NULL pointer");
    MESH_REQUIRE(dimage || dalpha, "c3d_mesh_view_bwd: no upstream gradient");
    MESH_REQUIRE(!d_v || vertex_topology, "c3d_mesh_view_bwd: the geometry gradient needs the vertex topology");
    return mesh_views_bwd(d, 1, f, vt, ft, raw_albedo, vertex_topology, scratch, state, dimage, dalpha, d_raw_albedo, d_v, 0, (hipStream_t)stream, true, nullptr, 0, nullptr);
}

}  // extern "C"

// ------------------------------------------------------------------------------------------------------------------------------------------
// Fused multi-view training step of DiffMesh (include/c3d_mesh.h: c3d_mesh_train_views): what DiffMesh.training_step does per step --
// render every view of the batch, image loss, backward, gradients summed over the views (diff_mesh.py:98-125 in the reference) -- as ONE
// library call.  Round 4: the views go through every stage TOGETHER (groups of <= MESH_MAX_VIEWS views, one launch per stage: ~16 launches per group
// where rounds 2-3 enqueued ~15 per VIEW on a pool of streams and were bound by the host's enqueue rate).
// ------------------------------------------------------------------------------------------------------------------------------------------
#include "../../include/c3d_loss.h"
int ms_value_grad_images(const float* const* x, const float* const* y, const float* const* mask, float* const* dy, int clamp_y, int B, int C, int H, int W, float grad_scale,
                         int accumulate, float va, float vb, float* out0, size_t out_stride, void* workspace, hipStream_t s);

// images [B, H, W, 3] -> planes [B, 3, H, W] (what the MS-SSIM kernels read)
__global__ void __launch_bounds__(256) k_mesh_hwc_to_chw(const float* __restrict__ hwc, long long P, float* __restrict__ chw) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= P) return;
    hwc += 3 * (size_t)blockIdx.y * P; chw += 3 * (size_t)blockIdx.y * P;
#pragma unroll
    for (int c = 0; c < 3; c++) chw[c * P + i] = hwc[3 * i + c];
}
#define MESH_LOSS_SLOTS 1032      // per view: <= 1024 workgroup partials of the pixel loss, then the view's MS-SSIM term at [1024]
__global__ void __launch_bounds__(256) k_mesh_sum_loss(const float* __restrict__ parts, int n_views, int nblk, int ssim, float* __restrict__ loss_out) {
    __shared__ float red[4];
    float total = 0.f;
    for (int v = 0; v < n_views; v++) {            // views in order, partials strided over the lanes, fixed tree: the same value every run
        const float* q = parts + (size_t)v * MESH_LOSS_SLOTS;
        float l = 0.f;
        for (int i = threadIdx.x; i < nblk; i += 256) l += q[i];
        l = c3d_wave_sum(l);
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = l;
        __syncthreads();
        if (threadIdx.x == 0) total += (red[0] + red[1] + red[2] + red[3]) + (ssim ? q[1024] : 0.f);
        __syncthreads();
    }
    if (threadIdx.x == 0) atomicAdd(loss_out, total);      // ONE add per call
}
// out (+)= sum_k in_k, k in fixed order (per-view buffers of the vertex gradient)
struct MeshSumSrc { int n; const float* p[64]; };
__global__ void __launch_bounds__(256) k_mesh_sum(MeshSumSrc src, long long count, int accumulate, float* __restrict__ out) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    float a = accumulate ? out[i] : 0.f;
    for (int k = 0; k < src.n; k++) a += src.p[k][i];
    out[i] = a;
}

namespace {
struct MeshStepWs { TexAcc acc; char* state; char* bwd; char* raster; float* image; float* alpha; float* image_chw; float* dssim; char* ms_ws; float* d_v; size_t d_v_stride; float* loss_part; size_t bytes; };
// G = views per group (<= MESH_MAX_VIEWS): the batched state of ONE group, reused by the groups of a step in stream order
void carve_mesh_step(char* base, int V, int T, int H, int W, int Ht, int Wt, int n_views, MeshStepWs& w) {
    size_t off = 0;
    const int G = n_views < 1 ? 1 : (n_views > MESH_MAX_VIEWS ? MESH_MAX_VIEWS : n_views);
    const size_t P = (size_t)G * H * W;
    auto take = [&](size_t b) { char* p = base ? base + off : nullptr; off += c3d_align(b); return p; };
    const size_t ntex = 3 * (size_t)Ht * Wt;
    w.acc.lo = (long long*)take(8 * ntex + 64); w.acc.hi = (long long*)take(8 * ntex); w.acc.bad = base ? (uint32_t*)(w.acc.lo + ntex) : nullptr;      // the integer planes ALL views of the step add their texel gradients into
    { ViewState st; carve_view_state(nullptr, G, V, T, H, W, st); w.state = take(st.bytes); }
    { ViewBwdScratch sc; carve_view_bwd(nullptr, G, V, T, H, W, 0, 0, sc); w.bwd = take(sc.bytes); }
    w.raster = take(c3d_mesh_raster_scratch_bytes(G, H, W, T));
    w.image = (float*)take(12 * P); w.alpha = (float*)take(4 * P); w.image_chw = (float*)take(12 * P); w.dssim = (float*)take(12 * P);
    w.ms_ws = take((H > 160 && W > 160) ? c3d_msssim_workspace_bytes(G, 3, H, W) : 0);
    w.d_v_stride = c3d_align(12 * (size_t)(V > 0 ? V : 1));
    w.d_v = (float*)take(w.d_v_stride * (size_t)(n_views > 0 ? n_views : 1));
    w.loss_part = (float*)take(sizeof(float) * MESH_LOSS_SLOTS * (size_t)(n_views > 0 ? n_views : 1));
    w.bytes = off;
}
}  // namespace

extern "C" {

size_t c3d_mesh_step_workspace_bytes(int32_t V, int32_t T, int32_t H, int32_t W, int32_t Ht, int32_t Wt, int32_t n_views, int32_t lanes) {
    (void)lanes;      // kept in the signature (ABI of rounds 2-3): the views of a step now go through every stage together instead of on a pool of streams
    MeshStepWs w;
    carve_mesh_step(nullptr, V, T, H, W, Ht, Wt, n_views, w);
    return w.bytes;
}

int c3d_mesh_train_views(const c3d_mesh_view* views, int32_t n_views, const float* v, const float* v_offsets, const int32_t* f, const float* vt, const int32_t* ft,
                         const float* raw_albedo, const void* aa_topology, const void* vertex_topology, const float* const* target_chw, const float* const* mask,
                         const c3d_mesh_step_loss* loss, float* d_raw_albedo, float* d_v_offsets, float* loss_out, int32_t accumulate, int32_t lanes, void* workspace,
                         c3d_stream_t stream) {
    hipStream_t s0 = (hipStream_t)stream;
    if (n_views <= 0) return 0;
    MESH_REQUIRE(views && v && f && vt && ft && raw_albedo && aa_topology && target_chw && loss && d_raw_albedo && workspace, "c3d_mesh_train_views: NULL pointer");
    MESH_REQUIRE(!d_v_offsets || vertex_topology, "c3d_mesh_train_views: the geometry gradient needs the vertex topology");
    MESH_REQUIRE(lanes >= 1 && lanes <= C3D_MAX_LANES, "c3d_mesh_train_views: lanes must be in [1, 8]");
    MESH_REQUIRE(n_views <= 64, "c3d_mesh_train_views: at most 64 views per call");
    const c3d_mesh_view& d0 = views[0];
    for (int i = 0; i < n_views; i++) {
        const c3d_mesh_view& d = views[i];
        MESH_REQUIRE(d.V == d0.V && d.T == d0.T && d.Vt == d0.Vt && d.H == d0.H && d.W == d0.W && d.Ht == d0.Ht && d.Wt == d0.Wt, "c3d_mesh_train_views: all views must share the mesh and the resolution");
        MESH_REQUIRE(target_chw[i], "c3d_mesh_train_views: NULL target");
    }
    MESH_REQUIRE(d0.V > 0 && d0.T > 0 && d0.H > 0 && d0.W > 0 && d0.Ht > 0 && d0.Wt > 0 && d0.Vt > 0, "c3d_mesh_train_views: empty mesh / image / texture");
    const bool ssim = loss->w_ssim != 0.f;
    MESH_REQUIRE(!ssim || (d0.H > 160 && d0.W > 160), "c3d_mesh_train_views: the MS-SSIM term needs image sides > 160");
    const long long P = (long long)d0.H * d0.W;
    const int nblk = c3d_cdiv(P, 256) < 1024 ? c3d_cdiv(P, 256) : 1024;
    const size_t ntex = 3 * (size_t)d0.Ht * d0.Wt;
    MeshStepWs w;
    carve_mesh_step((char*)workspace, d0.V, d0.T, d0.H, d0.W, d0.Ht, d0.Wt, n_views, w);
    // the shared integer planes are cleared once per step
    C3D_CHECK(hipMemsetAsync(w.acc.lo, 0, 8 * ntex + 64, s0));
    C3D_CHECK(hipMemsetAsync(w.acc.hi, 0, 8 * ntex, s0));
    const int G = n_views > MESH_MAX_VIEWS ? MESH_MAX_VIEWS : n_views;
    for (int v0 = 0; v0 < n_views; v0 += G) {
        const int g = (n_views - v0) < G ? (n_views - v0) : G;
        int rc;
        if ((rc = mesh_views_fwd(views + v0, g, v, v_offsets, f, vt, ft, raw_albedo, aa_topology, w.raster, w.state, w.image, w.alpha, s0))) return rc;
        float* lp = w.loss_part + (size_t)v0 * MESH_LOSS_SLOTS;      // these views' loss partials: the value is summed in a fixed order at the end
        if (ssim) {   // + scale * w_ssim * (1 - MS-SSIM(target m, image m)) of every view: values into the views' slots, gradients (planes) into w.dssim -- one batched call
            C3dProfScope ps(C3D_P_OTHER, s0);
            const float ws_ = loss->scale * loss->w_ssim;
            hipLaunchKernelGGL(k_mesh_hwc_to_chw, dim3(c3d_cdiv(P, 256), g), dim3(256), 0, s0, w.image, P, w.image_chw);
            const float* ys[MESH_MAX_VIEWS]; float* dys[MESH_MAX_VIEWS];
            for (int i = 0; i < g; i++) { ys[i] = w.image_chw + 3 * (size_t)i * P; dys[i] = w.dssim + 3 * (size_t)i * P; }
            if ((rc = ms_value_grad_images(target_chw + v0, ys, mask ? mask + v0 : nullptr, dys, 0, g, 3, d0.H, d0.W, -ws_, 0, ws_, -ws_, loss_out ? lp + 1024 : nullptr,
                                           sizeof(float) * MESH_LOSS_SLOTS, w.ms_ws, s0))) return rc;
        }
        ViewLossIn li{};      // the pixel loss inside the backward pass's first kernel: d/dimage is never materialised
        li.image = w.image; li.dssim_chw = ssim ? w.dssim : nullptr; li.w = loss->scale * loss->w_mse; li.loss_part = loss_out ? lp : nullptr; li.loss_stride = MESH_LOSS_SLOTS;
        for (int i = 0; i < g; i++) { li.target_chw[i] = target_chw[v0 + i]; li.mask[i] = mask ? mask[v0 + i] : nullptr; }
        float* dv = d_v_offsets ? (float*)((char*)w.d_v + (size_t)v0 * w.d_v_stride) : nullptr;
        if ((rc = mesh_views_bwd(views + v0, g, f, vt, ft, raw_albedo, vertex_topology, w.bwd, w.state, nullptr, nullptr, nullptr, dv, w.d_v_stride / sizeof(float), s0, false, &li, nblk, &w.acc))) return rc;
    }
    {
        C3dProfScope ps(C3D_P_OTHER, s0);
        hipLaunchKernelGGL(k_tex_acc_finalize, dim3(c3d_cdiv((long long)ntex, 256)), dim3(256), 0, s0, w.acc, (long long)ntex, accumulate, d_raw_albedo);
        if (d_v_offsets) {
            MeshSumSrc b; b.n = n_views;
            for (int i = 0; i < n_views; i++) b.p[i] = (const float*)((const char*)w.d_v + (size_t)i * w.d_v_stride);
            hipLaunchKernelGGL(k_mesh_sum, dim3(c3d_cdiv(3ll * d0.V, 256)), dim3(256), 0, s0, b, 3ll * d0.V, accumulate, d_v_offsets);
        }
        if (loss_out) hipLaunchKernelGGL(k_mesh_sum_loss, dim3(1), dim3(256), 0, s0, w.loss_part, n_views, nblk, ssim ? 1 : 0, loss_out);
    }
    C3D_LAUNCH_CHECK();
    return 0;
}

}  // extern "C"
