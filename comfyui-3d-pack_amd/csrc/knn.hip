// knn.hip -- mean squared distance to the 3 nearest neighbours (include/c3d_knn.h): the scale initialisation of GaussianModel.create_from_pcd
// (reference main_3DGS_renderer.py:407-433 through simple_knn.distCUDA2).  Uniform grid + the stable radix sort of the binning stage; exact.
#include "../../include/c3d_knn.h"
#include "c3d_common.h"

#define KNN_MAX_DIM 256     // cells per axis (24-bit cell ids: three 8-bit sort passes)

struct KnnGrid { float lo[3]; float inv_cell; float cell; int dim[3]; };

struct KnnWs { uint32_t* key[2]; uint32_t* val[2]; uint32_t* cell_start; void* tmp; size_t bytes; };
__device__ __forceinline__ int knn_cell_coord(float x, float lo, float inv, int dim) {
    int c = (int)floorf((x - lo) * inv);
    return min(max(c, 0), dim - 1);
}
__global__ void __launch_bounds__(256) k_knn_cells(const float* __restrict__ pts, int N, KnnGrid g, uint32_t* __restrict__ key) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    const int cx = knn_cell_coord(pts[3 * i], g.lo[0], g.inv_cell, g.dim[0]);
    const int cy = knn_cell_coord(pts[3 * i + 1], g.lo[1], g.inv_cell, g.dim[1]);
    const int cz = knn_cell_coord(pts[3 * i + 2], g.lo[2], g.inv_cell, g.dim[2]);
    key[i] = (uint32_t)((cz * g.dim[1] + cy) * g.dim[0] + cx);
}
// cell_start[c] = first sorted slot with key >= c (c in [0, cells]); same construction as the mesh vertex topology
__global__ void __launch_bounds__(256) k_knn_starts(const uint32_t* __restrict__ skey, int N, int cells, uint32_t* __restrict__ start) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i > N) return;
    const long long prev = i ? (long long)skey[i - 1] : -1ll, cur = i < N ? (long long)skey[i] : (long long)cells;
    for (long long c = prev + 1; c <= cur && c <= (long long)cells; c++) start[c] = (uint32_t)i;
}
__device__ __forceinline__ void knn_push(float d2, float best[3]) {
    if (d2 < best[2]) {
        if (d2 < best[1]) { best[2] = best[1]; if (d2 < best[0]) { best[1] = best[0]; best[0] = d2; } else best[1] = d2; }
        else best[2] = d2;
    }
}
__global__ void __launch_bounds__(256) k_knn_query(const float* __restrict__ pts, int N, KnnGrid g, const uint32_t* __restrict__ order,
                                                    const uint32_t* __restrict__ start, float* __restrict__ out) {
    const int s = blockIdx.x * blockDim.x + threadIdx.x;    // sorted slot: neighbouring lanes work in neighbouring cells
    if (s >= N) return;
    const uint32_t i = order[s];
    const float px = pts[3 * i], py = pts[3 * i + 1], pz = pts[3 * i + 2];
    const int cx = knn_cell_coord(px, g.lo[0], g.inv_cell, g.dim[0]), cy = knn_cell_coord(py, g.lo[1], g.inv_cell, g.dim[1]);
    const int cz = knn_cell_coord(pz, g.lo[2], g.inv_cell, g.dim[2]);
    float best[3] = {3.0e38f, 3.0e38f, 3.0e38f};
    const int rmax = max(g.dim[0], max(g.dim[1], g.dim[2]));
    for (int r = 0; r <= rmax; r++) {
        // shell r of the cube around the home cell (r = 0: the cell itself)
        for (int dz = -r; dz <= r; dz++) {
            const int z = cz + dz;
            if (z < 0 || z >= g.dim[2]) continue;
            for (int dy = -r; dy <= r; dy++) {
                const int y = cy + dy;
                if (y < 0 || y >= g.dim[1]) continue;
                const bool face = (dz == -r || dz == r || dy == -r || dy == r);
                const int step = (face || r == 0) ? 1 : 2 * r;            // interior rows of the shell: only the two end cells
                for (int dx = -r; dx <= r; dx += step) {
                    const int x = cx + dx;
                    if (x < 0 || x >= g.dim[0]) continue;
                    const uint32_t c = (uint32_t)((z * g.dim[1] + y) * g.dim[0] + x);
                    for (uint32_t t = start[c], e = start[c + 1]; t < e; t++) {
                        const uint32_t j = order[t];
                        if (j == i) continue;
                        const float ddx = pts[3 * j] - px, ddy = pts[3 * j + 1] - py, ddz = pts[3 * j + 2] - pz;
                        knn_push(ddx * ddx + ddy * ddy + ddz * ddz, best);
                    }
                }
            }
        }
        // everything outside the searched cube is farther than the distance from the point to the cube's faces (>= r cells minus nothing:
        // the point lies inside its home cell, so at least r * cell away from any unsearched cell)
        const float reach = (float)r * g.cell;
        if (best[2] <= reach * reach) break;
    }
    float sum = 0.f;
#pragma unroll
    for (int k = 0; k < 3; k++) sum += best[k] < 3.0e38f ? best[k] : 0.f;
    out[i] = sum * (1.f / 3.f);
}

static size_t knn_bytes(int N, KnnWs& w, char* base) {
    size_t off = 0;
    auto take = [&](size_t b) { char* p = base ? base + off : nullptr; off += c3d_align(b); return p; };
    const size_t n = (size_t)(N > 0 ? N : 1);
    w.key[0] = (uint32_t*)take(4 * n); w.key[1] = (uint32_t*)take(4 * n);
    w.val[0] = (uint32_t*)take(4 * n); w.val[1] = (uint32_t*)take(4 * n);
    w.cell_start = (uint32_t*)take(4 * ((size_t)KNN_MAX_DIM * KNN_MAX_DIM * KNN_MAX_DIM + 2 > 4 * n + 1026 ? 4 * n + 1026 : (size_t)KNN_MAX_DIM * KNN_MAX_DIM * KNN_MAX_DIM + 2));
    w.tmp = take(c3d_sort_tmp_bytes(n));
    w.bytes = off;
    return off;
}

extern "C" {
size_t c3d_knn_scratch_bytes(int32_t N) { KnnWs w; return knn_bytes(N, w, nullptr); }

int c3d_knn3_mean_dist2(const float* points, int32_t N, const float bbox_lo[3], const float bbox_hi[3], void* scratch, float* out, c3d_stream_t stream) {
    hipStream_t s = (hipStream_t)stream;
    if (N <= 0) return 0;
    if (!points || !bbox_lo || !bbox_hi || !scratch || !out) { c3d_set_error("c3d_knn3_mean_dist2: NULL pointer"); return -1; }
    KnnWs w; knn_bytes(N, w, (char*)scratch);
    // ~2 points per cell on average, at most 4 N + 1024 cells and KNN_MAX_DIM cells per axis
    KnnGrid g;
    double ext[3], vol = 1.0;
    for (int k = 0; k < 3; k++) {
        ext[k] = (double)bbox_hi[k] - (double)bbox_lo[k];
        if (!(ext[k] >= 0.0)) { c3d_set_error("c3d_knn3_mean_dist2: empty bounding box"); return -1; }
        if (ext[k] < 1e-12) ext[k] = 1e-12;
        vol *= ext[k];
        g.lo[k] = bbox_lo[k];
    }
    double cell = cbrt(vol / (0.5 * (double)N));
    const double longest = ext[0] > ext[1] ? (ext[0] > ext[2] ? ext[0] : ext[2]) : (ext[1] > ext[2] ? ext[1] : ext[2]);
    if (cell < longest / KNN_MAX_DIM) cell = longest / KNN_MAX_DIM;
    long long cells = 1;
    for (int k = 0; k < 3; k++) {
        int d = (int)(ext[k] / cell) + 1;
        if (d > KNN_MAX_DIM) d = KNN_MAX_DIM;
        if (d < 1) d = 1;
        g.dim[k] = d;
        cells *= d;
    }
    while (cells > 4ll * N + 1024) {          // flat point sets: coarsen until the start table fits
        cell *= 1.26;
        cells = 1;
        for (int k = 0; k < 3; k++) { int d = (int)(ext[k] / cell) + 1; if (d > KNN_MAX_DIM) d = KNN_MAX_DIM; g.dim[k] = d; cells *= d; }
    }
    g.cell = (float)cell;
    g.inv_cell = (float)(1.0 / cell);
    hipLaunchKernelGGL(k_knn_cells, dim3(c3d_cdiv(N, 256)), dim3(256), 0, s, points, N, g, w.key[0]);
    int bits = 1;
    while ((1ll << bits) < cells) bits++;
    int res = 0, rc;
    if ((rc = c3d_sort_pairs_u32(w.key[0], w.key[1], w.val[0], w.val[1], true, (size_t)N, bits, w.tmp, &res, s))) return rc;
    hipLaunchKernelGGL(k_knn_starts, dim3(c3d_cdiv(N + 1, 256)), dim3(256), 0, s, w.key[res], N, (int)cells, w.cell_start);
    hipLaunchKernelGGL(k_knn_query, dim3(c3d_cdiv(N, 256)), dim3(256), 0, s, points, N, g, w.val[res], w.cell_start, out);
    C3D_LAUNCH_CHECK();
    return 0;
}
}
