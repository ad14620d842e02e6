// densify.hip -- device-side densify / prune (include/c3d_densify.h).
// Reference: MVs_Algorithms/GaussianSplatting/main_3DGS_renderer.py:558-781 (three rounds of boolean-mask indexing + torch.cat per array).
// Here: classify -> four single-pass prefix sums (scan_sort.hip) -> one host read of four counts -> source-index list -> one gather launch for every array.
#include "../../include/c3d_densify.h"
#include "c3d_common.h"

struct DensifyPlan {
    uint8_t* kind;        // bit 0 survivor alive, bit 1 clone alive, bit 2 split parent (children alive), bit 3 split parent (any)
    uint32_t* fk;         // flags as uint32 for the scans (survivor alive / clone alive / children alive / split any) ...
    uint32_t* fc;
    uint32_t* fs;
    uint32_t* fa;
    uint32_t* pk;         // ... and their exclusive prefix sums
    uint32_t* pc;
    uint32_t* ps;
    uint32_t* pa;
    void* tmp[4];         // scan state
    size_t bytes;
};
static void carve_plan(char* base, int N, DensifyPlan& p) {
    size_t n = (size_t)(N > 0 ? N : 1), off = 0;
    auto take = [&](size_t b) { char* q = base ? base + off : nullptr; off += c3d_align(b); return q; };
    p.kind = (uint8_t*)take(n);
    p.fk = (uint32_t*)take(4 * n); p.fc = (uint32_t*)take(4 * n); p.fs = (uint32_t*)take(4 * n); p.fa = (uint32_t*)take(4 * n);
    p.pk = (uint32_t*)take(4 * n); p.pc = (uint32_t*)take(4 * n); p.ps = (uint32_t*)take(4 * n); p.pa = (uint32_t*)take(4 * n);
    for (int i = 0; i < 4; i++) p.tmp[i] = take(c3d_scan_tmp_bytes(n));
    p.bytes = off;
}

__global__ void __launch_bounds__(256) k_densify_classify(int N, const float* __restrict__ grad_accum, const float* __restrict__ denom, const float* __restrict__ scaling_raw,
                                                           const float* __restrict__ opacity_raw, float max_grad, float dense_extent, float min_opacity, float max_scale,
                                                           DensifyPlan p, uint32_t* __restrict__ counts) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const bool in = i < N;
    const int ii = in ? i : N - 1;                        // out-of-range lanes recompute the last point and contribute nothing (the ballot below needs the whole wave)
    float g = grad_accum[ii] / denom[ii];
    if (g != g) g = 0.f;                                   // 0 / 0: a point no view has seen yet
    g = fabsf(g);
    const float s = fmaxf(fmaxf(expf(scaling_raw[3 * ii]), expf(scaling_raw[3 * ii + 1])), expf(scaling_raw[3 * ii + 2]));
    const float o = 1.f / (1.f + expf(-opacity_raw[ii]));
    const bool hot = in && g >= max_grad, small = s <= dense_extent;
    const bool clone = hot && small, split = hot && !small;
    const bool dead_op = o < min_opacity;
    const bool dead_self = dead_op || (max_scale > 0.f && s > max_scale);
    const bool dead_child = dead_op || (max_scale > 0.f && s / 1.6f > max_scale);
    const uint32_t k = (!split && !dead_self) ? 1u : 0u, c = (clone && !dead_self) ? 1u : 0u, sc = (split && !dead_child) ? 1u : 0u, sa = split ? 1u : 0u;
    if (in) {
        p.kind[i] = (uint8_t)(k | (c << 1) | (sc << 2) | (sa << 3));
        p.fk[i] = k; p.fc[i] = c; p.fs[i] = sc; p.fa[i] = sa;
    }
    // candidates before pruning (what the reference reports as cloned / split / pruned): integer sums, order independent
    const uint64_t mc = __ballot(clone);
    if (c3d_lane() == 0 && mc) atomicAdd(&counts[4], (uint32_t)__popcll(mc));
}
__global__ void k_densify_counts(int N, DensifyPlan p, uint32_t* __restrict__ counts) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        counts[0] = p.pk[N - 1] + p.fk[N - 1]; counts[1] = p.pc[N - 1] + p.fc[N - 1];
        counts[2] = p.ps[N - 1] + p.fs[N - 1]; counts[3] = p.pa[N - 1] + p.fa[N - 1];
    }
}
__global__ void __launch_bounds__(256) k_densify_fill(int N, DensifyPlan p, uint32_t nK, uint32_t nC, uint32_t nS, uint32_t* __restrict__ src, uint8_t* __restrict__ fresh,
                                                       uint32_t* __restrict__ child_rank) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N) return;
    const uint32_t kd = p.kind[i];
    if (kd & 1u) { const uint32_t j = p.pk[i]; src[j] = (uint32_t)i; fresh[j] = 0; }
    if (kd & 2u) { const uint32_t j = nK + p.pc[i]; src[j] = (uint32_t)i; fresh[j] = 1; }
    if (kd & 4u) {
        const uint32_t r = p.ps[i], ra = p.pa[i];
#pragma unroll
        for (uint32_t c = 0; c < 2; c++) {
            const uint32_t j = nK + nC + c * nS + r;
            src[j] = (uint32_t)i; fresh[j] = 1;
            child_rank[c * nS + r] = ra;
        }
    }
}

struct GatherDesc { const float* src[C3D_GATHER_MAX]; float* dst[C3D_GATHER_MAX]; int row[C3D_GATHER_MAX]; int zero[C3D_GATHER_MAX]; int n; };
// one workgroup row-block x all arrays: a wave copies rows of one array with lane-consecutive floats (rows are 1 .. 45 floats wide)
__global__ void __launch_bounds__(256) k_gather_rows(GatherDesc d, const uint32_t* __restrict__ idx, const uint8_t* __restrict__ fresh, long long M) {
    const int a = blockIdx.y;
    const int row = d.row[a];
    const float* __restrict__ s = d.src[a];
    float* __restrict__ o = d.dst[a];
    const bool zf = d.zero[a] != 0;
    const long long total = M * row;
    for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < total; e += (long long)gridDim.x * blockDim.x) {
        const long long j = e / row;
        const int c = (int)(e - j * row);
        o[e] = (zf && fresh && fresh[j]) ? 0.f : s[(size_t)idx[j] * row + c];
    }
}

extern "C" {
size_t c3d_densify_plan_bytes(int32_t N) { DensifyPlan p; carve_plan(nullptr, N, p); return p.bytes + 256; }

int c3d_densify_plan(int32_t N, const float* grad_accum, const float* denom, const float* scaling_raw, const float* opacity_raw, float max_grad, float dense_extent,
                     float min_opacity, float max_scale, void* plan, uint32_t* counts, c3d_stream_t stream) {
    hipStream_t s = (hipStream_t)stream;
    if (!counts) { c3d_set_error("c3d_densify_plan: counts is NULL"); return -1; }
    if (N <= 0) { C3D_CHECK(hipMemsetAsync(counts, 0, 32, s)); return 0; }
    if (!grad_accum || !denom || !scaling_raw || !opacity_raw || !plan) { c3d_set_error("c3d_densify_plan: NULL pointer"); return -1; }
    DensifyPlan p; carve_plan((char*)plan, N, p);
    C3D_CHECK(hipMemsetAsync(counts, 0, 32, s));
    hipLaunchKernelGGL(k_densify_classify, dim3(c3d_cdiv(N, 256)), dim3(256), 0, s, N, grad_accum, denom, scaling_raw, opacity_raw, max_grad, dense_extent, min_opacity, max_scale, p, counts);
    C3D_LAUNCH_CHECK();
    int rc;
    if ((rc = c3d_scan_u32(p.fk, p.pk, (size_t)N, true, p.tmp[0], s))) return rc;
    if ((rc = c3d_scan_u32(p.fc, p.pc, (size_t)N, true, p.tmp[1], s))) return rc;
    if ((rc = c3d_scan_u32(p.fs, p.ps, (size_t)N, true, p.tmp[2], s))) return rc;
    if ((rc = c3d_scan_u32(p.fa, p.pa, (size_t)N, true, p.tmp[3], s))) return rc;
    hipLaunchKernelGGL(k_densify_counts, dim3(1), dim3(64), 0, s, N, p, counts);
    C3D_LAUNCH_CHECK();
    return 0;
}

int c3d_densify_fill(int32_t N, const void* plan, const uint32_t* counts_host, uint32_t* src, uint8_t* fresh, uint32_t* child_rank, c3d_stream_t stream) {
    if (N <= 0) return 0;
    if (!plan || !counts_host || !src || !fresh) { c3d_set_error("c3d_densify_fill: NULL pointer"); return -1; }
    if (counts_host[2] && !child_rank) { c3d_set_error("c3d_densify_fill: child_rank is NULL"); return -1; }
    DensifyPlan p; carve_plan((char*)plan, N, p);
    hipLaunchKernelGGL(k_densify_fill, dim3(c3d_cdiv(N, 256)), dim3(256), 0, (hipStream_t)stream, N, p, counts_host[0], counts_host[1], counts_host[2], src, fresh, child_rank);
    C3D_LAUNCH_CHECK();
    return 0;
}

int c3d_gather_rows(int32_t n_arrays, const float* const* src, float* const* dst, const int32_t* row_floats, const int32_t* zero_fresh, const uint32_t* idx,
                    const uint8_t* fresh, int64_t M, c3d_stream_t stream) {
    if (n_arrays <= 0 || M <= 0) return 0;
    if (n_arrays > C3D_GATHER_MAX) { c3d_set_error("c3d_gather_rows: at most %d arrays per call", C3D_GATHER_MAX); return -1; }
    if (!src || !dst || !row_floats || !idx) { c3d_set_error("c3d_gather_rows: NULL pointer"); return -1; }
    GatherDesc d; d.n = n_arrays;
    int maxrow = 1;
    for (int a = 0; a < n_arrays; a++) {
        if (!src[a] || !dst[a] || row_floats[a] <= 0) { c3d_set_error("c3d_gather_rows: array %d: NULL pointer or empty row", a); return -1; }
        d.src[a] = src[a]; d.dst[a] = dst[a]; d.row[a] = row_floats[a]; d.zero[a] = zero_fresh ? zero_fresh[a] : 0;
        if (row_floats[a] > maxrow) maxrow = row_floats[a];
    }
    long long blocks = (M * maxrow + 255) / 256;
    if (blocks > 256 * 16) blocks = 256 * 16;
    hipLaunchKernelGGL(k_gather_rows, dim3((unsigned)blocks, (unsigned)n_arrays), dim3(256), 0, (hipStream_t)stream, d, idx, fresh, (long long)M);
    C3D_LAUNCH_CHECK();
    return 0;
}
}  // extern "C"
