/*
 * c3d_densify.h -- C-ABI of the device-side densify / prune step of the shared-Gaussian training loop (libc3d_hip.so).
 *
 * Replaces, for this path, the clone -> split -> prune sequence the reference runs every `densification_interval` steps:
 *   /root/reference/MVs_Algorithms/GaussianSplatting/main_3DGS_renderer.py:748-781  (densify_and_prune)
 *   /root/reference/MVs_Algorithms/GaussianSplatting/main_3DGS_renderer.py:641-690  (densify_and_split with N = 2, densify_and_clone)
 *   /root/reference/MVs_Algorithms/GaussianSplatting/main_3DGS_renderer.py:558-637  (prune_points / cat_tensors_to_optimizer / densification_postfix)
 * which there are three rounds of boolean-mask indexing and torch.cat over 6 parameters, 12 Adam moments and 4 side arrays (a device -> host
 * synchronisation per mask).  Here: ONE classification pass, prefix sums with the library's single-pass scan, ONE host read of four
 * counts (the new point count is needed to allocate), one pass that writes the source-index list of the result, and ONE gather launch for all arrays.
 *
 * Result order, as the reference's three rounds leave it: surviving points in index order, clones in index order, first children of the split points
 * in index order, second children.  A clone / child that the prune criteria would remove is never created.
 * All pointers are DEVICE pointers to contiguous arrays; calls are asynchronous on `stream`; 0 = success (c3d_last_error() otherwise).
 */
#ifndef C3D_DENSIFY_H
#define C3D_DENSIFY_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif
#ifndef C3D_STREAM_T
#define C3D_STREAM_T
typedef void* c3d_stream_t; /* hipStream_t */
#endif

/* scratch of c3d_densify_plan for N points */
size_t c3d_densify_plan_bytes(int32_t N);

/* Classification + scans.  Per point i:  g = ||grad_accum[i] / denom[i]|| (NaN -> 0, grad_accum [N,1] so g = |.|), s = max_k exp(scaling_raw[i,k]),
 * o = sigmoid(opacity_raw[i]);  hot = g >= max_grad;  clone = hot && s <= dense_extent;  split = hot && s > dense_extent;
 * dead = o < min_opacity || (max_scale > 0 && s_result > max_scale) with s_result = s for survivors and clones, s / 1.6 for children.
 * counts (device, 8 x uint32): {alive survivors, alive clones, alive split PARENTS, all split parents, all clone candidates, 0, 0, 0}.  `plan` keeps what
 * c3d_densify_fill needs. */
int c3d_densify_plan(int32_t N, const float* grad_accum, const float* denom, const float* scaling_raw, const float* opacity_raw, float max_grad,
                     float dense_extent, float min_opacity, float max_scale, void* plan, uint32_t* counts, c3d_stream_t stream);

/* Source-index list of the result (M = counts[0] + counts[1] + 2 counts[2] rows): src[j] = the point row j is copied from, fresh[j] = 1 for clones and
 * children (their Adam moments start at zero), and for the 2 counts[2] children child_rank[c * counts[2] + k] = rank of the parent among ALL split
 * parents (row c * counts[3] + rank of the reference's noise tensor randn(2 counts[3], 3)).  `counts_host` = the counts read back by the caller. */
int c3d_densify_fill(int32_t N, const void* plan, const uint32_t* counts_host, uint32_t* src, uint8_t* fresh, uint32_t* child_rank, c3d_stream_t stream);

/* One launch gathers rows of up to C3D_GATHER_MAX arrays: dst[a][j, :] = zero_fresh[a] && fresh[j] ? 0 : src[a][idx[j], :]  (row_floats[a] floats per row). */
#define C3D_GATHER_MAX 24
int c3d_gather_rows(int32_t n_arrays, const float* const* src, float* const* dst, const int32_t* row_floats, const int32_t* zero_fresh, const uint32_t* idx,
                    const uint8_t* fresh, int64_t M, c3d_stream_t stream);
#ifdef __cplusplus
}
#endif
#endif
