/*
 * c3d_knn.h -- C-ABI of the nearest-neighbour statistic the Gaussian model is initialised with (libc3d_hip.so).
 *
 * Replaces, for this path, `simple_knn._C.distCUDA2` (un-vendored CUDA wheel, my-reqs / dependencies.txt) as called at
 *   /root/reference/MVs_Algorithms/GaussianSplatting/main_3DGS_renderer.py:408,420   (create_from_pcd: scales = log(sqrt(clamp_min(dist2, 1e-7))))
 * out[i] = mean of the squared distances from point i to its 3 nearest OTHER points (exact; ties and duplicate positions allowed;
 * with fewer than 4 points the missing neighbours count as distance 0 -- only N >= 4 is meaningful, as for the wheel).
 *
 * Method: uniform grid over the bounding box (cell edge chosen for ~2 points per cell), points ordered by cell with the radix sort of the
 * binning stage, one lane per point searching growing cubes of cells until the third-best distance is inside the searched cube.
 * All pointers are DEVICE pointers, contiguous float32 [N,3] / [N]; bbox_lo / bbox_hi are HOST values; asynchronous on `stream`.
 */
#ifndef C3D_KNN_H
#define C3D_KNN_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif
#ifndef C3D_STREAM_T
#define C3D_STREAM_T
typedef void* c3d_stream_t; /* hipStream_t */
#endif
size_t c3d_knn_scratch_bytes(int32_t N);
int c3d_knn3_mean_dist2(const float* points, int32_t N, const float bbox_lo[3], const float bbox_hi[3], void* scratch, float* out,
                        c3d_stream_t stream);
#ifdef __cplusplus
}
#endif
#endif
