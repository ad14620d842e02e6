/*
 * c3d_loss.h -- C-ABI of the fused multi-scale SSIM term of the training loss (libc3d_hip.so).
 *
 * Replaces, for this path, the pytorch_msssim.MS_SSIM(data_range=1, size_average=True, channel=3) call the reference's trainers make
 * every step:
 *   /root/reference/MVs_Algorithms/GaussianSplatting/main_3DGS.py:102,192     loss += lambda_ssim * (1 - ms_ssim_loss(refs, imgs))
 *   /root/reference/MVs_Algorithms/DiffRastMesh/diff_mesh.py:56,123           (same, default weight 0.5)
 * The wheel is not vendored; its published algorithm (restated in torch in comfyui-3d-pack_amd/shared_utils/msssim.py, which is what the
 * kernels are tested against) is: 11-tap Gaussian window (sigma 1.5), "valid" separable filtering of x, y, x^2, y^2, xy per channel,
 *   cs = (2 s_xy + C2) / (s_xx + s_yy + C2),  ssim = (2 mu_x mu_y + C1) / (mu_x^2 + mu_y^2 + C1) * cs,  C1 = 0.01^2, C2 = 0.03^2,
 * five scales (2x2 average pooling with padding = size % 2 between them), per (image, channel):
 *   ms = prod_{l<4} relu(mean cs_l)^w_l * relu(mean ssim_4)^w_4,  w = (0.0448, 0.2856, 0.3001, 0.2363, 0.1333);  result = mean over (image, channel).
 * At 8 x 3 x 1080 x 1920 the torch op chain (50 grouped convolutions + ~100 elementwise launches, forward and backward) takes 65 ms per
 * step on an MI355X -- nine times the whole rasterizer step (profiles/r02b); here it is ~20 launches of LDS-tiled separable filters.
 *
 * One call computes the value AND the gradient w.r.t. y (x, the reference image, is a constant of the training loop):
 *   ms_out[0]  (device)  += mean MS-SSIM                                (optional, may be NULL)
 *   dL_dy                 = or += grad_scale * d(mean MS-SSIM)/dy        ([B,C,H,W])
 * Level-0 input transform (optional, folds the reference's masking and render()'s clamp into the kernels, main_3DGS.py:169-173):
 *   mask != NULL ([B,1,H,W]):  x_eff = x * mask,  y_eff = clamp(y, 0, 1) * mask  when clamp_y != 0 (y * mask otherwise),
 *   and the gradient is chained through both (zero where y was clamped).
 * All pointers are DEVICE pointers to contiguous float32.  min(H, W) must exceed 160 (five scales of an 11-tap window).
 * The call is asynchronous on `stream`, uses no atomics and is bit-reproducible.
 */
#ifndef C3D_LOSS_H
#define C3D_LOSS_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif
#ifndef C3D_STREAM_T
#define C3D_STREAM_T
typedef void* c3d_stream_t; /* hipStream_t */
#endif
size_t c3d_msssim_workspace_bytes(int32_t B, int32_t C, int32_t H, int32_t W);
int c3d_msssim_value_grad(const float* x, const float* y, const float* mask, int32_t clamp_y, int32_t B, int32_t C, int32_t H, int32_t W,
                          float grad_scale, int32_t accumulate, float* dL_dy, float* ms_out, void* workspace, c3d_stream_t stream);
#ifdef __cplusplus
}
#endif
#endif
