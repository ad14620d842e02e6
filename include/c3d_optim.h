/*
 * c3d_optim.h -- C-ABI of the fused optimizer step used by the shared-Gaussian training loop (libc3d_hip.so).
 *
 * Replaces, for this path, the torch.optim.Adam step the reference takes at
 *   /root/reference/MVs_Algorithms/GaussianSplatting/main_3DGS.py:206            (optimizer.step())
 * over the parameter groups declared at
 *   /root/reference/MVs_Algorithms/GaussianSplatting/main_3DGS_renderer.py:435-453 (Adam, lr per group, eps = 1e-15).
 * Semantics are torch.optim.Adam's (no weight decay, no amsgrad, bias correction by `step`), one call per tensor:
 *   m = b1 m + (1-b1) g;  v = b2 v + (1-b2) g^2;  p -= (lr / (1-b1^t)) * m / (sqrt(v)/sqrt(1-b2^t) + eps)
 * Hyper-parameters are doubles so that (1 - beta) is formed in double like torch does (beta2 = 0.999: 1e-5 relative
 * difference otherwise).  All pointers are DEVICE pointers to contiguous float32, 16-byte aligned; the call is asynchronous on `stream`.
 */
#ifndef C3D_OPTIM_H
#define C3D_OPTIM_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif
#ifndef C3D_STREAM_T
#define C3D_STREAM_T
typedef void* c3d_stream_t; /* hipStream_t */
#endif
int c3d_adam_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n, double lr, double beta1,
                  double beta2, double eps, int64_t step, c3d_stream_t stream);

/* The same update for `count` tensors in one launch per 16 tensors (round 3; the reference's optimizer has six groups): identical results to `count`
 * calls of c3d_adam_step. */
#define C3D_ADAM_MAX_TENSORS 16
typedef struct c3d_adam_tensor {
    float* param; const float* grad; float* exp_avg; float* exp_avg_sq;
    int64_t n, step;
    double lr, beta1, beta2, eps;
} c3d_adam_tensor;
int c3d_adam_step_multi(const c3d_adam_tensor* tensors, int32_t count, c3d_stream_t stream);

/* Fixed-order sum of the per-rank gradient copies an all-gather delivered (view-parallel training):
 *   dst[i] = scale * (gathered[0*n + i] + gathered[1*n + i] + ... + gathered[(world-1)*n + i]),  ranks added in that order,
 * so that every replica computes the same bits.  One streaming pass (world reads + 1 write per element) instead of world-1
 * read-modify-write sweeps.  dst must not overlap gathered; pointers 16-byte aligned. */
int c3d_reduce_ranks_f32(float* dst, const float* gathered, int32_t world, int64_t n, float scale, c3d_stream_t stream);
#ifdef __cplusplus
}
#endif
#endif
