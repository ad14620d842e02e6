// adam.hip -- fused Adam step for the Gaussian parameter tensors.
// Replaces the torch.optim.Adam step the reference takes at MVs_Algorithms/GaussianSplatting/main_3DGS.py:206 over the six
// parameter groups built at main_3DGS_renderer.py:435-453 (eps = 1e-15, no weight decay, no amsgrad).  Pure streaming:
// 16 B read of p,g,m,v and 12 B write of p,m,v per element (28 B/element), 16-B lane-consecutive accesses.
#include "../../include/c3d_optim.h"
#include "c3d_common.h"

// ONE spelling of the update for the float4 body and the tail: a tensor updated as a whole and the same tensor updated slice by slice (ZeRO-1:
// c3d_hip/parallel.py ZeroOneAdam, slices start at multiples of 4 elements) give the same bits for every element.
__device__ __forceinline__ void adam1(float& p, float g, float& m, float& v, float lr_over_bc1, float inv_sqrt_bc2, float b1, float b2, float omb1, float omb2, float eps) {
    m = __builtin_fmaf(b1, m, omb1 * g);
    v = __builtin_fmaf(b2, v, (omb2 * g) * g);
    p = p - (lr_over_bc1 * m) / __builtin_fmaf(sqrtf(v), inv_sqrt_bc2, eps);
}
__global__ void __launch_bounds__(256) k_adam(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
                                               long long n, float lr_over_bc1, float inv_sqrt_bc2, float b1, float b2, float omb1, float omb2, float eps) {
    const long long n4 = n >> 2;
    const long long stride = (long long)gridDim.x * blockDim.x;
    float4* p4 = reinterpret_cast<float4*>(p); const float4* g4 = reinterpret_cast<const float4*>(g);
    float4* m4 = reinterpret_cast<float4*>(m); float4* v4 = reinterpret_cast<float4*>(v);
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
        float4 pp = p4[i], mm = m4[i], vv = v4[i];
        const float4 gg = g4[i];
        adam1(pp.x, gg.x, mm.x, vv.x, lr_over_bc1, inv_sqrt_bc2, b1, b2, omb1, omb2, eps);
        adam1(pp.y, gg.y, mm.y, vv.y, lr_over_bc1, inv_sqrt_bc2, b1, b2, omb1, omb2, eps);
        adam1(pp.z, gg.z, mm.z, vv.z, lr_over_bc1, inv_sqrt_bc2, b1, b2, omb1, omb2, eps);
        adam1(pp.w, gg.w, mm.w, vv.w, lr_over_bc1, inv_sqrt_bc2, b1, b2, omb1, omb2, eps);
        p4[i] = pp; m4[i] = mm; v4[i] = vv;
    }
    // tail (n % 4) by the first lanes of block 0
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
        const long long i = (n4 << 2) + threadIdx.x;
        float pp = p[i], mm = m[i], vv = v[i];
        adam1(pp, g[i], mm, vv, lr_over_bc1, inv_sqrt_bc2, b1, b2, omb1, omb2, eps);
        p[i] = pp; m[i] = mm; v[i] = vv;
    }
}

extern "C" int c3d_adam_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n, double lr, double beta1,
                             double beta2, double eps, int64_t step, c3d_stream_t stream) {
    if (n <= 0) return 0;
    if (!param || !grad || !exp_avg || !exp_avg_sq) { c3d_set_error("c3d_adam_step: NULL pointer"); return -1; }
    if (step < 1) { c3d_set_error("c3d_adam_step: step must be >= 1"); return -1; }
    if (((uintptr_t)param | (uintptr_t)grad | (uintptr_t)exp_avg | (uintptr_t)exp_avg_sq) & 15) { c3d_set_error("c3d_adam_step: pointers must be 16-byte aligned"); return -1; }
    const double bc1 = 1.0 - pow(beta1, (double)step), bc2 = 1.0 - pow(beta2, (double)step);
    hipStream_t s = (hipStream_t)stream;
    C3dProfScope ps(C3D_P_ADAM, s);
    long long blocks = ((n >> 2) + 255) / 256;
    if (blocks > 256 * 8) blocks = 256 * 8;   // 8 workgroups per CU, grid-stride the rest
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(k_adam, dim3((unsigned)blocks), dim3(256), 0, s, param, grad, exp_avg, exp_avg_sq, (long long)n,
                       (float)(lr / bc1), (float)(1.0 / sqrt(bc2)), (float)beta1, (float)beta2, (float)(1.0 - beta1), (float)(1.0 - beta2), (float)eps);
    C3D_LAUNCH_CHECK();
    return 0;
}

// Every tensor of the optimizer in ONE launch (round 3): the reference's six parameter groups were six launches of 7 us each at its default scene size (10 k
// points, bench.py --workload ref-default) -- 4 % of a 1 ms iteration in launches and gaps.  grid.y = tensor; same adam1(), same bits as c3d_adam_step.
struct AdamDev { float* p; const float* g; float* m; float* v; long long n; float lr_over_bc1, inv_sqrt_bc2, b1, b2, omb1, omb2, eps; int pad; };
struct AdamBatch { AdamDev t[C3D_ADAM_MAX_TENSORS]; };
__global__ void __launch_bounds__(256) k_adam_multi(AdamBatch b) {
    const AdamDev& a = b.t[blockIdx.y];
    const long long n4 = a.n >> 2, stride = (long long)gridDim.x * blockDim.x;
    float4* p4 = reinterpret_cast<float4*>(a.p); const float4* g4 = reinterpret_cast<const float4*>(a.g);
    float4* m4 = reinterpret_cast<float4*>(a.m); float4* v4 = reinterpret_cast<float4*>(a.v);
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
        float4 pp = p4[i], mm = m4[i], vv = v4[i];
        const float4 gg = g4[i];
        adam1(pp.x, gg.x, mm.x, vv.x, a.lr_over_bc1, a.inv_sqrt_bc2, a.b1, a.b2, a.omb1, a.omb2, a.eps);
        adam1(pp.y, gg.y, mm.y, vv.y, a.lr_over_bc1, a.inv_sqrt_bc2, a.b1, a.b2, a.omb1, a.omb2, a.eps);
        adam1(pp.z, gg.z, mm.z, vv.z, a.lr_over_bc1, a.inv_sqrt_bc2, a.b1, a.b2, a.omb1, a.omb2, a.eps);
        adam1(pp.w, gg.w, mm.w, vv.w, a.lr_over_bc1, a.inv_sqrt_bc2, a.b1, a.b2, a.omb1, a.omb2, a.eps);
        p4[i] = pp; m4[i] = mm; v4[i] = vv;
    }
    if (blockIdx.x == 0 && threadIdx.x < (a.n & 3)) {
        const long long i = (n4 << 2) + threadIdx.x;
        float pp = a.p[i], mm = a.m[i], vv = a.v[i];
        adam1(pp, a.g[i], mm, vv, a.lr_over_bc1, a.inv_sqrt_bc2, a.b1, a.b2, a.omb1, a.omb2, a.eps);
        a.p[i] = pp; a.m[i] = mm; a.v[i] = vv;
    }
}
extern "C" int c3d_adam_step_multi(const c3d_adam_tensor* tensors, int32_t count, c3d_stream_t stream) {
    if (count <= 0) return 0;
    if (!tensors) { c3d_set_error("c3d_adam_step_multi: NULL pointer"); return -1; }
    hipStream_t s = (hipStream_t)stream;
    C3dProfScope ps(C3D_P_ADAM, s);
    for (int c0 = 0; c0 < count; c0 += C3D_ADAM_MAX_TENSORS) {
        AdamBatch b;
        const int nt = (count - c0) < C3D_ADAM_MAX_TENSORS ? (count - c0) : C3D_ADAM_MAX_TENSORS;
        long long blocks = 1;
        for (int i = 0; i < nt; i++) {
            const c3d_adam_tensor& t = tensors[c0 + i];
            if (t.n < 0 || (t.n > 0 && (!t.param || !t.grad || !t.exp_avg || !t.exp_avg_sq))) { c3d_set_error("c3d_adam_step_multi: NULL pointer in tensor %d", c0 + i); return -1; }
            if (t.step < 1) { c3d_set_error("c3d_adam_step_multi: step must be >= 1"); return -1; }
            if (((uintptr_t)t.param | (uintptr_t)t.grad | (uintptr_t)t.exp_avg | (uintptr_t)t.exp_avg_sq) & 15) { c3d_set_error("c3d_adam_step_multi: pointers must be 16-byte aligned"); return -1; }
            const double bc1 = 1.0 - pow(t.beta1, (double)t.step), bc2 = 1.0 - pow(t.beta2, (double)t.step);
            b.t[i] = AdamDev{t.param, t.grad, t.exp_avg, t.exp_avg_sq, (long long)t.n, (float)(t.lr / bc1), (float)(1.0 / sqrt(bc2)), (float)t.beta1, (float)t.beta2,
                             (float)(1.0 - t.beta1), (float)(1.0 - t.beta2), (float)t.eps, 0};
            const long long nb = ((t.n >> 2) + 255) / 256;
            if (nb > blocks) blocks = nb;
        }
        if (blocks > 256 * 8) blocks = 256 * 8;
        hipLaunchKernelGGL(k_adam_multi, dim3((unsigned)blocks, (unsigned)nt), dim3(256), 0, s, b);
    }
    C3D_LAUNCH_CHECK();
    return 0;
}

// dst = scale * sum over ranks (rank order) of the gathered gradient copies; see include/c3d_optim.h
__global__ void __launch_bounds__(256) k_reduce_ranks(float* __restrict__ dst, const float* __restrict__ src, int world, long long n, float scale) {
    const long long n4 = n >> 2, stride = (long long)gridDim.x * blockDim.x;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
        float4 a = reinterpret_cast<const float4*>(src)[i];
        for (int r = 1; r < world; r++) {
            const float4 b = reinterpret_cast<const float4*>(src + (size_t)r * n)[i];
            a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
        }
        reinterpret_cast<float4*>(dst)[i] = make_float4(a.x * scale, a.y * scale, a.z * scale, a.w * scale);
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
        const long long i = (n4 << 2) + threadIdx.x;
        float a = src[i];
        for (int r = 1; r < world; r++) a += src[(size_t)r * n + i];
        dst[i] = a * scale;
    }
}
extern "C" int c3d_reduce_ranks_f32(float* dst, const float* gathered, int32_t world, int64_t n, float scale, c3d_stream_t stream) {
    if (n <= 0 || world <= 0) return 0;
    if (!dst || !gathered) { c3d_set_error("c3d_reduce_ranks_f32: NULL pointer"); return -1; }
    if ((((uintptr_t)dst | (uintptr_t)gathered) & 15) || (world > 1 && (n & 3))) { c3d_set_error("c3d_reduce_ranks_f32: pointers must be 16-byte aligned and n a multiple of 4"); return -1; }
    hipStream_t s = (hipStream_t)stream;
    C3dProfScope ps(C3D_P_OTHER, s);
    long long blocks = ((n >> 2) + 255) / 256;
    if (blocks > 256 * 8) blocks = 256 * 8;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(k_reduce_ranks, dim3((unsigned)blocks), dim3(256), 0, s, dst, gathered, world, (long long)n, scale);
    C3D_LAUNCH_CHECK();
    return 0;
}
