// xcd_atomics.hip -- where do float atomics execute on gfx950, and what do they cost?
//   (a) atomicAdd (agent scope) into one shared buffer                      -- what a CUDA-style scatter-add does
//   (b) workgroup-scope atomic add into a buffer PRIVATE to the issuing XCD   -- stays in that XCD's L2 (if the hardware honours the scope)
// Pattern: every thread adds 1.0f to `touches` pseudo-random slots of a table of `slots` floats (texture-gradient-like scatter).
// Checks the sums of (b) after an 8-way reduction against (a) and against the expected total.
// build: hipcc --offload-arch=gfx950 -O3 -munsafe-fp-atomics -o xcd_atomics xcd_atomics.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>

__device__ __forceinline__ unsigned xcc_id() {
    unsigned v;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
    return v & 15u;
}
__device__ __forceinline__ unsigned hash(unsigned x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }

template <int MODE>
__global__ void __launch_bounds__(256) k_scatter(float* table, int slots, int touches, unsigned* xcc_hist) {
    const unsigned gid = blockIdx.x * blockDim.x + threadIdx.x;
    const unsigned x = xcc_id();
    if (MODE == 1 && threadIdx.x == 0) atomicAdd(&xcc_hist[x], 1u);
    float* base = MODE == 1 ? table + (size_t)x * slots : table;
    // neighbouring threads hit neighbouring slots (as neighbouring pixels hit neighbouring texels), with a random block origin
    const unsigned origin = hash(blockIdx.x) % (unsigned)slots;
    for (int t = 0; t < touches; t++) {
        const unsigned s = (origin + threadIdx.x * 3 + t * 1031) % (unsigned)slots;
        if (MODE == 0) atomicAdd(base + s, 1.0f);
        else __hip_atomic_fetch_add(base + s, 1.0f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
}
__global__ void k_reduce8(const float* priv, float* out, int slots) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= slots) return;
    float s = 0.f;
    for (int x = 0; x < 8; x++) s += priv[(size_t)x * slots + i];
    out[i] = s;
}

int main() {
    const int slots = 3 << 20, touches = 12, blocks = 4096;   // 12 MB table, 4096*256*12 = 12.6 M atomics
    float *shared_t, *priv_t, *red_t; unsigned* hist;
    hipMalloc(&shared_t, sizeof(float) * slots); hipMalloc(&priv_t, sizeof(float) * slots * 8); hipMalloc(&red_t, sizeof(float) * slots); hipMalloc(&hist, 64);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float ms;
    for (int rep = 0; rep < 2; rep++) {
        hipMemset(shared_t, 0, sizeof(float) * slots); hipMemset(priv_t, 0, sizeof(float) * slots * 8); hipMemset(hist, 0, 64);
        hipDeviceSynchronize();
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL(k_scatter<0>, dim3(blocks), dim3(256), 0, 0, shared_t, slots, touches, hist);
        hipEventRecord(e1, 0); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
        if (rep) printf("(a) agent-scope atomicAdd, shared table     : %.3f ms  (%.1f G atomics/s)\n", ms, (double)blocks * 256 * touches / (ms * 1e-3) / 1e9);
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL(k_scatter<1>, dim3(blocks), dim3(256), 0, 0, priv_t, slots, touches, hist);
        hipLaunchKernelGGL(k_reduce8, dim3((slots + 255) / 256), dim3(256), 0, 0, priv_t, red_t, slots);
        hipEventRecord(e1, 0); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
        if (rep) printf("(b) workgroup-scope, XCD-private + reduce    : %.3f ms  (%.1f G atomics/s)\n", ms, (double)blocks * 256 * touches / (ms * 1e-3) / 1e9);
    }
    std::vector<float> a(slots), b(slots); unsigned h[16];
    hipMemcpy(a.data(), shared_t, sizeof(float) * slots, hipMemcpyDeviceToHost); hipMemcpy(b.data(), red_t, sizeof(float) * slots, hipMemcpyDeviceToHost);
    hipMemcpy(h, hist, 64, hipMemcpyDeviceToHost);
    double sa = 0, sb = 0; long long diff = 0;
    for (int i = 0; i < slots; i++) { sa += a[i]; sb += b[i]; diff += a[i] != b[i]; }
    printf("expected total %.0f ; (a) %.0f ; (b) %.0f ; slots that differ: %lld\n", (double)blocks * 256 * touches, sa, sb, diff);
    printf("blocks per XCC id:"); for (int i = 0; i < 16; i++) printf(" %u", h[i]); printf("\n");
    return 0;
}
