// xcd_atomic_min64.hip -- the z-buffer atomic of k_ras_tri (64-bit atomicMin, one per covered pixel, neighbouring lanes -> neighbouring pixels):
//   (a) agent scope on ONE shared buffer (what the kernel does: the atomic executes beyond the XCD-private L2)
//   (b) workgroup-scope encoding (no sc bits: executes in the issuing XCD's L2) on a buffer that only workgroups of ONE XCD touch (buffer = hardware XCC id)
// Question: what would a rasterizer whose views are assigned to XCDs by XCC id gain?  Checks (b) against a CPU min over the same keys.
// build: hipcc --offload-arch=gfx950 -O3 -o xcd_atomic_min64 xcd_atomic_min64.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
__device__ __forceinline__ unsigned xcc_id() { unsigned v; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v)); return v & 15u; }
__device__ __forceinline__ unsigned hash(unsigned x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }
// every thread is a "triangle" covering `px` pixels in a row at a pseudo-random position of a W x H buffer; 4 layers of depth complexity come from 4 x the pixel count of threads
template <int MODE>
__global__ void __launch_bounds__(256) k_plot(unsigned long long* buf, int W, int H, int px, unsigned* hist) {
    const unsigned gid = blockIdx.x * blockDim.x + threadIdx.x;
    const unsigned x = xcc_id();
    if (threadIdx.x == 0) atomicAdd(&hist[x], 1u);
    unsigned long long* base = MODE == 1 ? buf + (size_t)x * W * H : buf;
    // neighbouring triangles are neighbours on screen (mesh order), blocks land anywhere
    const unsigned h = hash(blockIdx.x);
    const unsigned y = (h >> 8) % (unsigned)H, x0 = (h % (unsigned)(W - 256 * px)) + threadIdx.x * px;
    const unsigned long long key = ((unsigned long long)hash(gid * 7u + 1u) << 32) | gid;
    for (int i = 0; i < px; i++) {
        unsigned long long* p = base + (size_t)y * W + x0 + i;
        if (MODE == 0) atomicMin(p, key);
        else __hip_atomic_fetch_min(p, key, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
}
int main() {
    const int W = 1024, H = 1024, px = 2, blocks = 8 * 1024;     // 8 views' worth: 2 M "triangles" x 2 pixels = 4.2 M atomics per view-equivalent
    unsigned long long *shared_b, *priv_b; unsigned* hist;
    hipMalloc(&shared_b, 8ull * W * H); hipMalloc(&priv_b, 8ull * W * H * 8); hipMalloc(&hist, 64);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float ms;
    for (int rep = 0; rep < 3; rep++) {
        hipMemset(shared_b, 0xFF, 8ull * W * H); hipMemset(priv_b, 0xFF, 8ull * W * H * 8); hipMemset(hist, 0, 64);
        hipDeviceSynchronize();
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL(k_plot<0>, dim3(blocks), dim3(256), 0, 0, shared_b, W, H, px, hist);
        hipEventRecord(e1, 0); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
        if (rep) printf("(a) agent-scope atomicMin u64, one buffer        : %.3f ms  (%.1f G atomics/s)\n", ms, (double)blocks * 256 * px / (ms * 1e-3) / 1e9);
        hipMemset(hist, 0, 64);
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL(k_plot<1>, dim3(blocks), dim3(256), 0, 0, priv_b, W, H, px, hist);
        hipEventRecord(e1, 0); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
        if (rep) printf("(b) L2-scope atomicMin u64, buffer of the XCC id : %.3f ms  (%.1f G atomics/s)\n", ms, (double)blocks * 256 * px / (ms * 1e-3) / 1e9);
    }
    // check: min over the 8 private buffers == the shared buffer
    std::vector<unsigned long long> a((size_t)W * H), b((size_t)W * H * 8);
    hipMemcpy(a.data(), shared_b, 8ull * W * H, hipMemcpyDeviceToHost); hipMemcpy(b.data(), priv_b, 8ull * W * H * 8, hipMemcpyDeviceToHost);
    long long diff = 0;
    for (size_t i = 0; i < (size_t)W * H; i++) { unsigned long long m = ~0ull; for (int x = 0; x < 8; x++) m = b[(size_t)x * W * H + i] < m ? b[(size_t)x * W * H + i] : m; diff += m != a[i]; }
    unsigned h[16]; hipMemcpy(h, hist, 64, hipMemcpyDeviceToHost);
    printf("pixels where min over the XCD-private buffers differs from the shared buffer: %lld\n", diff);
    printf("blocks per XCC id:"); for (int i = 0; i < 16; i++) printf(" %u", h[i]); printf("\n");
    return 0;
}
