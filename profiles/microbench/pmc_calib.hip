// pmc_calib.hip -- calibrates rocprofv3's FETCH_SIZE / WRITE_SIZE on gfx950 for the access patterns of the 3DGS compositing kernels.
// MI355X_MICROARCH.md (HBM): FETCH_SIZE reports half the bytes of a wide coalesced streaming read, "other access widths and WRITE_SIZE are uncalibrated:
// calibrate on a known byte count in your own access pattern before trusting an absolute".  Each kernel below moves a KNOWN number of bytes in ONE pattern:
//   k_cal_stream_read    16 B per lane, lane-consecutive                               (the guide's reference pattern)
//   k_cal_gather64       48 B of a 64-byte record per splat id, three lanes per record  (gs_stage_round / k_composite_fwd_w's fetch: random ids)
//   k_cal_stream_write   16 B per lane, lane-consecutive
//   k_cal_record_write48 one 48-byte record per lane at a scattered record index        (the backward pass's per-pair gradient records)
//   k_cal_byte_rw        1 B per lane read + 1 B per lane written, lane-consecutive     (activity / valid bytes)
// The working sets (1 GiB table, 512 MiB streams) exceed the 256 MiB Infinity Cache.  Run under rocprofv3 --kernel-trace --pmc FETCH_SIZE and again with
// --pmc WRITE_SIZE (profiles/r03_run_c.sh); profiles/summarize_calib.py divides the counters by the known bytes -> profiles/r03_pmc_calibration.json.
// build: hipcc --offload-arch=gfx950 -O3 -o pmc_calib pmc_calib.hip
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ void k_cal_stream_read(const float4* __restrict__ src, size_t n4, float* __restrict__ sink) {
    float acc = 0.f;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) { const float4 v = src[i]; acc += v.x + v.y + v.z + v.w; }
    if (acc == 12345.678f) sink[0] = acc;
}
__global__ void k_cal_gather64(const float4* __restrict__ rec, const uint32_t* __restrict__ ids, size_t n, float* __restrict__ sink) {
    float acc = 0.f;
    for (size_t q = (size_t)blockIdx.x * blockDim.x + threadIdx.x; q < 3 * n; q += (size_t)gridDim.x * blockDim.x) {
        const size_t e = q / 3; const int part = (int)(q - 3 * e);
        const float4 v = rec[4 * (size_t)ids[e] + part];
        acc += v.x + v.w;
    }
    if (acc == 12345.678f) sink[0] = acc;
}
__global__ void k_cal_stream_write(float4* __restrict__ dst, size_t n4) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) dst[i] = make_float4(1.f, 2.f, 3.f, (float)i);
}
__global__ void k_cal_record_write48(float4* __restrict__ dst, const uint32_t* __restrict__ idx, size_t n) {
    for (size_t j = (size_t)blockIdx.x * blockDim.x + threadIdx.x; j < n; j += (size_t)gridDim.x * blockDim.x) {
        float4* o = dst + 3 * (size_t)idx[j];
        o[0] = make_float4(1.f, 2.f, 3.f, 4.f); o[1] = make_float4(5.f, 6.f, 7.f, 8.f); o[2] = make_float4(9.f, 0.f, 1.f, (float)j);
    }
}
__global__ void k_cal_byte_rw(const uint8_t* __restrict__ src, uint8_t* __restrict__ dst, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) dst[i] = (uint8_t)(src[i] + 1);
}

int main() {
    const size_t NREC = 16u << 20;             // 16 Mi records of 64 B = 1 GiB
    const size_t NG = 4u << 20;                // gathered / written records per launch
    const size_t NS4 = 32u << 20;              // 32 Mi float4 = 512 MiB streams
    const size_t NB = 256u << 20;              // 256 Mi bytes
    float4 *rec, *str, *dst; uint32_t *ids, *widx; float* sink; uint8_t *b0, *b1;
    CK(hipMalloc(&rec, NREC * 64)); CK(hipMalloc(&str, NS4 * 16)); CK(hipMalloc(&dst, NS4 * 16));
    CK(hipMalloc(&ids, NG * 4)); CK(hipMalloc(&widx, NG * 4)); CK(hipMalloc(&sink, 64)); CK(hipMalloc(&b0, NB)); CK(hipMalloc(&b1, NB));
    CK(hipMemset(rec, 0, NREC * 64)); CK(hipMemset(str, 0, NS4 * 16)); CK(hipMemset(b0, 1, NB));
    uint32_t* h = (uint32_t*)malloc(NG * 4);
    uint64_t s = 88172645463325252ull;
    for (size_t i = 0; i < NG; i++) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; h[i] = (uint32_t)(s % NREC); }      // random ids, as a depth-sorted tile list is
    CK(hipMemcpy(ids, h, NG * 4, hipMemcpyHostToDevice));
    for (size_t i = 0; i < NG; i++) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; h[i] = (uint32_t)(s % (NS4 / 3)); }  // scattered 48-byte record slots
    CK(hipMemcpy(widx, h, NG * 4, hipMemcpyHostToDevice));
    for (int rep = 0; rep < 3; rep++) {
        hipLaunchKernelGGL(k_cal_stream_read, dim3(4096), dim3(256), 0, 0, str, NS4, sink);
        hipLaunchKernelGGL(k_cal_gather64, dim3(4096), dim3(256), 0, 0, rec, ids, NG, sink);
        hipLaunchKernelGGL(k_cal_stream_write, dim3(4096), dim3(256), 0, 0, dst, NS4);
        hipLaunchKernelGGL(k_cal_record_write48, dim3(4096), dim3(256), 0, 0, dst, widx, NG);
        hipLaunchKernelGGL(k_cal_byte_rw, dim3(4096), dim3(256), 0, 0, b0, b1, NB);
        CK(hipDeviceSynchronize());
    }
    printf("{\"k_cal_stream_read\": {\"read\": %zu, \"write\": 0}, \"k_cal_gather64\": {\"read\": %zu, \"read_lines\": %zu, \"write\": 0}, \"k_cal_stream_write\": {\"read\": 0, \"write\": %zu}, "
           "\"k_cal_record_write48\": {\"read\": %zu, \"write\": %zu}, \"k_cal_byte_rw\": {\"read\": %zu, \"write\": %zu}}\n",
           NS4 * 16, NG * 48 + NG * 4, NG * 64 + NG * 4, NS4 * 16, NG * 4, NG * 48, NB, NB);
    return 0;
}
