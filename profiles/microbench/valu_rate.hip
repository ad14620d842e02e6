// valu_rate.hip -- issue cost of the VALU instructions the compositing kernels are made of, on gfx950 (wave64).
// Every wave runs ITER x 16 independent instances of ONE instruction (inline asm, 16 distinct destination registers, no dependency
// between neighbours); 8 waves per SIMD keep the issue port saturated.  Reported: SIMD cycles per wave-instruction
// (= elapsed shader cycles of a block / instructions issued on one SIMD), from s_memtime deltas of wave 0 of every block.
// build: hipcc --offload-arch=gfx950 -O3 -o valu_rate valu_rate.hip ; run: ./valu_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

#define ITER 2000
#define REP16(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13) X(14) X(15)

template <int OP>
__global__ void __launch_bounds__(256) k_rate(float* out, unsigned long long* cyc, float seed) {
    float r[16];
    typedef float f2 __attribute__((ext_vector_type(2)));
    f2 p[16];
#pragma unroll
    for (int i = 0; i < 16; i++) { r[i] = seed + i * 0.001f + threadIdx.x * 1e-6f; p[i] = f2{r[i], r[i] * 0.5f}; }
    float a = 1.0001f + seed, b = 0.0001f;
    f2 a2 = {a, a}, b2 = {b, b};
    __syncthreads();
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < ITER; it++) {
#define FMA(i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(r[i]) : "v"(a), "v"(b));
#define MUL(i) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(r[i]) : "v"(a));
#define ADD(i) asm volatile("v_add_f32 %0, %0, %1" : "+v"(r[i]) : "v"(b));
#define PKFMA(i) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[i]) : "v"(a2), "v"(b2));
#define PKMUL(i) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[i]) : "v"(a2));
#define PKADD(i) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[i]) : "v"(b2));
#define EXP(i) asm volatile("v_exp_f32 %0, %0" : "+v"(r[i]));
#define RCP(i) asm volatile("v_rcp_f32 %0, %0" : "+v"(r[i]));
#define DPP(i) asm volatile("v_add_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(r[i]));
#define DPPM(i) asm volatile("v_add_f32_dpp %0, %0, %0 row_mirror row_mask:0xf bank_mask:0xf" : "+v"(r[i]));
#define SWAP32(i) asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(r[i]), "+v"(r[(i + 1) & 15]));
#define CNDMASK(i) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(r[i]) : "v"(a));
#define CMP(i) asm volatile("v_cmp_lt_f32 vcc, %0, %1" : : "v"(r[i]), "v"(a) : "vcc");
#define MIN(i) asm volatile("v_min_f32 %0, %0, %1" : "+v"(r[i]) : "v"(a));
#define FMAC_S(i) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(r[i]) : "v"(a), "v"(b));
#define MOV(i) asm volatile("v_mov_b32 %0, %1" : "=v"(r[i]) : "v"(a));
        if (OP == 0) { REP16(FMA) }
        if (OP == 1) { REP16(MUL) }
        if (OP == 2) { REP16(ADD) }
        if (OP == 3) { REP16(PKFMA) }
        if (OP == 4) { REP16(PKMUL) }
        if (OP == 5) { REP16(PKADD) }
        if (OP == 6) { REP16(EXP) }
        if (OP == 7) { REP16(RCP) }
        if (OP == 8) { REP16(DPP) }
        if (OP == 9) { REP16(DPPM) }
        if (OP == 10) { REP16(SWAP32) }
        if (OP == 11) { REP16(CNDMASK) }
        if (OP == 12) { REP16(CMP) }
        if (OP == 13) { REP16(MIN) }
        if (OP == 14) { REP16(FMAC_S) }
        if (OP == 15) { REP16(MOV) }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 16; i++) s += r[i] + p[i].x + p[i].y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int OP>
void run(const char* name, float* out, unsigned long long* cyc, int blocks) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k_rate<OP>, dim3(blocks), dim3(256), 0, 0, out, cyc, 0.5f);
    hipDeviceSynchronize();
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(k_rate<OP>, dim3(blocks), dim3(256), 0, 0, out, cyc, 0.5f);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h(blocks);
    hipMemcpy(h.data(), cyc, sizeof(unsigned long long) * blocks, hipMemcpyDeviceToHost);
    double mean = 0;
    for (auto v : h) mean += (double)v;
    mean /= blocks;
    // 8 blocks per CU resident (256 threads = one wave per SIMD each) -> 8 waves per SIMD share the issue port
    const double instr_per_wave = (double)ITER * 16;
    const double total_wave_instr = instr_per_wave * blocks * 4;
    printf("%-22s wall %.3f ms  -> %.2f G wave-instr/s chip-wide ; counter ticks per wave-instr (8 waves/SIMD): %.2f (x8 waves => %.2f per SIMD slot)\n", name, ms,
           total_wave_instr / (ms * 1e-3) / 1e9, mean / instr_per_wave, mean / instr_per_wave / 8.0);
}

int main() {
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    const int cus = prop.multiProcessorCount, blocks = cus * 8;
    int wall_khz = 0;
    hipDeviceGetAttribute(&wall_khz, hipDeviceAttributeWallClockRate, 0);
    printf("device %s, %d CUs, clockRate %d kHz, wall-clock rate %d kHz; wave-instr/s per SIMD = chip-wide / %d\n", prop.name, cus, prop.clockRate, wall_khz, cus * 4);
    float* out; unsigned long long* cyc;
    hipMalloc(&out, sizeof(float) * blocks * 256);
    hipMalloc(&cyc, sizeof(unsigned long long) * blocks);
    run<0>("v_fma_f32", out, cyc, blocks);
    run<1>("v_mul_f32", out, cyc, blocks);
    run<2>("v_add_f32", out, cyc, blocks);
    run<14>("v_fmac_f32", out, cyc, blocks);
    run<3>("v_pk_fma_f32", out, cyc, blocks);
    run<4>("v_pk_mul_f32", out, cyc, blocks);
    run<5>("v_pk_add_f32", out, cyc, blocks);
    run<6>("v_exp_f32", out, cyc, blocks);
    run<7>("v_rcp_f32", out, cyc, blocks);
    run<8>("v_add_f32_dpp quad", out, cyc, blocks);
    run<9>("v_add_f32_dpp row_mirror", out, cyc, blocks);
    run<10>("v_permlane32_swap", out, cyc, blocks);
    run<11>("v_cndmask_b32", out, cyc, blocks);
    run<12>("v_cmp_lt_f32", out, cyc, blocks);
    run<13>("v_min_f32", out, cyc, blocks);
    run<15>("v_mov_b32", out, cyc, blocks);
    return 0;
}
