// valu_rate.hip -- issue cost of the VALU instructions the compositing kernels are made of, on gfx950 (wave64).
// Every wave runs ITER x 16 independent instances of ONE instruction (inline asm, 16 distinct destination registers, no dependency
// between neighbours); 8 waves per SIMD keep the issue port saturated.  Reported: SIMD cycles per wave-instruction
// (= elapsed shader cycles of a block / instructions issued on one SIMD), from s_memtime deltas of wave 0 of every block.
// build: hipcc --offload-arch=gfx950 -O3 -o valu_rate valu_rate.hip ; run: ./valu_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

#define REP16(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13) X(14) X(15)

template <int OP>
__global__ void __launch_bounds__(256) k_rate(float* out, unsigned long long* cyc, float seed, int ITER, unsigned long long msk_in) {
    float r[16];
    typedef float f2 __attribute__((ext_vector_type(2)));
    f2 p[16];
#pragma unroll
    for (int i = 0; i < 16; i++) { r[i] = seed + i * 0.001f + threadIdx.x * 1e-6f; p[i] = f2{r[i], r[i] * 0.5f}; }
    float a = 1.0001f + seed, b = 0.0001f;
    f2 a2 = {a, a}, b2 = {b, b};
    unsigned long long msk = msk_in, msk2 = 0;
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
#define CNDMASK64(i) asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(r[i]) : "v"(a), "s"(msk));
#define CMPI(i) asm volatile("v_cmp_lt_i32 vcc, %0, %1" : : "v"(r[i]), "v"(a) : "vcc");
#define CMP64(i) asm volatile("v_cmp_lt_f32_e64 %0, %1, %2" : "=s"(msk2) : "v"(r[i]), "v"(a));
#define MAXF(i) asm volatile("v_max_f32 %0, %0, %1" : "+v"(r[i]) : "v"(a));
#define SWAP16(i) asm volatile("v_permlane16_swap_b32 %0, %1" : "+v"(r[i]), "+v"(r[(i + 1) & 15]));
#define MOVDPP(i) asm volatile("v_mov_b32_dpp %0, %1 row_ror:8 row_mask:0xf bank_mask:0xf" : "+v"(r[i]) : "v"(a));
#define ANDOR(i) asm volatile("v_and_or_b32 %0, %0, %1, %2" : "+v"(r[i]) : "v"(a), "v"(b));
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
        if (OP == 16) { REP16(CNDMASK64) }
        if (OP == 17) { REP16(CMPI) }
        if (OP == 18) { REP16(CMP64) }
        if (OP == 19) { REP16(MAXF) }
        if (OP == 20) { REP16(SWAP16) }
        if (OP == 21) { REP16(MOVDPP) }
        if (OP == 22) { REP16(ANDOR) }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 16; i++) s += r[i] + p[i].x + p[i].y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s + (float)(msk2 & 1);
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int OP>
double once(float* out, unsigned long long* cyc, int blocks, int iter) {
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(k_rate<OP>, dim3(blocks), dim3(256), 0, 0, out, cyc, 0.5f, iter, 0x5555555555555555ull);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0, 0);
    hipLaunchKernelGGL(k_rate<OP>, dim3(blocks), dim3(256), 0, 0, out, cyc, 0.5f, iter, 0x5555555555555555ull);
    (void)hipEventRecord(e1, 0);
    (void)hipEventSynchronize(e1);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    return ms;
}
template <int OP>
void run(const char* name, float* out, unsigned long long* cyc, int blocks) {
    const int i0 = 2000, i1 = 12000;
    const double t0 = once<OP>(out, cyc, blocks, i0), t1 = once<OP>(out, cyc, blocks, i1);
    // slope: time of (i1 - i0) * 16 instructions per wave, 8 waves per SIMD, one SIMD issue port
    const double instr_per_simd = (double)(i1 - i0) * 16 * 8;
    const double ns_per_instr = (t1 - t0) * 1e6 / instr_per_simd;
    printf("%-26s %7.3f ms / %7.3f ms  -> %.3f ns per wave-instruction per SIMD = %.2f cycles at 2.4 GHz\n", name, t0, t1, ns_per_instr, ns_per_instr * 2.4);
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
    run<16>("v_cndmask_b32_e64 sgpr", out, cyc, blocks);
    run<17>("v_cmp_lt_i32 vcc", out, cyc, blocks);
    run<18>("v_cmp_lt_f32_e64 sgpr", out, cyc, blocks);
    run<19>("v_max_f32", out, cyc, blocks);
    run<20>("v_permlane16_swap", out, cyc, blocks);
    run<21>("v_mov_b32_dpp row_ror", out, cyc, blocks);
    run<22>("v_and_or_b32", out, cyc, blocks);
    return 0;
}
