// valu_rate.hip — issue cost of the VALU instructions the raster kernels are made of (gfx950).
// 8 waves per SIMD, 16 independent destination registers per instruction kind, 4096 x 16
// instructions per wave; prints SIMD cycles per wave-instruction at the measured clock.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <string.h>

#define REP16(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13) X(14) X(15)
constexpr int kIters = 4096;

template <int KIND>
__global__ __launch_bounds__(256) void k(float *out, float seed) {
    float r[16];
    float2 q[16];
#pragma unroll
    for (int i = 0; i < 16; i++) { r[i] = seed + i + threadIdx.x; q[i] = make_float2(r[i], r[i] + 1.f); }
    float a = seed * 0.5f, b = seed * 0.25f;
    for (int it = 0; it < kIters; it++) {
#define FMA(i) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(r[i]) : "v"(a), "v"(b));
#define PKFMA(i) asm volatile("v_pk_fma_f32 %0, %1, %1, %0" : "+v"(q[i]) : "v"(q[(i + 1) & 15]));
#define PKMUL(i) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(q[i]) : "v"(q[(i + 1) & 15]));
#define EXP(i) asm volatile("v_exp_f32 %0, %0" : "+v"(r[i]));
#define RCP(i) asm volatile("v_rcp_f32 %0, %0" : "+v"(r[i]));
#define MIN(i) asm volatile("v_min_f32 %0, %1, %0" : "+v"(r[i]) : "v"(a));
#define CMP(i) asm volatile("v_cmp_le_f32 vcc, %0, %1" ::"v"(r[i]), "v"(a) : "vcc");
#define CMPS(i) asm volatile("v_cmp_le_f32_e64 s[20:21], %0, %1" ::"v"(r[i]), "v"(a) : "s20", "s21");
#define CND(i) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(r[i]) : "v"(a) : "vcc");
#define CNDS(i) asm volatile("v_cndmask_b32_e64 %0, %0, %1, s[20:21]" : "+v"(r[i]) : "v"(a));
#define MUL(i) asm volatile("v_mul_f32 %0, %1, %0" : "+v"(r[i]) : "v"(a));
#define DPP(i) asm volatile("v_add_f32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(r[i]));
        if (KIND == 0) { REP16(FMA) }
        if (KIND == 1) { REP16(PKFMA) }
        if (KIND == 2) { REP16(PKMUL) }
        if (KIND == 3) { REP16(EXP) }
        if (KIND == 4) { REP16(RCP) }
        if (KIND == 5) { REP16(MIN) }
        if (KIND == 6) { REP16(CMP) }
        if (KIND == 7) { REP16(CMPS) }
        if (KIND == 8) { REP16(CND) }
        if (KIND == 9) { REP16(CNDS) }
        if (KIND == 10) { REP16(MUL) }
        if (KIND == 11) { REP16(DPP) }
    }
    float s = 0;
#pragma unroll
    for (int i = 0; i < 16; i++) s += r[i] + q[i].x + q[i].y;
    if (s == 12345.678f) out[0] = s;
}

template <int KIND>
double run(const char *name, float *d, double ghz) {
    const int blocks = 256 * 8;  // 8 workgroups of 4 waves per CU -> 8 waves per SIMD
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, d, 1.0f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, d, 1.0f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double instr_per_simd = 8.0 * kIters * 16;  // 8 waves on each SIMD
    const double cyc = ms * 1e-3 * ghz * 1e9 / instr_per_simd;
    printf("%-14s %8.3f ms  %6.2f SIMD-cycles per wave64 instruction (at %.2f GHz)\n", name, ms, cyc, ghz);
    return cyc;
}

int main() {
    float *d; hipMalloc(&d, 4);
    int khz = 0; hipDeviceGetAttribute(&khz, hipDeviceAttributeClockRate, 0);
    const double ghz = khz * 1e-6;
    run<0>("v_fma_f32", d, ghz); run<0>("v_fma_f32", d, ghz);
    run<10>("v_mul_f32", d, ghz); run<1>("v_pk_fma_f32", d, ghz); run<2>("v_pk_mul_f32", d, ghz);
    run<3>("v_exp_f32", d, ghz); run<4>("v_rcp_f32", d, ghz); run<5>("v_min_f32", d, ghz);
    run<6>("v_cmp (vcc)", d, ghz); run<7>("v_cmp (sgpr)", d, ghz); run<8>("v_cndmask vcc", d, ghz);
    run<9>("v_cndmask sgpr", d, ghz); run<11>("v_add_f32 dpp", d, ghz);
    return 0;
}
