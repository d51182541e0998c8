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
    typedef float f4v __attribute__((ext_vector_type(4)));
    f4v acc[3] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
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
        // mixes (issue-port sharing between the VALU and the matrix pipe): 96 fma, then nothing / 9 MFMA / 32 DPP adds
        if (KIND >= 12 && KIND <= 14) { REP16(FMA) REP16(FMA) REP16(FMA) REP16(FMA) REP16(FMA) REP16(FMA) }
        if (KIND == 13 || KIND == 15) {
#define MF(i) asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(acc[(i) % 3]) : "v"(r[i]), "v"(b));
            MF(0) MF(1) MF(2) MF(3) MF(4) MF(5) MF(6) MF(7) MF(8)
        }
        if (KIND == 14) { REP16(DPP) REP16(DPP) }
    }
    if (KIND == 13 || KIND == 15) r[0] += acc[0].x + acc[1].y + acc[2].z;
    float s = 0;
#pragma unroll
    for (int i = 0; i < 16; i++) s += r[i] + q[i].x + q[i].y;
    if (s == 12345.678f) out[0] = s;
}

template <int KIND>
double run(const char *name, float *d, double ghz, int wg_per_cu = 8) {
    const int blocks = 256 * wg_per_cu;  // wg_per_cu workgroups of 4 waves per CU -> as many waves per SIMD
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, d, 1.0f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, d, 1.0f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double instr_per_simd = (double)wg_per_cu * kIters * 16;
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
    // per iteration of 16 "slots": cycles x 16 = SIMD cycles per iteration and wave
    printf("mixes at 4 waves per SIMD; SIMD cycles per iteration and wave = printed value x 16\n");
    run<12>("96 fma", d, ghz, 4); run<13>("96 fma + 9 mfma16x16x4", d, ghz, 4); run<14>("96 fma + 32 dpp add", d, ghz, 4);
    run<15>("9 mfma16x16x4", d, ghz, 4);
    return 0;
}
