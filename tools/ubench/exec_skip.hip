// exec_skip.hip — does gfx950 issue a wave64 VALU instruction faster when a whole 32-lane half (or all but one 16-lane
// row) of EXEC is zero?  If a pass over an all-zero half is skipped, masking the lanes of a half-quadrant a splat cannot
// reach would make the compositing kernels' evaluation cheaper; if not, only wave-uniform branches (what the kernels use)
// remove issue time.  16 independent v_fma_f32 (and v_exp_f32) per iteration, 4 waves per SIMD, under different masks.
#include <hip/hip_runtime.h>
#include <stdio.h>

#define REP16(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13) X(14) X(15)
constexpr int kIters = 4096;

template <int KIND>
__global__ __launch_bounds__(256) void k(float *out, float seed, unsigned long long mask) {
    float r[16];
#pragma unroll
    for (int i = 0; i < 16; i++) r[i] = seed + i + threadIdx.x;
    const float a = seed * 0.5f, b = seed * 0.25f;
    const unsigned lane = threadIdx.x & 63u;
    if ((mask >> lane) & 1ull) {  // EXEC = mask for the whole loop
        for (int it = 0; it < kIters; it++) {
#define FMA(i) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(r[i]) : "v"(a), "v"(b));
#define EXP(i) asm volatile("v_exp_f32 %0, %0" : "+v"(r[i]));
            if (KIND == 0) { REP16(FMA) }
            if (KIND == 1) { REP16(EXP) }
        }
    }
    float s = 0;
#pragma unroll
    for (int i = 0; i < 16; i++) s += r[i];
    if (s == 12345.678f) out[0] = s;
}

template <int KIND>
void run(const char *name, float *d, double ghz, unsigned long long mask) {
    const int wg_per_cu = 4, blocks = 256 * wg_per_cu;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, d, 1.0f, mask);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, d, 1.0f, mask);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double cyc = ms * 1e-3 * ghz * 1e9 / ((double)wg_per_cu * kIters * 16);
    printf("  {\"op\": \"%s\", \"exec\": \"0x%016llx\", \"ms\": %.3f, \"simd_cycles_per_wave_instruction\": %.2f},\n", name, mask, ms, cyc);
}

int main() {
    float *d;
    hipMalloc(&d, 4);
    int khz = 0;
    hipDeviceGetAttribute(&khz, hipDeviceAttributeClockRate, 0);
    const double ghz = khz * 1e-6;
    printf("{\"what\": \"issue cost of a wave64 VALU instruction under partial EXEC masks, 4 waves per SIMD, gfx950\", \"rows\": [\n");
    const unsigned long long masks[] = {~0ull, 0x00000000FFFFFFFFull, 0xFFFFFFFF00000000ull, 0x000000000000FFFFull,
                                        0x0000FFFF0000FFFFull, 0x5555555555555555ull, 0x0000000000000001ull};
    for (auto m : masks) run<0>("v_fma_f32", d, ghz, m);
    for (auto m : masks) run<1>("v_exp_f32", d, ghz, m);
    printf("  {}]}\n");
    return 0;
}
