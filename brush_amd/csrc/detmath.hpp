// detmath.hpp — deterministic f32 exp/log for the per-splat stages (device side).
//
// The reference evaluates WGSL exp()/log() (project_forward.wgsl:37,
// project_visible.wgsl:149-151,176, helpers.wgsl:271), which are implementation-defined to a
// few ULP.  The per-splat stages decide integer results from them (cull, tile counts), so this
// library fixes the evaluation to a recipe built only from correctly rounded operations:
//   exp: k = rint(x*log2e); Cody-Waite r = x - k*ln2 (hi/lo, fma); degree-7 Taylor (Horner,
//        fma); ldexp.
//   log: frexp to m in [sqrt(.5), sqrt(2)); s = f/(2+f); odd series in s to s^9 (fma);
//        + e*ln2 (hi/lo, fma).
// Max error 2 ULP.  The per-pixel compositing kernels do NOT use this (they use v_exp_f32).
#pragma once
#include <hip/hip_runtime.h>

namespace brush {

__device__ __forceinline__ float det_expf(float x) {
    if (x != x) return x;
    if (x > 88.72283f) return __builtin_inff();
    if (x < -103.97208f) return 0.0f;
    const float k = __builtin_rintf(x * 1.44269504088896341f);
    float r = __builtin_fmaf(k, -0.693145751953125f, x);
    r = __builtin_fmaf(k, -1.42860682030941723e-6f, r);
    float p = 1.98412698412698413e-4f;
    p = __builtin_fmaf(p, r, 1.38888888888888894e-3f);
    p = __builtin_fmaf(p, r, 8.33333333333333322e-3f);
    p = __builtin_fmaf(p, r, 4.16666666666666644e-2f);
    p = __builtin_fmaf(p, r, 1.66666666666666657e-1f);
    p = __builtin_fmaf(p, r, 0.5f);
    p = __builtin_fmaf(p, r, 1.0f);
    p = __builtin_fmaf(p, r, 1.0f);
    return ldexpf(p, (int)k);
}

__device__ __forceinline__ float det_logf(float x) {
    if (x != x) return x;
    if (x < 0.0f) return __builtin_nanf("");
    if (x == 0.0f) return -__builtin_inff();
    if (x == __builtin_inff()) return x;
    int e;
    float m = frexpf(x, &e);
    if (m < 0.707106769084930419921875f) {
        m = m * 2.0f;
        e -= 1;
    }
    const float f = m - 1.0f;
    const float s = f / (2.0f + f);
    const float z = s * s;
    float p = 0.222222222222222222f;
    p = __builtin_fmaf(p, z, 0.285714285714285714f);
    p = __builtin_fmaf(p, z, 0.4f);
    p = __builtin_fmaf(p, z, 0.666666666666666667f);
    const float lm = __builtin_fmaf(s * z, p, 2.0f * s);
    const float fe = (float)e;
    const float lo = __builtin_fmaf(fe, 1.42860682030941723e-6f, lm);
    return __builtin_fmaf(fe, 0.693145751953125f, lo);
}

__device__ __forceinline__ float det_sigmoid(float x) { return 1.0f / (1.0f + det_expf(-x)); }

}  // namespace brush
