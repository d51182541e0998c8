/*
 * detmath.h (oracle copy) — deterministic f32 exp / log.
 *
 * TEST INFRASTRUCTURE (see brush_oracle.h).  The reference calls WGSL exp()/log()
 * (project_forward.wgsl:37, project_visible.wgsl:149-151,176, helpers.wgsl:271), whose
 * results are implementation-defined to a few ULP.  To make the *integer* results of the
 * per-splat stages (visible set, tile counts, tile lists) comparable bit-for-bit between this
 * restatement and the GPU path, both sides evaluate exp/log with the same published
 * recipe, written independently on each side from this description:
 *
 *   exp(x):  k = rint(x * log2(e));  r = fma(k, -ln2_hi, x);  r = fma(k, -ln2_lo, r)
 *            (Cody–Waite two-constant reduction, |r| <= ln2/2), then the degree-7 Taylor
 *            polynomial of e^r in Horner form with fma, then ldexp(p, k).
 *   log(x):  frexp to m in [sqrt(1/2), sqrt(2)), f = m-1, s = f/(2+f), z = s*s,
 *            log(m) = 2s + s*z*(2/3 + z*(2/5 + z*(2/7 + z*2/9))), result
 *            = fma(e, ln2_hi, log(m) + e*ln2_lo).
 *
 * Only +, *, /, fma, rint, frexp, ldexp are used, all of which are correctly rounded on
 * x86-64 (SSE2/FMA) and gfx950, so both sides agree bit-for-bit.  Accuracy vs libm is
 * <= 2 ULP (tests/test_oracle_units.py).
 */
#ifndef BRUSH_ORACLE_DETMATH_H
#define BRUSH_ORACLE_DETMATH_H

#include <math.h>

static inline float det_expf(float x) {
    if (x != x) return x;
    if (x > 88.72283f) return INFINITY;
    if (x < -103.97208f) return 0.0f;
    float k = rintf(x * 1.44269504088896341f);
    float r = fmaf(k, -0.693145751953125f, x);       /* ln2_hi: 0x3f317200 */
    r = fmaf(k, -1.42860682030941723e-6f, r);        /* ln2_lo */
    float p = 1.98412698412698413e-4f;               /* 1/5040 */
    p = fmaf(p, r, 1.38888888888888894e-3f);         /* 1/720 */
    p = fmaf(p, r, 8.33333333333333322e-3f);         /* 1/120 */
    p = fmaf(p, r, 4.16666666666666644e-2f);         /* 1/24 */
    p = fmaf(p, r, 1.66666666666666657e-1f);         /* 1/6 */
    p = fmaf(p, r, 0.5f);
    p = fmaf(p, r, 1.0f);
    p = fmaf(p, r, 1.0f);
    return ldexpf(p, (int)k);
}

static inline float det_logf(float x) {
    if (x != x) return x;
    if (x < 0.0f) return NAN;
    if (x == 0.0f) return -INFINITY;
    if (x == INFINITY) return x;
    int e;
    float m = frexpf(x, &e); /* m in [0.5, 1) */
    if (m < 0.707106769084930419921875f) {
        m = m * 2.0f;
        e -= 1;
    }
    float f = m - 1.0f;
    float s = f / (2.0f + f);
    float z = s * s;
    float p = 0.222222222222222222f;                 /* 2/9 */
    p = fmaf(p, z, 0.285714285714285714f);           /* 2/7 */
    p = fmaf(p, z, 0.4f);                            /* 2/5 */
    p = fmaf(p, z, 0.666666666666666667f);           /* 2/3 */
    float lm = fmaf(s * z, p, 2.0f * s);
    float fe = (float)e;
    float lo = fmaf(fe, 1.42860682030941723e-6f, lm);
    return fmaf(fe, 0.693145751953125f, lo);
}

#endif
