// lazy_sh.hpp — deferred Adam of the spherical-harmonics block (BrushLazySh, include/brush_hip.h): the kernel-side view
// of the state and the replay of pending zero-gradient steps.
//
// The reference steps every parameter of every splat each iteration (train.rs:318-359: five Adam groups over whole
// tensors).  At SH degree 3 the coefficients and their two moments are 48 of the 59 floats per splat, i.e. 81 % of the
// 24 bytes per parameter the optimizer moves, although ~90 % of the splats are outside the view, have a zero gradient,
// and their coefficients are read by nothing until the splat is visible again.  A zero-gradient Adam step is a pure
// function of (m, v, x) and the step's constants, so it can be applied LATER, bit for bit: lazy_step_elem() below is
// adam_elem4() / copy_out() of project_bwd.hip with g = 0, and every translation unit that includes this file is built
// with -ffp-contract=off, so a replayed step rounds exactly as the eager one would have.
#pragma once
#include <math.h>

#include "common.hpp"

namespace brush {

struct LazySh {  // by value in kernel arguments; table == nullptr: eager
    const float4 *table;  // row i: (1 / (1 - beta1^t), 1 / (1 - beta2^t), lr_sh, sh_rest_lerp) of optimizer time t = base + 1 + i
    uint32_t base, now;
    uint32_t *sh_time;    // [N]
    float *m1, *m2;       // SH segments of the moment arrays, [N][row_floats]
    float beta1, beta2, eps;
    __host__ __device__ bool on() const { return table != nullptr; }
};

// Host-side check + conversion.  row_floats = 3 (sh_degree + 1)^2 must be a multiple of 4 (degree 1 or 3): every
// 16-byte chunk then lies inside one splat's row.
inline bool make_lazy_sh(const BrushLazySh *l, uint32_t sh_degree, LazySh *out) {
    *out = LazySh{};
    if (!l) return true;
    const uint32_t row = 3u * (sh_degree + 1u) * (sh_degree + 1u);
    auto aligned = [](const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
    if ((row & 3u) != 0u || !l->table || !l->sh_time || !l->sh_moment1 || !l->sh_moment2) return false;
    if (!aligned(l->table) || !aligned(l->sh_moment1) || !aligned(l->sh_moment2)) return false;
    if (l->now < l->base || l->now - l->base > l->capacity) return false;  // every pending time has its table row
    out->table = reinterpret_cast<const float4 *>(l->table);
    out->base = l->base, out->now = l->now;
    out->sh_time = l->sh_time;
    out->m1 = l->sh_moment1, out->m2 = l->sh_moment2;
    out->beta1 = l->beta1, out->beta2 = l->beta2, out->eps = l->epsilon;
    return true;
}

// burn 0.16 Adam::step divides by 1 - beta^time (f32); the kernels multiply by the reciprocals (adam_stepped()).  One
// host function for every optimizer entry point and for the table of deferred steps.
inline void adam_bias_corrections(float beta1, float beta2, uint32_t time, float *rbc1, float *rbc2) {
    *rbc1 = 1.0f / (1.0f - powf(beta1, (float)time));
    *rbc2 = 1.0f / (1.0f - powf(beta2, (float)time));
}

// The update of burn 0.16 Adam::step on one element (moments already advanced): x - lr (m / bc1) / (sqrt(v / bc2) + eps)
// with the two divisions by the step's constants as multiplications by their host-computed reciprocals, v_sqrt_f32 and
// v_rcp_f32 (1 ulp each; WGSL, which the reference's optimizer runs in, specifies its division to 2.5 ulp and inherits
// the square root's accuracy: no refinement sequences).  ~8 instructions instead of ~45: irrelevant where the optimizer
// is an HBM stream, but the replay of deferred steps below is pure arithmetic.  ONE function for the eager fused step and
// for the replay, in translation units built without FMA contraction: a replayed step gives the eager step's bits.
__device__ __forceinline__ float adam_stepped(float m, float v, float x, float rbc1, float rbc2, float eps, float lr) {
#pragma clang fp contract(off)
    const float denom = __builtin_amdgcn_sqrtf(v * rbc2) + eps;
    return x - ((m * rbc1) * __builtin_amdgcn_rcpf(denom)) * lr;
}

// One zero-gradient step on one element, with the constants `c` = (1 / bc1, 1 / bc2, lr, lerp) of its optimizer time;
// `rest`: an SH coefficient >= 1, which takes the lerp of train.rs:336-351.  Expression for expression adam_elem4()
// and the lerp of copy_out() (project_bwd.hip) with g = 0.
__device__ __forceinline__ void lazy_step_elem(const LazySh &z, const float4 c, bool rest, float &m, float &v, float &x) {
#pragma clang fp contract(off)
    const float g = 0.0f;
    m = m * z.beta1 + g * (1.0f - z.beta1);
    v = v * z.beta2 + (g * g) * (1.0f - z.beta2);
    const float st = adam_stepped(m, v, x, c.x, c.y, z.eps, c.z);
    x = rest ? x * (1.0f - c.w) + st * c.w : st;
}

// The pending steps t0+1 .. z.now of one 16-byte chunk; k0 = position of its first float in the splat's row.
__device__ __forceinline__ void lazy_replay4(const LazySh &z, uint32_t t0, uint32_t k0, float4 &m, float4 &v, float4 &x) {
    // (a block older than the table — a caller that did not flush before rebasing — is replayed from the table's first
    // row on: wrong values for that misuse, but never a read in front of the table)
    for (uint32_t t = max(t0, z.base); t < z.now; t++) {
        const float4 c = z.table[t - z.base];  // row of optimizer time t + 1
        lazy_step_elem(z, c, k0 + 0u >= 3u, m.x, v.x, x.x);
        lazy_step_elem(z, c, k0 + 1u >= 3u, m.y, v.y, x.y);
        lazy_step_elem(z, c, k0 + 2u >= 3u, m.z, v.z, x.z);
        lazy_step_elem(z, c, k0 + 3u >= 3u, m.w, v.w, x.w);
    }
}

// A splat's coefficient row as the eager optimizer would hold it now: the stored row with its pending steps replayed
// in registers.  Nothing is written (the forward stays a pure function of its inputs).
template <uint32_t ROW>
__device__ __forceinline__ void lazy_current_row(const LazySh &z, const float *__restrict__ sh, uint32_t g, float *out) {
    static_assert(ROW % 4u == 0u, "whole 16-byte chunks per row");
    const uint32_t t0 = z.sh_time[g];
    const float4 *row = reinterpret_cast<const float4 *>(sh + (size_t)g * ROW);
    float4 x[ROW / 4u];
#pragma unroll
    for (uint32_t j = 0; j < ROW / 4u; j++) x[j] = row[j];
    if (t0 < z.now) {
        const float4 *m1 = reinterpret_cast<const float4 *>(z.m1 + (size_t)g * ROW);
        const float4 *m2 = reinterpret_cast<const float4 *>(z.m2 + (size_t)g * ROW);
#pragma unroll
        for (uint32_t j = 0; j < ROW / 4u; j++) {
            float4 m = m1[j], v = m2[j];
            lazy_replay4(z, t0, j * 4u, m, v, x[j]);
        }
    }
#pragma unroll
    for (uint32_t j = 0; j < ROW / 4u; j++) out[4 * j] = x[j].x, out[4 * j + 1] = x[j].y, out[4 * j + 2] = x[j].z, out[4 * j + 3] = x[j].w;
}

}  // namespace brush
