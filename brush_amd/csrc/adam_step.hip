// adam_step.hip — the optimizer-side entry points that have a fused twin in project_bwd.hip: brush_adam_step (one Adam
// step on the five parameter groups from the dense gradient arrays of brush_render_backward, train.rs:318-359, update
// rule of burn 0.16 Adam::step), brush_normalize_quats (gaussian_splats.rs:174-175) and brush_refine_stats
// (train.rs:284-316).  Their own translation unit, built WITHOUT FMA contraction like project_bwd.hip (a
// `#pragma clang fp contract(off)` inside train_step.hip, which is built with -ffp-contract=fast for the loss kernels, did
// not take): the arithmetic is that of the fused forms (adam_elem4 / adam_stepped, next_quats_fed, the statistics of
// brush_render_backward_adam), expression for expression, so that the separate calls and the fused call give the same
// bits from the same gradients (tests/test_gpu_train.py::test_fused_and_separate_optimizer_paths_give_the_same_bits).
// Roofline: HBM streams (Adam: 28 bytes per parameter).
#include "internal.hpp"

#pragma clang fp contract(off)

namespace brush {
namespace {

struct AdamArgs {
    float lr[5];  // means, log_scales, quats, raw_opac, sh (dc)
    float sh_lerp, beta1, beta2, eps, rbc1, rbc2;  // rbc = 1 / (1 - beta^time), as the fused forms (adam_stepped)
    uint32_t n, ncoef, quat_vjp;
};

template <int VEC>
__global__ __launch_bounds__(256) void k_adam(AdamArgs a, float *__restrict__ means, float *__restrict__ log_scales,
                                              float *__restrict__ quats, float *__restrict__ raw_opac,
                                              float *__restrict__ sh, const float *__restrict__ g_means,
                                              const float *__restrict__ g_scales, const float *__restrict__ g_quats,
                                              const float *__restrict__ g_opac, const float *__restrict__ g_sh,
                                              float *__restrict__ m1, float *__restrict__ m2, size_t total) {
    const size_t n = a.n;
    const size_t e0 = ((size_t)blockIdx.x * 256 + threadIdx.x) * VEC;
    if (e0 >= total) return;
    // segment of the moment arrays [means 3N | log_scales 3N | quats 4N | raw_opac N | sh 3CN]
    float *p;
    const float *grad;
    float lr;
    size_t rel;
    bool is_sh = false, is_quat = false;
    if (e0 < 3 * n) p = means, grad = g_means, rel = e0, lr = a.lr[0];
    else if (e0 < 6 * n) p = log_scales, grad = g_scales, rel = e0 - 3 * n, lr = a.lr[1];
    else if (e0 < 10 * n) p = quats, grad = g_quats, rel = e0 - 6 * n, lr = a.lr[2], is_quat = true;
    else if (e0 < 11 * n) p = raw_opac, grad = g_opac, rel = e0 - 10 * n, lr = a.lr[3];
    else p = sh, grad = g_sh, rel = e0 - 11 * n, lr = a.lr[4], is_sh = true;
    float g[VEC], mo[VEC], ve[VEC], x[VEC];
    typedef float v4f __attribute__((ext_vector_type(4)));  // streaming (non-temporal) 16-byte accesses
    if constexpr (VEC == 4) {
        *reinterpret_cast<v4f *>(g) = __builtin_nontemporal_load(reinterpret_cast<const v4f *>(grad + rel));
        *reinterpret_cast<v4f *>(mo) = __builtin_nontemporal_load(reinterpret_cast<const v4f *>(m1 + e0));
        *reinterpret_cast<v4f *>(ve) = __builtin_nontemporal_load(reinterpret_cast<const v4f *>(m2 + e0));
        *reinterpret_cast<v4f *>(x) = __builtin_nontemporal_load(reinterpret_cast<const v4f *>(p + rel));
    } else {
        g[0] = grad[rel], mo[0] = m1[e0], ve[0] = m2[e0], x[0] = p[rel];
    }
    if (is_quat && a.quat_vjp) {
        // The op was fed rot / |rot| (gaussian_splats.rs:174-175): chain v_q back to the raw rotation,
        // v_rot = v_q / s - rot * (v_q . rot) / s^3 with s = |rot|.
        if constexpr (VEC == 4) {
            const float s2 = x[0] * x[0] + x[1] * x[1] + x[2] * x[2] + x[3] * x[3];
            const float inv_s = 1.0f / sqrtf(s2);
            const float dot = (g[0] * x[0] + g[1] * x[1] + g[2] * x[2] + g[3] * x[3]) * (inv_s * inv_s * inv_s);
#pragma unroll
            for (int i = 0; i < 4; i++) g[i] = g[i] * inv_s - x[i] * dot;
        } else {
            // scalar layout: the thread of component 0 updates the whole quaternion (the other three
            // must not read components a neighbour is rewriting)
            if (rel & 3) return;
            float q[4], vq[4];
#pragma unroll
            for (int i = 0; i < 4; i++) q[i] = p[rel + i], vq[i] = grad[rel + i];
            const float s2 = q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3];
            const float inv_s = 1.0f / sqrtf(s2);
            const float dot = (vq[0] * q[0] + vq[1] * q[1] + vq[2] * q[2] + vq[3] * q[3]) * (inv_s * inv_s * inv_s);
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const float gi = vq[i] * inv_s - q[i] * dot;
                const float mi = m1[e0 + i] * a.beta1 + gi * (1.0f - a.beta1);
                const float vi = m2[e0 + i] * a.beta2 + (gi * gi) * (1.0f - a.beta2);
                m1[e0 + i] = mi, m2[e0 + i] = vi;
                p[rel + i] = adam_stepped(mi, vi, q[i], a.rbc1, a.rbc2, a.eps, lr);
            }
            return;
        }
    }
    const uint32_t row = 3 * a.ncoef;
    uint32_t k = is_sh ? (uint32_t)(rel % row) : 0;
#pragma unroll
    for (int i = 0; i < VEC; i++) {
        mo[i] = mo[i] * a.beta1 + g[i] * (1.0f - a.beta1);
        ve[i] = ve[i] * a.beta2 + (g[i] * g[i]) * (1.0f - a.beta2);
        const float stepped = adam_stepped(mo[i], ve[i], x[i], a.rbc1, a.rbc2, a.eps, lr);
        // higher-order SH: lerp(old, stepped, 1/lr_coeffs_sh_scale), train.rs:336-351
        x[i] = (is_sh && k >= 3) ? x[i] * (1.0f - a.sh_lerp) + stepped * a.sh_lerp : stepped;
        k = k + 1 == row ? 0 : k + 1;
    }
    if constexpr (VEC == 4) {
        __builtin_nontemporal_store(*reinterpret_cast<v4f *>(mo), reinterpret_cast<v4f *>(m1 + e0));
        __builtin_nontemporal_store(*reinterpret_cast<v4f *>(ve), reinterpret_cast<v4f *>(m2 + e0));
        __builtin_nontemporal_store(*reinterpret_cast<v4f *>(x), reinterpret_cast<v4f *>(p + rel));
    } else {
        m1[e0] = mo[0], m2[e0] = ve[0], p[rel] = x[0];
    }
}

// gaussian_splats.rs:174-175: the op is fed rotation / |rotation|.
__global__ __launch_bounds__(256) void k_normalize_quats(const float4 *__restrict__ rot, float4 *__restrict__ out,
                                                         uint32_t n) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float4 q = rot[i];
    const float s = sqrtf(q.x * q.x + q.y * q.y + q.z * q.z + q.w * q.w);
    out[i] = make_float4(q.x / s, q.y / s, q.z / s, q.w / s);
}

__global__ __launch_bounds__(256) void k_refine_stats(const uint32_t *__restrict__ num_visible,
                                                      const uint32_t *__restrict__ global_from_compact,
                                                      const float2 *__restrict__ v_xy, uint32_t n, float half_w,
                                                      float half_h, float *__restrict__ grad_2d_accum,
                                                      float *__restrict__ xy_grad_counts) {
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float2 v = v_xy[i];
    const float vx = v.x * half_w, vy = v.y * half_h;
    grad_2d_accum[i] += sqrtf(vx * vx + vy * vy);
    if (i < min(*num_visible, n)) {
        const uint32_t g = global_from_compact[i];  // distinct per compact id: plain read-modify-write
        if (g < n) xy_grad_counts[g] += 1.0f;
    }
}

}  // namespace
}  // namespace brush

using namespace brush;

extern "C" int brush_adam_step(const BrushAdamConfig *cfg, uint32_t n, uint32_t sh_degree, float *means,
                               float *log_scales, float *quats, float *raw_opac, float *sh, const float *v_means,
                               const float *v_scales, const float *v_quats, const float *v_opac, const float *v_sh,
                               float *moment1, float *moment2, brush_stream_t stream) {
    if (!cfg || sh_degree > 4 || cfg->time == 0) return BRUSH_ERR_INVALID_ARG;
    if (n == 0) return BRUSH_OK;
    if (!means || !log_scales || !quats || !raw_opac || !sh || !v_means || !v_scales || !v_quats || !v_opac || !v_sh ||
        !moment1 || !moment2)
        return BRUSH_ERR_INVALID_ARG;
    const uint32_t C = (sh_degree + 1) * (sh_degree + 1);
    const size_t total = (size_t)n * (11 + 3 * C);
    if (total / 256 > 0x7FFFFFFFull) return BRUSH_ERR_INVALID_ARG;
    AdamArgs a;
    a.lr[0] = cfg->lr_mean, a.lr[1] = cfg->lr_scale, a.lr[2] = cfg->lr_rotation, a.lr[3] = cfg->lr_opac;
    a.lr[4] = cfg->lr_coeffs_dc;
    a.sh_lerp = cfg->sh_rest_lerp, a.beta1 = cfg->beta1, a.beta2 = cfg->beta2, a.eps = cfg->epsilon;
    adam_bias_corrections(cfg->beta1, cfg->beta2, cfg->time, &a.rbc1, &a.rbc2);  // burn Adam: 1 - beta^time
    a.n = n, a.ncoef = C, a.quat_vjp = cfg->rotation_grad_wrt_normalized;
    hipStream_t s = static_cast<hipStream_t>(stream);
    auto aligned = [](const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
    const bool vec = (n % 4 == 0) && aligned(means) && aligned(log_scales) && aligned(quats) && aligned(raw_opac) &&
                     aligned(sh) && aligned(v_means) && aligned(v_scales) && aligned(v_quats) && aligned(v_opac) &&
                     aligned(v_sh) && aligned(moment1) && aligned(moment2);
    if (vec)
        hipLaunchKernelGGL(k_adam<4>, dim3((uint32_t)((total / 4 + 255) / 256)), dim3(256), 0, s, a, means, log_scales,
                           quats, raw_opac, sh, v_means, v_scales, v_quats, v_opac, v_sh, moment1, moment2, total);
    else
        hipLaunchKernelGGL(k_adam<1>, dim3((uint32_t)((total + 255) / 256)), dim3(256), 0, s, a, means, log_scales,
                           quats, raw_opac, sh, v_means, v_scales, v_quats, v_opac, v_sh, moment1, moment2, total);
    BRUSH_HIP_CHECK(hipGetLastError());
    return BRUSH_OK;
}

extern "C" int brush_normalize_quats(const float *rotation, float *normalized, uint32_t n, brush_stream_t stream) {
    if (n == 0) return BRUSH_OK;
    if (!rotation || !normalized || (reinterpret_cast<uintptr_t>(rotation) & 15) || (reinterpret_cast<uintptr_t>(normalized) & 15))
        return BRUSH_ERR_INVALID_ARG;
    hipLaunchKernelGGL(k_normalize_quats, dim3(ceil_div(n, 256u)), dim3(256), 0, static_cast<hipStream_t>(stream),
                       reinterpret_cast<const float4 *>(rotation), reinterpret_cast<float4 *>(normalized), n);
    BRUSH_HIP_CHECK(hipGetLastError());
    return BRUSH_OK;
}

extern "C" int brush_refine_stats(const BrushAux *h_aux, const float *v_xy, uint32_t n, uint32_t w, uint32_t h,
                                  float *grad_2d_accum, float *xy_grad_counts, brush_stream_t stream) {
    if (!h_aux || !h_aux->num_visible || !h_aux->global_from_compact_gid) return BRUSH_ERR_INVALID_ARG;
    if (n == 0) return BRUSH_OK;
    if (!v_xy || !grad_2d_accum || !xy_grad_counts) return BRUSH_ERR_INVALID_ARG;
    hipLaunchKernelGGL(k_refine_stats, dim3(ceil_div(n, 256u)), dim3(256), 0, static_cast<hipStream_t>(stream),
                       h_aux->num_visible, h_aux->global_from_compact_gid, reinterpret_cast<const float2 *>(v_xy), n,
                       (float)w / 2.0f, (float)h / 2.0f, grad_2d_accum, xy_grad_counts);
    BRUSH_HIP_CHECK(hipGetLastError());
    return BRUSH_OK;
}
