// dist_expand.hip — multi-GPU gradient exchange: expand compact per-view records into the dense
// gradient arrays (build extension; the reference is single-device).
//
// A view's parameter gradient is non-zero only for its visible splats and its SH row is rank one,
// v_sh[g] = Y(dir_view(g)) (x) v_rgb[g] (gather_grads.wgsl:186-222), so each rank all-gathers
// 64-byte records  [gid | v_means(3) | v_scales(3) | v_quats(4) | v_opac | v_sh[g,0,:](3) | valid]
// of every view (brush_amd/dist.py) and this kernel sums them into the dense arrays:
// one wave64 per record, lane k < C recomputes Y_k(dir) and adds its 3 contiguous SH floats, the
// next 11 lanes add the 11 scalar gradients — every atomic wave-instruction hits one splat's
// contiguous rows.  A rank normally passes its own view index as `skip_view`: its dense arrays
// already hold its own (exact) contribution, so only the other views are added on top and no
// zero-fill is needed; with skip_view = 0xFFFFFFFF the outputs are zero-filled first and every
// view is expanded.  Roofline: HBM / float-atomic rate (MI355X_MICROARCH.md "global float atomics").
#include "internal.hpp"
#include "splat_math.hpp"

namespace brush {
namespace {

constexpr uint32_t kRec = 16;  // floats per record
constexpr float kShC0 = 0.2820947917738781f;

template <int DEG>
__global__ __launch_bounds__(256) void k_expand_records(const float *__restrict__ records, uint32_t num_records,
                                                        uint32_t rows_per_view, const float *__restrict__ campos,
                                                        const float *__restrict__ means, uint32_t n,
                                                        uint32_t skip_view, float *__restrict__ v_means, float *__restrict__ v_scales,
                                                        float *__restrict__ v_quats, float *__restrict__ v_opac,
                                                        float *__restrict__ v_sh) {
    constexpr uint32_t C = (DEG + 1) * (DEG + 1);
    const uint32_t lane = lane_id();
    const uint32_t waves = gridDim.x * (blockDim.x / kWave);
    for (uint32_t r = blockIdx.x * (blockDim.x / kWave) + threadIdx.x / kWave; r < num_records; r += waves) {
        const float *rec = records + (size_t)r * kRec;
        if (!(rec[15] > 0.5f)) continue;  // padding row (wave-uniform)
        const uint32_t gid = __float_as_uint(rec[0]);
        if (gid >= n) continue;
        const uint32_t view = r / rows_per_view;
        if (view == skip_view) continue;
        if (lane < C) {
            float dir[3] = {means[(size_t)gid * 3] - campos[view * 3], means[(size_t)gid * 3 + 1] - campos[view * 3 + 1],
                            means[(size_t)gid * 3 + 2] - campos[view * 3 + 2]};
            const float len = sqrtf(dir[0] * dir[0] + dir[1] * dir[1] + dir[2] * dir[2]);
            dir[0] /= len;
            dir[1] /= len;
            dir[2] /= len;
            float Y[C];
            sh_basis<C>(DEG, dir, Y);
            float yk = Y[0];
#pragma unroll
            for (uint32_t k = 1; k < C; k++) yk = (lane == k) ? Y[k] : yk;
            float *dst = v_sh + ((size_t)gid * C + lane) * 3;
            const float s = yk * (1.0f / kShC0);  // v_rgb = v_sh0 / Y0
            unsafeAtomicAdd(dst + 0, s * rec[12]);
            unsafeAtomicAdd(dst + 1, s * rec[13]);
            unsafeAtomicAdd(dst + 2, s * rec[14]);
        } else if (lane < C + 11) {
            const uint32_t e = lane - C;  // 0..10
            float *dst = e < 3 ? v_means + (size_t)gid * 3 + e
                       : e < 6 ? v_scales + (size_t)gid * 3 + (e - 3)
                       : e < 10 ? v_quats + (size_t)gid * 4 + (e - 6)
                                : v_opac + gid;
            unsafeAtomicAdd(dst, rec[1 + e]);
        }
    }
}

}  // namespace
}  // namespace brush

using namespace brush;

extern "C" int brush_expand_view_records(const float *records, uint32_t num_records, uint32_t rows_per_view,
                                         const float *campos, const float *means, uint32_t n, uint32_t sh_degree,
                                         uint32_t skip_view, float *v_means, float *v_scales, float *v_quats,
                                         float *v_opac, float *v_sh, brush_stream_t stream) {
    if (sh_degree > 4 || rows_per_view == 0) return BRUSH_ERR_INVALID_ARG;
    if (n == 0) return BRUSH_OK;
    if (!means || !v_means || !v_scales || !v_quats || !v_opac || !v_sh) return BRUSH_ERR_INVALID_ARG;
    if (num_records > 0 && (!records || !campos)) return BRUSH_ERR_INVALID_ARG;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const uint32_t C = (sh_degree + 1) * (sh_degree + 1);
    if (skip_view == 0xFFFFFFFFu) {
        BRUSH_HIP_CHECK(hipMemsetAsync(v_means, 0, sizeof(float) * 3 * (size_t)n, s));
        BRUSH_HIP_CHECK(hipMemsetAsync(v_scales, 0, sizeof(float) * 3 * (size_t)n, s));
        BRUSH_HIP_CHECK(hipMemsetAsync(v_quats, 0, sizeof(float) * 4 * (size_t)n, s));
        BRUSH_HIP_CHECK(hipMemsetAsync(v_opac, 0, sizeof(float) * (size_t)n, s));
        BRUSH_HIP_CHECK(hipMemsetAsync(v_sh, 0, sizeof(float) * 3 * C * (size_t)n, s));
    }
    if (num_records == 0) return BRUSH_OK;
    const dim3 grid(min(ceil_div(num_records, 4u), 8192u)), block(256);
#define BRUSH_EXPAND(D)                                                                                          \
    hipLaunchKernelGGL(k_expand_records<D>, grid, block, 0, s, records, num_records, rows_per_view, campos, means, \
                       n, skip_view, v_means, v_scales, v_quats, v_opac, v_sh)
    switch (sh_degree) {
        case 0: BRUSH_EXPAND(0); break;
        case 1: BRUSH_EXPAND(1); break;
        case 2: BRUSH_EXPAND(2); break;
        case 3: BRUSH_EXPAND(3); break;
        default: BRUSH_EXPAND(4); break;
    }
#undef BRUSH_EXPAND
    BRUSH_HIP_CHECK(hipGetLastError());
    return BRUSH_OK;
}
