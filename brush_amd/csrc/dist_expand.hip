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

// One lane per record: gathers the 15 numbers of visible splat c from the dense arrays.
__global__ __launch_bounds__(256) void k_pack_records(const uint32_t *__restrict__ num_visible,
                                                      const uint32_t *__restrict__ global_from_compact, uint32_t n,
                                                      uint32_t ncoef, uint32_t max_rows,
                                                      const float *__restrict__ v_means, const float *__restrict__ v_scales,
                                                      const float *__restrict__ v_quats, const float *__restrict__ v_opac,
                                                      const float *__restrict__ v_sh, float4 *__restrict__ records) {
    const uint32_t V = min(min(*num_visible, n), max_rows);
    for (uint32_t c = blockIdx.x * blockDim.x + threadIdx.x; c < V; c += gridDim.x * blockDim.x) {
        const uint32_t g = global_from_compact[c];
        const float *m = v_means + (size_t)g * 3, *sc = v_scales + (size_t)g * 3, *sh0 = v_sh + (size_t)g * ncoef * 3;
        const float4 q = reinterpret_cast<const float4 *>(v_quats)[g];
        float4 *row = records + (size_t)c * (kRec / 4);
        row[0] = make_float4(__uint_as_float(g), m[0], m[1], m[2]);
        row[1] = make_float4(sc[0], sc[1], sc[2], q.x);
        row[2] = make_float4(q.y, q.z, q.w, v_opac[g]);
        row[3] = make_float4(sh0[0], sh0[1], sh0[2], 1.0f);
    }
}

template <int DEG>
__global__ __launch_bounds__(256) void k_expand_records(const float *__restrict__ records, uint32_t num_records,
                                                        uint32_t rows_per_view, const uint32_t *__restrict__ view_rows,
                                                        const float *__restrict__ campos,
                                                        const float *__restrict__ means, uint32_t n,
                                                        uint32_t skip_view, float *__restrict__ v_means,
                                                        float *__restrict__ v_scales, float *__restrict__ v_quats,
                                                        float *__restrict__ v_opac, float *__restrict__ v_sh) {
    constexpr uint32_t C = (DEG + 1) * (DEG + 1);
    constexpr uint32_t kOut = 3 * C + 11;  // floats added per record
    const uint32_t lane = lane_id();
    const uint32_t waves = gridDim.x * (blockDim.x / kWave);
    for (uint32_t r = blockIdx.x * (blockDim.x / kWave) + threadIdx.x / kWave; r < num_records; r += waves) {
        const uint32_t view = r / rows_per_view;
        if (view == skip_view || (r - view * rows_per_view) >= view_rows[view]) continue;  // wave-uniform
        const float *rec = records + (size_t)r * kRec;
        const uint32_t gid = __float_as_uint(rec[0]);
        if (gid >= n) continue;
        float dir[3] = {means[(size_t)gid * 3] - campos[view * 3], means[(size_t)gid * 3 + 1] - campos[view * 3 + 1],
                        means[(size_t)gid * 3 + 2] - campos[view * 3 + 2]};
        const float len = sqrtf(dir[0] * dir[0] + dir[1] * dir[1] + dir[2] * dir[2]);
        dir[0] /= len;
        dir[1] /= len;
        dir[2] /= len;
        float Y[C];
        sh_basis<C>(DEG, dir, Y);
        // lane j < 3C adds float j of the splat's contiguous SH row, the next 11 lanes the scalars
        for (uint32_t j = lane; j < kOut; j += kWave) {
            if (j < 3 * C) {
                const uint32_t k = j / 3, ch = j - k * 3;
                float yk = Y[0];
#pragma unroll
                for (uint32_t i = 1; i < C; i++) yk = (k == i) ? Y[i] : yk;
                const float rgb = ch == 0 ? rec[12] : (ch == 1 ? rec[13] : rec[14]);
                unsafeAtomicAdd(v_sh + (size_t)gid * 3 * C + j, yk * (1.0f / kShC0) * rgb);  // v_rgb = v_sh0 / Y0
            } else {
                const uint32_t e = j - 3 * C;  // 0..10
                float *dst = e < 3 ? v_means + (size_t)gid * 3 + e
                           : e < 6 ? v_scales + (size_t)gid * 3 + (e - 3)
                           : e < 10 ? v_quats + (size_t)gid * 4 + (e - 6)
                                    : v_opac + gid;
                unsafeAtomicAdd(dst, rec[1 + e]);
            }
        }
    }
}

}  // namespace
}  // namespace brush

using namespace brush;

extern "C" int brush_pack_view_records(const BrushAux *h_aux, uint32_t n, uint32_t sh_degree, const float *v_means,
                                       const float *v_scales, const float *v_quats, const float *v_opac,
                                       const float *v_sh, float *records, uint32_t max_rows, brush_stream_t stream) {
    if (!h_aux || !h_aux->num_visible || !h_aux->global_from_compact_gid || sh_degree > 4) return BRUSH_ERR_INVALID_ARG;
    if (n == 0 || max_rows == 0) return BRUSH_OK;
    if (!v_means || !v_scales || !v_quats || !v_opac || !v_sh || !records) return BRUSH_ERR_INVALID_ARG;
    const uint32_t C = (sh_degree + 1) * (sh_degree + 1);
    const uint32_t rows = min(n, max_rows);
    hipLaunchKernelGGL(k_pack_records, dim3(min(ceil_div(rows, 256u), 2048u)), dim3(256), 0,
                       static_cast<hipStream_t>(stream), h_aux->num_visible, h_aux->global_from_compact_gid, n, C,
                       max_rows, v_means, v_scales, v_quats, v_opac, v_sh, reinterpret_cast<float4 *>(records));
    BRUSH_HIP_CHECK(hipGetLastError());
    return BRUSH_OK;
}

extern "C" int brush_expand_view_records(const float *records, uint32_t num_records, uint32_t rows_per_view,
                                         const uint32_t *view_rows, const float *campos, const float *means,
                                         uint32_t n, uint32_t sh_degree,
                                         uint32_t skip_view, float *v_means, float *v_scales, float *v_quats,
                                         float *v_opac, float *v_sh, brush_stream_t stream) {
    if (sh_degree > 4 || rows_per_view == 0) return BRUSH_ERR_INVALID_ARG;
    if (n == 0) return BRUSH_OK;
    if (!means || !v_means || !v_scales || !v_quats || !v_opac || !v_sh) return BRUSH_ERR_INVALID_ARG;
    if (num_records > 0 && (!records || !campos || !view_rows)) return BRUSH_ERR_INVALID_ARG;
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
    hipLaunchKernelGGL(k_expand_records<D>, grid, block, 0, s, records, num_records, rows_per_view, view_rows,   \
                       campos, means, n, skip_view, v_means, v_scales, v_quats, v_opac, v_sh)
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
