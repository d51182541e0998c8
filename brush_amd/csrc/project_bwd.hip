// project_bwd.hip — parameter VJPs: SH / opacity / xy gather + projection backward, fused.
//
// Replaces:
//   GatherGrads       crates/brush-render/src/shaders/gather_grads.wgsl:165-232
//   ProjectBackwards  crates/brush-render/src/shaders/project_backwards.wgsl:75-227
//   and the nine zero-fills of crates/brush-render/src/render.rs:505-507,539-547,573-575.
//
// One launch over GLOBAL splat ids.  Lane g looks up its compact id through the inverse map
// the forward produced; visible splats gather their 9 compact-order gradients (36 B) and
// compute the six dense outputs, non-visible splats store zeros.  Every dense gradient element
// is therefore written exactly once, in global order, with no separate memset and no scatter.
// Traffic per splat: 4 B (map) + 40 B params + 36 B compact grads (visible only) read,
// 52 + 12*C B written.  Roofline: HBM.
//
// Store shape: a lane's v_sh row is 12*C contiguous bytes, so per-lane stores would touch 64
// different rows per instruction.  Each wave instead parks its 64 rows in a private LDS buffer
// (odd row stride, so the column writes are bank-conflict free) and copies the block out as contiguous
// 16-byte-per-lane stores: one full KiB per wave-instruction.  v_means / v_scales (12 B rows) go
// through the same buffer; v_quats / v_xy / v_opac are already lane-contiguous.
//
// Compiled with -ffp-contract=off (same expression trees as the forward projection).
#include "internal.hpp"
#include "splat_math.hpp"
#include "trace.hpp"

#pragma clang fp contract(off)

namespace brush {
namespace {

constexpr uint32_t kThreads = 256;
constexpr uint32_t kRecFloats = 16;  // floats per per-view gradient record (k_project_backward_records)

// project_backwards.wgsl:25-57; G(a,b) = WGSL v_R[a][b] = column a, row b.
__device__ __forceinline__ void quat_to_rotmat_vjp(const float q[4], const Mat3 &vR, float o[4]) {
#define G(a, b) (vR.m[b][a])
    const float w = q[0], x = q[1], y = q[2], z = q[3];
    o[0] = 2.0f * ((x * (G(1, 2) - G(2, 1)) + y * (G(2, 0) - G(0, 2))) + z * (G(0, 1) - G(1, 0)));
    o[1] = 2.0f * (((-2.0f * x * (G(1, 1) + G(2, 2)) + y * (G(0, 1) + G(1, 0))) + z * (G(0, 2) + G(2, 0))) +
                   w * (G(1, 2) - G(2, 1)));
    o[2] = 2.0f * (((x * (G(0, 1) + G(1, 0)) - 2.0f * y * (G(0, 0) + G(2, 2))) + z * (G(1, 2) + G(2, 1))) +
                   w * (G(2, 0) - G(0, 2)));
    o[3] = 2.0f * (((x * (G(0, 2) + G(2, 0)) + y * (G(1, 2) + G(2, 1))) - 2.0f * z * (G(0, 0) + G(1, 1))) +
                   w * (G(0, 1) - G(1, 0)));
#undef G
}

// project_backwards.wgsl:59-72
__device__ __forceinline__ void cov2d_to_conic_vjp(const float conic[3], const float v_conic[3], float o[3]) {
    const float X[2][2] = {{conic[0], conic[1]}, {conic[1], conic[2]}};
    const float Gm[2][2] = {{v_conic[0], v_conic[1] / 2.0f}, {v_conic[1] / 2.0f, v_conic[2]}};
    float XG[2][2], S[2][2];
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
        for (int j = 0; j < 2; j++) XG[i][j] = X[i][0] * Gm[0][j] + X[i][1] * Gm[1][j];
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
        for (int j = 0; j < 2; j++) S[i][j] = XG[i][0] * X[0][j] + XG[i][1] * X[1][j];
    o[0] = -S[0][0];
    o[1] = -(S[0][1] + S[1][0]);
    o[2] = -S[1][1];
}

__global__ __launch_bounds__(kThreads) void k_zero_compact_grads(const uint32_t *__restrict__ num_visible,
                                                                 uint32_t n, float4 *__restrict__ v_compact) {
    BRUSH_KTRACE(kTrZeroGrads, 0);
    const uint32_t V = min(*num_visible, n);
    BRUSH_KTRACE_MARK(1, V);
    const uint32_t words = V * (kCompactStride / 4);
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < words; i += gridDim.x * blockDim.x)
        v_compact[i] = make_float4(0.f, 0.f, 0.f, 0.f);
}

// ProjectBackwards for one splat (project_backwards.wgsl:83-226): (v_xy, v_conic) -> v_mean, v_scale (log
// space), v_quat.  Shared by the dense kernel, the per-view record kernel and nothing else.
__device__ __forceinline__ void splat_projection_vjp(const ViewParams &vp, const float mean[3], const float scale[3],
                                                     const float quat[4], const float vxy[2], const float vconic[3],
                                                     float o_mean[3], float o_scale[3], float o_quat[4]) {
    const Mat3 W = view_rot(vp);
    float p_view[3];
    to_view(vp, mean, p_view);
    float vpj[3];
    {  // project_pix_vjp :19-23
        const float rw = 1.0f / (p_view[2] + 1e-6f);
        const float vp0 = vp.focal[0] * vxy[0], vp1 = vp.focal[1] * vxy[1];
        vpj[0] = vp0 * rw;
        vpj[1] = vp1 * rw;
        vpj[2] = -(vp0 * p_view[0] + vp1 * p_view[1]) * rw * rw;
    }
    float vm[3];
#pragma unroll
    for (int i = 0; i < 3; i++) vm[i] = W.m[0][i] * vpj[0] + W.m[1][i] * vpj[1] + W.m[2][i] * vpj[2];

    float cov2d[3], conic[3], v_cov2d[3];
    calc_cov2d(vp, p_view, scale, quat, cov2d);
    cov_to_conic(cov2d, conic);
    cov2d_to_conic_vjp(conic, vconic, v_cov2d);

    const float rz = 1.0f / p_view[2];
    const float rz2 = rz * rz;
    // J from the UNCLAMPED p_view (project_backwards.wgsl:134-138)
    Mat3 J;
    J.m[0][0] = vp.focal[0] * rz; J.m[0][1] = 0.0f; J.m[0][2] = (-vp.focal[0]) * p_view[0] * rz2;
    J.m[1][0] = 0.0f; J.m[1][1] = vp.focal[1] * rz; J.m[1][2] = (-vp.focal[1]) * p_view[1] * rz2;
    J.m[2][0] = 0.0f; J.m[2][1] = 0.0f; J.m[2][2] = 0.0f;
    const Mat3 R = quat_to_rotmat(quat);
    Mat3 S;
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 3; j++) S.m[i][j] = (i == j) ? scale[i] : 0.0f;
    const Mat3 M = mul(R, S);
    const Mat3 V = mul(M, transpose(M));
    Mat3 v_cov;
    v_cov.m[0][0] = v_cov2d[0]; v_cov.m[0][1] = 0.5f * v_cov2d[1]; v_cov.m[0][2] = 0.0f;
    v_cov.m[1][0] = 0.5f * v_cov2d[1]; v_cov.m[1][1] = v_cov2d[2]; v_cov.m[1][2] = 0.0f;
    v_cov.m[2][0] = 0.0f; v_cov.m[2][1] = 0.0f; v_cov.m[2][2] = 0.0f;
    const Mat3 T = mul(J, W);
    const Mat3 Tt = transpose(T);
    const Mat3 Vt = transpose(V);
    const Mat3 v_V = mul(mul(Tt, v_cov), T);
    const Mat3 v_T = add(mul(mul(v_cov, T), Vt), mul(mul(transpose(v_cov), T), V));

    const float c0 = v_V.m[0][0];
    const float c1 = v_V.m[1][0] + v_V.m[0][1];
    const float c2 = v_V.m[2][0] + v_V.m[0][2];
    const float c3 = v_V.m[1][1];
    const float c4 = v_V.m[2][1] + v_V.m[1][2];
    const float c5 = v_V.m[2][2];

    const Mat3 v_J = mul(v_T, transpose(W));
    const float rz3 = rz2 * rz;
    const float vJ02 = v_J.m[0][2], vJ12 = v_J.m[1][2], vJ00 = v_J.m[0][0], vJ11 = v_J.m[1][1];
    float v_t[3];
    v_t[0] = (-vp.focal[0]) * rz2 * vJ02;
    v_t[1] = (-vp.focal[1]) * rz2 * vJ12;
    v_t[2] = (((-vp.focal[0]) * rz2 * vJ00 + 2.0f * vp.focal[0] * p_view[0] * rz3 * vJ02) -
              vp.focal[1] * rz2 * vJ11) +
             2.0f * vp.focal[1] * p_view[1] * rz3 * vJ12;
#pragma unroll
    for (int i = 0; i < 3; i++)
        o_mean[i] = vm[i] + ((v_t[0] * W.m[0][i] + v_t[1] * W.m[1][i]) + v_t[2] * W.m[2][i]);

    Mat3 two_vVs;
    two_vVs.m[0][0] = 2.0f * c0; two_vVs.m[0][1] = 2.0f * (0.5f * c1); two_vVs.m[0][2] = 2.0f * (0.5f * c2);
    two_vVs.m[1][0] = 2.0f * (0.5f * c1); two_vVs.m[1][1] = 2.0f * c3; two_vVs.m[1][2] = 2.0f * (0.5f * c4);
    two_vVs.m[2][0] = 2.0f * (0.5f * c2); two_vVs.m[2][1] = 2.0f * (0.5f * c4); two_vVs.m[2][2] = 2.0f * c5;
    const Mat3 v_M = mul(two_vVs, M);
#pragma unroll
    for (int j = 0; j < 3; j++) {
        const float vs = (R.m[0][j] * v_M.m[0][j] + R.m[1][j] * v_M.m[1][j]) + R.m[2][j] * v_M.m[2][j];
        o_scale[j] = vs * scale[j];  // log-space (:219)
    }
    const Mat3 v_R = mul(v_M, S);
    quat_to_rotmat_vjp(quat, v_R, o_quat);
}

// Deterministic mode (BRUSH_DETERMINISTIC=1): where the compact-order sums of splat c come from.
//   rows     [I][12]   one row per intersection in emission order (grouped by splat), written by the compositing
//                      backward: [9 sums | compact gid | 0 | 0]
//   k_sum_isect_rows   one lane per row, segmented scan inside each 64-row chunk: a splat whose rows lie inside one
//                      chunk gets its sum in v_compact[c]; a splat that crosses chunk borders leaves partial sums
//                      partials[chunk][0] (rows of a splat that began in an earlier chunk) / [1] (rows of a splat
//                      that continues into the next chunk), which its consumer adds in chunk order.
// Same rows, same tree, same order every run: bitwise reproducible, no atomics, no zero-fill.
struct DetSums {
    const uint32_t *cum_tiles_hit;      // [N] inclusive (aux)
    const uint32_t *num_intersections;  // [1]
    const float *partials;              // [ceil(cap / 64)][2][12]; nullptr = atomic mode
    uint32_t cap;
};

__device__ __forceinline__ void load_compact_sums(const float *__restrict__ v_compact, const DetSums &det, uint32_t c,
                                                  float4 &r0, float4 &r1, float4 &r2) {
    const float4 *row = reinterpret_cast<const float4 *>(v_compact) + (size_t)c * (kCompactStride / 4);
    if (!det.partials) {
        r0 = row[0], r1 = row[1], r2 = row[2];
        return;
    }
    const uint32_t I = min(*det.num_intersections, det.cap);
    const uint32_t u0 = c ? min(det.cum_tiles_hit[c - 1], I) : 0u, u1 = min(det.cum_tiles_hit[c], I);
    r0 = r1 = r2 = make_float4(0.f, 0.f, 0.f, 0.f);
    if (u1 <= u0) return;  // no intersection survived (exact tile test / capacity): zero gradient
    const uint32_t k0 = u0 / kWave, k1 = (u1 - 1u) / kWave;
    if (k0 == k1) {
        r0 = row[0], r1 = row[1], r2 = row[2];
        return;
    }
    for (uint32_t k = k0; k <= k1; k++) {  // chunk order
        const float4 *p = reinterpret_cast<const float4 *>(det.partials) + ((size_t)k * 2 + (k == k0 ? 1 : 0)) * kCompactVec;
        const float4 a = p[0], b = p[1], d = p[2];
        r0.x += a.x, r0.y += a.y, r0.z += a.z, r0.w += a.w;
        r1.x += b.x, r1.y += b.y, r1.z += b.z, r1.w += b.w;
        r2.x += d.x;
    }
}

__global__ __launch_bounds__(kThreads) void k_sum_isect_rows(const float4 *__restrict__ rows,
                                                             const uint32_t *__restrict__ num_intersections,
                                                             const uint32_t *__restrict__ cum_tiles_hit, uint32_t cap,
                                                             float4 *__restrict__ v_compact,
                                                             float4 *__restrict__ partials) {
    const uint32_t I = min(*num_intersections, cap);
    const uint32_t lane = threadIdx.x & (kWave - 1);
    const uint32_t waves = gridDim.x * (kThreads / kWave);
    for (uint32_t k = blockIdx.x * (kThreads / kWave) + threadIdx.x / kWave; (uint64_t)k * kWave < I; k += waves) {
        const uint32_t base = k * kWave, u = base + lane;
        const bool valid = u < I;
        float v[12];
        uint32_t c = 0, first = lane, last = lane;
        bool starts_here = true, ends_here = true;
#pragma unroll
        for (uint32_t i = 0; i < 12; i++) v[i] = 0.f;
        if (valid) {
            const float4 a = rows[(size_t)u * kCompactVec], b = rows[(size_t)u * kCompactVec + 1], d = rows[(size_t)u * kCompactVec + 2];
            v[0] = a.x, v[1] = a.y, v[2] = a.z, v[3] = a.w, v[4] = b.x, v[5] = b.y, v[6] = b.z, v[7] = b.w, v[8] = d.x;
            c = __float_as_uint(d.y);
            const uint32_t u0 = c ? min(cum_tiles_hit[c - 1], I) : 0u, u1 = min(cum_tiles_hit[c], I);
            starts_here = u0 >= base;
            ends_here = u1 <= base + kWave;
            first = max(u0, base) - base;
            last = min(u1, base + kWave) - 1u - base;
        }
        // inclusive scan inside the segment [first, last] (lanes of one splat are contiguous)
#pragma unroll
        for (uint32_t dist = 1; dist < kWave; dist <<= 1) {
#pragma unroll
            for (uint32_t i = 0; i < 9; i++) {
                const float up = __shfl_up(v[i], dist, 64);
                if (lane >= first + dist) v[i] += up;
            }
        }
        if (valid && lane == last) {
            float4 *dst = (starts_here && ends_here) ? v_compact + (size_t)c * kCompactVec
                                                     : partials + ((size_t)k * 2 + (starts_here ? 1 : 0)) * kCompactVec;
            dst[0] = make_float4(v[0], v[1], v[2], v[3]);
            dst[1] = make_float4(v[4], v[5], v[6], v[7]);
            dst[2] = make_float4(v[8], 0.f, 0.f, 0.f);
        }
    }
}

typedef float v4f __attribute__((ext_vector_type(4)));
// Streaming 16-byte accesses: data that is touched once per step and is far larger than the caches.
__device__ __forceinline__ float4 nt_load4(const float *p) {
    const v4f v = __builtin_nontemporal_load(reinterpret_cast<const v4f *>(p));
    return make_float4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ void nt_store4(float *p, float4 v) {
    const v4f nv = {v.x, v.y, v.z, v.w};
    __builtin_nontemporal_store(nv, reinterpret_cast<v4f *>(p));
}

// One Adam update of burn 0.16 `Adam::step` (see train_step.hip:k_adam) on element `e` of the moment
// arrays; returns the stepped parameter value.
__device__ __forceinline__ float adam_elem(const AdamFuse &a, size_t e, float g, float x, float lr) {
    const float m = a.m1[e] * a.beta1 + g * (1.0f - a.beta1);
    const float v = a.m2[e] * a.beta2 + (g * g) * (1.0f - a.beta2);
    a.m1[e] = m, a.m2[e] = v;
    return adam_stepped(m, v, x, a.rbc1, a.rbc2, a.eps, lr);
}
__device__ __forceinline__ float4 adam_elem4(const AdamFuse &a, size_t e, float4 g, float4 x, float4 mo, float4 vo,
                                             float lr) {
    float4 m, v, r;
#define BRUSH_ADAM_C(c)                                                        \
    m.c = mo.c * a.beta1 + g.c * (1.0f - a.beta1);                             \
    v.c = vo.c * a.beta2 + (g.c * g.c) * (1.0f - a.beta2);                     \
    r.c = adam_stepped(m.c, v.c, x.c, a.rbc1, a.rbc2, a.eps, lr);
    BRUSH_ADAM_C(x) BRUSH_ADAM_C(y) BRUSH_ADAM_C(z) BRUSH_ADAM_C(w)
#undef BRUSH_ADAM_C
    nt_store4(a.m1 + e, m);
    nt_store4(a.m2 + e, v);
    return r;
}
__device__ __forceinline__ float4 adam_elem4(const AdamFuse &a, size_t e, float4 g, float4 x, float lr) {
    return adam_elem4(a, e, g, x, nt_load4(a.m1 + e), nt_load4(a.m2 + e), lr);
}

// Last phase of the backward for the 64 splats [g0, g0+64) of one wave: the lane that owns splat g0+lane holds its
// parameter gradients; they are either stored (dense arrays, every element written once, coalesced through the
// per-wave LDS staging rows) or, ADAM, sent straight through the optimizer update of their parameter.
// `stage`: kStageFloats of LDS private to the wave.
template <int DEG, bool ADAM, bool ROWS_READY>
__device__ __forceinline__ void store_gradients_or_step(
    const AdamFuse &af, uint32_t n, uint32_t g0, uint32_t lane, float *stage,
    const uint32_t *row_t0 /* deferred SH (ROWS_READY form): per row of the wave, the time its block is current for */,
    const float o_mean[3],
    const float o_scale[3], const float o_quat[4], float o_opac, const float o_xy[2], float stat_norm,
    float stat_count, const float *Y, const float vcol[3], float *__restrict__ v_means, float *__restrict__ v_xy,
    float *__restrict__ v_scales, float *__restrict__ v_quats, float *__restrict__ v_sh, float *__restrict__ v_opac) {
    constexpr uint32_t ncoef = (DEG + 1) * (DEG + 1);
    constexpr uint32_t kRow = ncoef * 3;     // floats per v_sh row
    constexpr uint32_t kRowPad = kRow | 1u;  // odd LDS row stride: conflict-free column access
    const uint32_t g_own = g0 + lane;
    const bool in_range = g_own < n;
    const uint32_t rows = min(kWave, n - g0);  // rows this wave owns (64 except at the tail)
    const size_t nn = n;
    if (in_range) {
        if (v_xy) reinterpret_cast<float2 *>(v_xy)[g_own] = make_float2(o_xy[0], o_xy[1]);
        if (!ADAM) {
            reinterpret_cast<float4 *>(v_quats)[g_own] = make_float4(o_quat[0], o_quat[1], o_quat[2], o_quat[3]);
            v_opac[g_own] = o_opac;
        } else {
            // rotation: the op was fed rot/|rot| (gaussian_splats.rs:174-175); chain v_q to the raw parameter
            float4 r = reinterpret_cast<const float4 *>(af.rotation)[g_own];
            float4 gq = make_float4(o_quat[0], o_quat[1], o_quat[2], o_quat[3]);
            if (af.quat_vjp) {
                const float s2 = r.x * r.x + r.y * r.y + r.z * r.z + r.w * r.w;
                const float inv_s = 1.0f / sqrtf(s2);
                const float dot = (gq.x * r.x + gq.y * r.y + gq.z * r.z + gq.w * r.w) * (inv_s * inv_s * inv_s);
                gq = make_float4(gq.x * inv_s - r.x * dot, gq.y * inv_s - r.y * dot, gq.z * inv_s - r.z * dot,
                                 gq.w * inv_s - r.w * dot);
            }
            const size_t e = 6 * nn + (size_t)g_own * 4;
            if (af.vec_ok) {
                r = adam_elem4(af, e, gq, r, af.lr[2]);
            } else {
                r.x = adam_elem(af, e + 0, gq.x, r.x, af.lr[2]);
                r.y = adam_elem(af, e + 1, gq.y, r.y, af.lr[2]);
                r.z = adam_elem(af, e + 2, gq.z, r.z, af.lr[2]);
                r.w = adam_elem(af, e + 3, gq.w, r.w, af.lr[2]);
            }
            reinterpret_cast<float4 *>(af.rotation)[g_own] = r;
            if (af.norm_rot_out) {  // what the next forward will be fed (gaussian_splats.rs:174-175)
                const float s = sqrtf(r.x * r.x + r.y * r.y + r.z * r.z + r.w * r.w);
                reinterpret_cast<float4 *>(af.norm_rot_out)[g_own] = make_float4(r.x / s, r.y / s, r.z / s, r.w / s);
            }
            if (af.grad_2d_accum) {  // train.rs:284-316
                af.grad_2d_accum[g_own] += stat_norm * af.stat_scale;
                if (stat_count != 0.0f) af.xy_grad_counts[g_own] += stat_count;
            }
            af.raw_opac[g_own] = adam_elem(af, 10 * nn + g_own, o_opac, af.raw_opac[g_own], af.lr[3]);
        }
    }

    // Copies `rows` rows of ROWF floats (row r at stage[r*STRIDE]) to dst, contiguous across lanes.
    // ADAM: `dst` is the parameter array, `seg` the segment's offset in the moment arrays; the staged
    // gradient updates the parameter in place (SH coefficients >= 1 with the lerp of train.rs:336-351).
    auto copy_out = [&](float *dst, uint32_t rowf, uint32_t stride, size_t seg, float lr, bool is_sh) {
        const uint32_t total = rows * rowf;  // floats; dst is 16-B aligned when g0*rowf % 4 == 0
        if constexpr (ADAM && ROWS_READY) {
            if (is_sh && af.lazy.on()) {
                // Deferred Adam of the SH block (lazy_sh.hpp) in the data-parallel reduction: the blocks of splats NO view
                // saw are left alone, their step stays pending; a seen splat's block first replays what is pending, then
                // takes this step.  Rows are whole 16-byte chunks (make_lazy_sh).
                for (uint32_t j = lane * 4; j < total; j += kWave * 4) {
                    const uint32_t r = j / rowf, k0 = j - r * rowf;
                    const uint32_t t0 = row_t0[r];
                    if (t0 == kInvalid) continue;
                    float4 x = *reinterpret_cast<const float4 *>(dst + j);
                    float4 mo = *reinterpret_cast<const float4 *>(af.m1 + seg + j);
                    float4 vo = *reinterpret_cast<const float4 *>(af.m2 + seg + j);
                    lazy_replay4(af.lazy, t0, k0, mo, vo, x);
                    float4 v;
                    float *e = reinterpret_cast<float *>(&v);
#pragma unroll
                    for (uint32_t i = 0; i < 4; i++) e[i] = stage[r * stride + k0 + i];
                    float4 st = adam_elem4(af, seg + j, v, x, mo, vo, lr);
                    st.x = k0 + 0 >= 3 ? x.x * (1.0f - af.sh_lerp) + st.x * af.sh_lerp : st.x;
                    st.y = k0 + 1 >= 3 ? x.y * (1.0f - af.sh_lerp) + st.y * af.sh_lerp : st.y;
                    st.z = k0 + 2 >= 3 ? x.z * (1.0f - af.sh_lerp) + st.z * af.sh_lerp : st.z;
                    st.w = k0 + 3 >= 3 ? x.w * (1.0f - af.sh_lerp) + st.w * af.sh_lerp : st.w;
                    *reinterpret_cast<float4 *>(dst + j) = st;
                }
                return;
            }
        }
        auto one = [&](uint32_t f) {
            const float gv = stage[(f / rowf) * stride + (f % rowf)];
            if (!ADAM) {
                dst[f] = gv;
            } else {
                const float x = dst[f];
                const float st = adam_elem(af, seg + f, gv, x, lr);
                dst[f] = (is_sh && (f % rowf) >= 3) ? x * (1.0f - af.sh_lerp) + st * af.sh_lerp : st;
            }
        };
        if (((rowf & 3u) == 0 || rows == kWave) && (!ADAM || af.vec_ok)) {
            // float4 path: rowf*64 is a multiple of 4 and the wave's base offset is 16-B aligned.  ADAM: the three
            // streams of kUnroll chunks are requested before the first one is used — 12 KiB in flight per wave
            // instead of 3 (the kernel runs three waves per SIMD and a request takes ~2 us under load: Little's law asks for
            // ~10 MB in flight on the chip at 5 TB/s, one chunk at a time gave 9).
            constexpr uint32_t kUnroll = 4;
            auto staged4 = [&](uint32_t j) {
                float4 v;
                float *e = reinterpret_cast<float *>(&v);
#pragma unroll
                for (uint32_t i = 0; i < 4; i++) {
                    const uint32_t f = j + i;
                    e[i] = stage[(f / rowf) * stride + (f % rowf)];
                }
                return v;
            };
            for (uint32_t j0 = lane * 4; j0 < total; j0 += kWave * 4 * kUnroll) {
                float4 x[kUnroll], mo[kUnroll], vo[kUnroll];
                if (ADAM) {
#pragma unroll
                    for (uint32_t u = 0; u < kUnroll; u++) {
                        const uint32_t j = j0 + u * kWave * 4;
                        if (j + 4 <= total) {
                            x[u] = nt_load4(dst + j);
                            mo[u] = nt_load4(af.m1 + seg + j);
                            vo[u] = nt_load4(af.m2 + seg + j);
                        }
                    }
                }
#pragma unroll
                for (uint32_t u = 0; u < kUnroll; u++) {
                    const uint32_t j = j0 + u * kWave * 4;
                    if (j + 4 <= total) {
                        const float4 v = staged4(j);
                        if (!ADAM) {
                            nt_store4(dst + j, v);  // write-once stream
                        } else {
                            float4 st = adam_elem4(af, seg + j, v, x[u], mo[u], vo[u], lr);
                            if (is_sh) {
                                const uint32_t k0 = j % rowf;  // position in the SH row; rows are rowf floats
                                const float4 xo = x[u];
                                st.x = (k0 + 0) % rowf >= 3 ? xo.x * (1.0f - af.sh_lerp) + st.x * af.sh_lerp : st.x;
                                st.y = (k0 + 1) % rowf >= 3 ? xo.y * (1.0f - af.sh_lerp) + st.y * af.sh_lerp : st.y;
                                st.z = (k0 + 2) % rowf >= 3 ? xo.z * (1.0f - af.sh_lerp) + st.z * af.sh_lerp : st.z;
                                st.w = (k0 + 3) % rowf >= 3 ? xo.w * (1.0f - af.sh_lerp) + st.w * af.sh_lerp : st.w;
                            }
                            nt_store4(dst + j, st);
                        }
                    } else if (j < total) {
                        for (uint32_t f = j; f < total; f++) one(f);
                    }
                }
            }
        } else {
            for (uint32_t f = lane; f < total; f += kWave) one(f);
        }
    };

    // v_sh: row = Y[k] * v_rgb (ROWS_READY: the caller has already summed the rows of several views in `stage`)
    {
        if (!ROWS_READY) {
            float *row = stage + lane * kRowPad;
#pragma unroll
            for (uint32_t k = 0; k < ncoef; k++) {
                row[k * 3 + 0] = Y[k] * vcol[0];
                row[k * 3 + 1] = Y[k] * vcol[1];
                row[k * 3 + 2] = Y[k] * vcol[2];
            }
        }
        __builtin_amdgcn_wave_barrier();
        copy_out((ADAM ? af.sh : v_sh) + (size_t)g0 * kRow, kRow, kRowPad, 11 * nn + (size_t)g0 * kRow, af.lr[4], true);
        __builtin_amdgcn_wave_barrier();
    }
    // v_means, v_scales: 3 floats per row
    {
        stage[lane * 4 + 0] = o_mean[0];
        stage[lane * 4 + 1] = o_mean[1];
        stage[lane * 4 + 2] = o_mean[2];
        stage[256 + lane * 4 + 0] = o_scale[0];
        stage[256 + lane * 4 + 1] = o_scale[1];
        stage[256 + lane * 4 + 2] = o_scale[2];
        __builtin_amdgcn_wave_barrier();
        copy_out((ADAM ? af.means : v_means) + (size_t)g0 * 3, 3, 4, (size_t)g0 * 3, af.lr[0], false);
        stage += 256;
        copy_out((ADAM ? af.log_scales : v_scales) + (size_t)g0 * 3, 3, 4, 3 * nn + (size_t)g0 * 3, af.lr[1], false);
    }
}

// ---- dense gradients (brush_render_backward): the zeros ------------------------------------------------------------
// ~90 % of the splats are not visible from the view, so most of the 52 + 12C bytes per splat the backward writes are
// zeros.  The lanes that own the addresses store them straight from registers: consecutive lanes, consecutive 16-byte
// words of the wave's contiguous regions, the rows of visible splats skipped by their bit in the wave's visibility mask
// `vis`; the visible splats' rows are written by the lanes that computed them (k_project_backward).  No LDS staging, no
// transposes (the staged form spent 56 % of its LDS cycles in bank conflicts): 65 -> 49 us at 1 M splats.
// The v_sh rows are whole cache lines (192 B at degree 3), so their zeros are streaming stores; the small arrays share
// lines between neighbouring splats, visible or not, and use ordinary stores, which the L2 merges into full lines (a
// streaming store of part of a line costs a whole line at the memory: 92 us).  For the same reason the visible rows are
// written by the workgroup that owns their neighbours: a variant with separate workgroups walking the visible splats in
// depth order wrote the same bytes 35 % slower at 21 M splats (1.73 vs 1.28 ms), the partial lines no longer meeting
// in the L2.
template <int DEG>
__device__ __forceinline__ void zero_invisible_rows(uint32_t n, uint32_t g0, uint32_t lane, uint64_t vis,
                                                    float *__restrict__ v_means, float *__restrict__ v_xy,
                                                    float *__restrict__ v_scales, float *__restrict__ v_quats,
                                                    float *__restrict__ v_sh, float *__restrict__ v_opac) {
    constexpr uint32_t kRow = (DEG + 1) * (DEG + 1) * 3;  // floats per v_sh row
    const uint32_t rows = min(kWave, n - g0);
    const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
    float *sh = v_sh + (size_t)g0 * kRow;
    if constexpr (kRow % 16 == 0) {  // rows of whole 64-byte lines
        constexpr uint32_t kPerRow = kRow / 4;
        const uint32_t total = rows * kPerRow;
#pragma unroll
        for (uint32_t it = 0; it < kPerRow; it++) {
            const uint32_t q = it * kWave + lane;
            if (q < total && !((vis >> (q / kPerRow)) & 1ull)) nt_store4(sh + (size_t)q * 4, z4);
        }
    } else {
        const uint32_t total = rows * kRow;
#pragma unroll
        for (uint32_t it = 0; it < kRow; it++) {
            const uint32_t f = it * kWave + lane;
            if (f < total && !((vis >> (f / kRow)) & 1ull)) sh[f] = 0.0f;
        }
    }
    if (lane < rows && !((vis >> lane) & 1ull)) {
        const size_t g = (size_t)g0 + lane;
        if (v_xy) reinterpret_cast<float2 *>(v_xy)[g] = make_float2(0.f, 0.f);
        reinterpret_cast<float4 *>(v_quats)[g] = z4;
        v_opac[g] = 0.0f;
    }
    // v_means / v_scales: 3 floats per splat; 16-byte word `lane` of the wave's region covers floats 4 lane .. 4 lane + 3,
    // i.e. rows (4 lane) / 3 and (4 lane + 3) / 3
    if (lane < 48u) {
        const uint32_t f0 = lane * 4u, ra = f0 / 3u, rb = (f0 + 3u) / 3u;
        const bool a_vis = (vis >> ra) & 1ull, b_vis = (vis >> rb) & 1ull;
        float *m = v_means + (size_t)g0 * 3 + f0, *sc = v_scales + (size_t)g0 * 3 + f0;
        if (rb < rows && !a_vis && !b_vis) {
            *reinterpret_cast<float4 *>(m) = z4;
            *reinterpret_cast<float4 *>(sc) = z4;
        } else {
#pragma unroll
            for (uint32_t i = 0; i < 4; i++) {
                const uint32_t r = (f0 + i) / 3u;
                if (r < rows && !((vis >> r) & 1ull)) m[i] = 0.0f, sc[i] = 0.0f;
            }
        }
    }
}

// GatherGrads + ProjectBackwards of one visible splat `g` from its compact-order sums (r0, r1, r2): the parameter
// gradients and the factors of its v_sh row (Y[k] * vcol).  Shared by the dense kernels; same expression trees in both.
template <int DEG>
__device__ __forceinline__ void visible_splat_vjp(const ViewParams &vp, const float *means, const float *log_scales,
                                                  const float *__restrict__ quats, const float *raw_opac, uint32_t g,
                                                  const float4 r0, const float4 r1, const float4 r2, float o_mean[3],
                                                  float o_scale[3], float o_quat[4], float &o_opac, float o_xy[2],
                                                  float vcol[3], float *Y) {
    constexpr uint32_t ncoef = (DEG + 1) * (DEG + 1);
    const float vxy[2] = {r0.x, r0.y};
    const float vconic[3] = {r0.z, r0.w, r1.x};
    vcol[0] = r1.y;
    vcol[1] = r1.z;
    vcol[2] = r1.w;
    const float v_alpha_sum = r2.x;

    const float mean[3] = {means[(size_t)g * 3], means[(size_t)g * 3 + 1], means[(size_t)g * 3 + 2]};
    const float scale[3] = {det_expf(log_scales[(size_t)g * 3]), det_expf(log_scales[(size_t)g * 3 + 1]),
                            det_expf(log_scales[(size_t)g * 3 + 2])};
    const float4 q4 = reinterpret_cast<const float4 *>(quats)[g];
    const float quat[4] = {q4.x, q4.y, q4.z, q4.w};

    // ---- GatherGrads (gather_grads.wgsl:174-231)
    float dir[3];
    view_dir(vp, mean, dir);
    sh_basis<ncoef>(DEG, dir, Y);
    const float sg = det_sigmoid(raw_opac[g]);
    o_opac = v_alpha_sum * (sg * (1.0f - sg));
    o_xy[0] = vxy[0];
    o_xy[1] = vxy[1];

    // ---- ProjectBackwards (project_backwards.wgsl:83-226)
    splat_projection_vjp(vp, mean, scale, quat, vxy, vconic, o_mean, o_scale, o_quat);
}

// The dense-gradient rows of visible splat `g`, written by the lane that computed them.
template <int DEG>
__device__ __forceinline__ void store_visible_rows(uint32_t g, const float o_mean[3], const float o_scale[3],
                                                   const float o_quat[4], float o_opac, const float o_xy[2],
                                                   const float vcol[3], const float *Y, float *__restrict__ v_means,
                                                   float *__restrict__ v_xy, float *__restrict__ v_scales,
                                                   float *__restrict__ v_quats, float *__restrict__ v_sh,
                                                   float *__restrict__ v_opac) {
    constexpr uint32_t kRow = (DEG + 1) * (DEG + 1) * 3;  // floats per v_sh row
    const size_t gg = g;
    if (v_xy) reinterpret_cast<float2 *>(v_xy)[gg] = make_float2(o_xy[0], o_xy[1]);
    reinterpret_cast<float4 *>(v_quats)[gg] = make_float4(o_quat[0], o_quat[1], o_quat[2], o_quat[3]);
    v_opac[gg] = o_opac;
#pragma unroll
    for (int k = 0; k < 3; k++) v_means[gg * 3 + k] = o_mean[k], v_scales[gg * 3 + k] = o_scale[k];
    float *row = v_sh + gg * kRow;  // v_sh row = Y[k] * v_rgb (gather_grads.wgsl:186-222)
    if constexpr (kRow % 4 == 0) {
#pragma unroll
        for (uint32_t j = 0; j < kRow / 4; j++)
            reinterpret_cast<float4 *>(row)[j] =
                make_float4(Y[(4 * j) / 3] * vcol[(4 * j) % 3], Y[(4 * j + 1) / 3] * vcol[(4 * j + 1) % 3],
                            Y[(4 * j + 2) / 3] * vcol[(4 * j + 2) % 3], Y[(4 * j + 3) / 3] * vcol[(4 * j + 3) % 3]);
    } else {
#pragma unroll
        for (uint32_t e = 0; e < kRow; e++) row[e] = Y[e / 3] * vcol[e % 3];
    }
}

// ADAM: instead of storing the dense parameter gradients, every element goes straight through the
// optimizer update of its parameter (brush_render_backward_adam): the 52+12C bytes per splat of
// gradients are never written to nor re-read from HBM.  v_xy is still stored (refinement statistics).
// In this mode `means`/`log_scales`/`raw_opac` alias the parameters being updated: each lane reads
// its own splat before the wave writes the same 64 splats, and no other wave touches them.
// (Measured and rejected, round 3: handing the 13 small-array results of a visible splat back through LDS to the lane
// that owns the splat, so that v_means / v_xy / v_scales / v_quats / v_opac leave as whole cache lines, zeros and values
// together: 1.48 vs 1.39 ms at 21 M splats, 52.2 vs 51.7 us at 1 M.  The extra barrier costs more than the partial
// lines.)
// PREZEROED: the compositing backward in front of this launch has already zeroed the dense arrays in passing (ZeroFill,
// internal.hpp): only the visible splats' rows are written here, by the lanes that compute them.  (Still one lane per
// GLOBAL id: a launch over the visible splats in depth order writes the same rows slower at every size measured — 27.5
// vs 25.9 us at 1 M splats, 0.80 vs 0.57 ms at 21 M — because the partial lines of neighbouring splats no longer meet
// in the L2; profiles/r04_zero_fill_in_passing.json.)
template <int DEG, bool ADAM, bool PREZEROED = false>
__global__ __launch_bounds__(kThreads) void k_project_backward(
    ViewParams vp, const float *means, const float *log_scales, const float *__restrict__ quats,
    const float *raw_opac, const uint32_t *__restrict__ compact_from_global,
    const float *__restrict__ v_compact, float *__restrict__ v_means, float *__restrict__ v_xy,
    float *__restrict__ v_scales, float *__restrict__ v_quats, float *__restrict__ v_sh,
    float *__restrict__ v_opac, AdamFuse af, DetSums det) {
    constexpr uint32_t ncoef = (DEG + 1) * (DEG + 1);
    constexpr uint32_t kRow = ncoef * 3;                 // floats per v_sh row
    constexpr uint32_t kRowPad = kRow | 1u;              // odd LDS row stride: conflict-free column access
    constexpr uint32_t kRes = 16 + ncoef;  // per-splat VJP results: mean3 scale3 quat4 opac xy2 vcol3 | Y[ncoef]
    constexpr uint32_t kStageA = (kWave * kRowPad > 512u ? kWave * kRowPad : 512u);
    constexpr uint32_t kStageFloats = kStageA > kWave * kRes ? kStageA : kWave * kRes;
    __shared__ float stage_all[kThreads / kWave][kStageFloats];
    BRUSH_KTRACE(kTrProjectBwd, 0);
    const uint32_t wv = threadIdx.x / kWave;
    const uint32_t lane = threadIdx.x & (kWave - 1);
    float *stage = stage_all[wv];
    const uint32_t n = vp.total_splats;
    const uint32_t g0 = blockIdx.x * kThreads + wv * kWave;  // first splat of this wave
    const uint32_t g_own = g0 + lane;
    const bool in_range = g_own < n;
    const uint32_t c_own = in_range ? compact_from_global[g_own] : kInvalid;

    // Only ~10 % of the splats are visible, and the VJP below is ~1500 instructions: the visible
    // splats of the block's 256 are compacted (ballot + LDS) so the arithmetic runs on nearly full
    // waves, and the results travel back to the owning lane through LDS (aliasing the store staging).
    __shared__ uint32_t vis_cnt[kThreads / kWave];
    __shared__ uint16_t vis_list[kThreads];
    static_assert(kThreads * kRes <= (kThreads / kWave) * kStageFloats, "result rows must fit the staging buffer");
    float *res = &stage_all[0][0];
    const uint64_t own_vis = __ballot(c_own != kInvalid);
    {
        const uint64_t bal = own_vis;
        if (lane == 0) vis_cnt[wv] = __popcll(bal);
        __syncthreads();
        uint32_t pos = __popcll(bal & lanemask_lt());
        for (uint32_t w2 = 0; w2 < wv; w2++) pos += vis_cnt[w2];
        if (c_own != kInvalid) vis_list[pos] = (uint16_t)threadIdx.x;
        __syncthreads();
    }
    const uint32_t nvis = vis_cnt[0] + vis_cnt[1] + vis_cnt[2] + vis_cnt[3];
    const uint32_t li = threadIdx.x < nvis ? vis_list[threadIdx.x] : 0u;
    const uint32_t g = blockIdx.x * kThreads + li;
    const uint32_t c = threadIdx.x < nvis ? compact_from_global[g] : kInvalid;
    // Dense gradients: the zeros of the invisible splats go out first, so the stores are in flight during the VJP
    if (!ADAM && !PREZEROED && g0 < n) zero_invisible_rows<DEG>(n, g0, lane, own_vis, v_means, v_xy, v_scales, v_quats, v_sh, v_opac);

    float o_mean[3] = {0.f, 0.f, 0.f}, o_scale[3] = {0.f, 0.f, 0.f}, o_quat[4] = {0.f, 0.f, 0.f, 0.f};
    float o_xy[2] = {0.f, 0.f}, o_opac = 0.f;
    float vcol[3] = {0.f, 0.f, 0.f};
    float Y[ncoef];
#pragma unroll
    for (uint32_t k = 0; k < ncoef; k++) Y[k] = 0.f;

    if (c != kInvalid) {
        float4 r0, r1, r2;
        load_compact_sums(v_compact, det, c, r0, r1, r2);
        visible_splat_vjp<DEG>(vp, means, log_scales, quats, raw_opac, g, r0, r1, r2, o_mean, o_scale, o_quat, o_opac, o_xy,
                               vcol, Y);
        // dense gradients: the computing lane writes the visible splat's rows itself (ordinary stores)
        if (!ADAM) store_visible_rows<DEG>(g, o_mean, o_scale, o_quat, o_opac, o_xy, vcol, Y, v_means, v_xy, v_scales, v_quats, v_sh, v_opac);
        if (ADAM) {
            float *r = res + li * kRes;  // hand the results to the lane that owns splat `li`
            r[0] = o_mean[0], r[1] = o_mean[1], r[2] = o_mean[2];
            r[3] = o_scale[0], r[4] = o_scale[1], r[5] = o_scale[2];
            r[6] = o_quat[0], r[7] = o_quat[1], r[8] = o_quat[2], r[9] = o_quat[3];
            r[10] = o_opac, r[11] = o_xy[0], r[12] = o_xy[1];
            r[13] = vcol[0], r[14] = vcol[1], r[15] = vcol[2];
#pragma unroll
            for (uint32_t k = 0; k < ncoef; k++) r[16 + k] = Y[k];
        }
    }
    if (!ADAM) return;  // nothing left to exchange: no barrier below is reached by any wave of the block
    __syncthreads();
    {
        const float *r = res + threadIdx.x * kRes;
        const bool vis = c_own != kInvalid;
        o_mean[0] = vis ? r[0] : 0.f, o_mean[1] = vis ? r[1] : 0.f, o_mean[2] = vis ? r[2] : 0.f;
        o_scale[0] = vis ? r[3] : 0.f, o_scale[1] = vis ? r[4] : 0.f, o_scale[2] = vis ? r[5] : 0.f;
        o_quat[0] = vis ? r[6] : 0.f, o_quat[1] = vis ? r[7] : 0.f, o_quat[2] = vis ? r[8] : 0.f, o_quat[3] = vis ? r[9] : 0.f;
        o_opac = vis ? r[10] : 0.f, o_xy[0] = vis ? r[11] : 0.f, o_xy[1] = vis ? r[12] : 0.f;
        vcol[0] = vis ? r[13] : 0.f, vcol[1] = vis ? r[14] : 0.f, vcol[2] = vis ? r[15] : 0.f;
#pragma unroll
        for (uint32_t k = 0; k < ncoef; k++) Y[k] = vis ? r[16 + k] : 0.f;
    }
    __syncthreads();  // `res` aliases the store staging below
    if (g0 >= n) return;  // wave-uniform; past the last barrier
    float stat_norm = 0.0f;
    if (ADAM && af.grad_2d_accum) {  // train.rs:300-302
        const float vx = o_xy[0] * af.half_w, vy = o_xy[1] * af.half_h;
        stat_norm = sqrtf(vx * vx + vy * vy);
    }
    store_gradients_or_step<DEG, ADAM, false>(af, n, g0, lane, stage, nullptr, o_mean, o_scale, o_quat, o_opac, o_xy, stat_norm,
                                              c_own != kInvalid ? 1.0f : 0.0f, Y, vcol, v_means, v_xy, v_scales, v_quats,
                                              v_sh, v_opac);
}

// ---- fused backward + Adam with the SH block under deferred Adam (BrushAdamConfig::lazy_sh) ---------------------------
// With the SH block out of the stream the all-in-one kernel above is a chain of small dependent memory phases per wave
// (measured: 155 us at 1 M splats for 0.4 GB).  k_project_backward_lazy is built for what is left: a workgroup owns 256
// CONSECUTIVE global ids (a launch over the visible splats in depth order gathers from nine arrays at random pages per
// lane: no faster) and runs three phases:
//   1. its visible splats, compacted to the first lanes (as in k_project_backward): the projection VJP; the results
//      the other phases need stay in LDS (per visible splat: global id, the time its stored SH block is current for,
//      v_rgb, Y; per owned splat: the 11 small-group gradients and the screen-space statistic);
//   2. the visible splats' SH blocks, kChunks consecutive lanes per row: pending zero-gradient steps replayed, this step
//      applied, sh_time advanced (118 MB of read-modify-write at 1 M splats);
//   3. the workgroup's share of the small-group stream: the 704 16-byte chunks of means / log_scales / rotation /
//      raw_opacity of its 256 splats with their moments (285 MB at 1 M splats), three per lane requested together; the
//      rotation chunks (one splat each) also carry the per-splat duties: chain rule through the normalisation,
//      next_quats_fed, the refinement statistics, v_xy.
// Workgroups in different phases overlap (latency / arithmetic against bandwidth), which two launches — a visible-splat
// kernel (53 us) and a plain small-group stream (60 us at 4.7 TB/s) — could not.  Same expressions as
// store_gradients_or_step<ADAM>: the same bits as the all-in-one kernel.  Requires n % 4 == 0 and 16-byte aligned
// arrays (AdamFuse::vec_ok).
template <int DEG>
__global__ __launch_bounds__(kThreads) void k_project_backward_lazy(
    ViewParams vp, const float *means, const float *log_scales, const float *__restrict__ quats, const float *raw_opac,
    const uint32_t *__restrict__ compact_from_global, const float *__restrict__ v_compact, float *__restrict__ v_xy,
    AdamFuse af, DetSums det) {
    constexpr uint32_t ncoef = (DEG + 1) * (DEG + 1), kRow = ncoef * 3, kChunks = kRow / 4;
    static_assert(kRow % 4 == 0, "rows of whole 16-byte chunks");
    constexpr uint32_t kFac = (5 + ncoef) | 1u;
    constexpr uint32_t kSmall = 13;  // mean3 scale3 quat4 opac | statistic | visible flag
    __shared__ float fac[kThreads][kFac];
    __shared__ float small_g[kThreads][kSmall];
    __shared__ uint32_t vis_cnt[kThreads / kWave];
    __shared__ uint16_t vis_list[kThreads];
    const uint32_t wv = threadIdx.x / kWave;
    const uint32_t n = vp.total_splats, b0 = blockIdx.x * kThreads;
    const size_t nn = n;
    const uint32_t g_own = b0 + threadIdx.x;
    const uint32_t c_own = g_own < n ? compact_from_global[g_own] : kInvalid;
#pragma unroll
    for (uint32_t k = 0; k < kSmall; k++) small_g[threadIdx.x][k] = 0.0f;  // a splat the view does not see: zero gradient
    {
        const uint64_t bal = __ballot(c_own != kInvalid);
        if (lane_id() == 0) vis_cnt[wv] = __popcll(bal);
        __syncthreads();
        uint32_t pos = __popcll(bal & lanemask_lt());
        for (uint32_t w2 = 0; w2 < wv; w2++) pos += vis_cnt[w2];
        if (c_own != kInvalid) vis_list[pos] = (uint16_t)threadIdx.x;
        __syncthreads();
    }
    const uint32_t nvis = vis_cnt[0] + vis_cnt[1] + vis_cnt[2] + vis_cnt[3];
    // ---- 1. VJP of the visible splats
    uint32_t g = kInvalid;
    if (threadIdx.x < nvis) {
        const uint32_t li = vis_list[threadIdx.x];
        g = b0 + li;
        const uint32_t c = compact_from_global[g];
        float4 r0, r1, r2;
        load_compact_sums(v_compact, det, c, r0, r1, r2);
        float o_mean[3], o_scale[3], o_quat[4], o_xy[2], o_opac, vcol[3], Y[ncoef];
        visible_splat_vjp<DEG>(vp, means, log_scales, quats, raw_opac, g, r0, r1, r2, o_mean, o_scale, o_quat, o_opac, o_xy,
                               vcol, Y);
        const float vx = o_xy[0] * af.half_w, vy = o_xy[1] * af.half_h;  // train.rs:300-302
        reinterpret_cast<float2 *>(v_xy)[g] = make_float2(o_xy[0], o_xy[1]);
        float *sg = small_g[li];
        sg[0] = o_mean[0], sg[1] = o_mean[1], sg[2] = o_mean[2];
        sg[3] = o_scale[0], sg[4] = o_scale[1], sg[5] = o_scale[2];
        sg[6] = o_quat[0], sg[7] = o_quat[1], sg[8] = o_quat[2], sg[9] = o_quat[3];
        sg[10] = o_opac, sg[11] = sqrtf(vx * vx + vy * vy), sg[12] = 1.0f;
        float *f = fac[threadIdx.x];
        f[0] = __uint_as_float(g), f[1] = __uint_as_float(af.lazy.sh_time[g]);
        f[2] = vcol[0], f[3] = vcol[1], f[4] = vcol[2];
#pragma unroll
        for (uint32_t k = 0; k < ncoef; k++) f[5 + k] = Y[k];
    }
    __syncthreads();
    // ---- 2. the SH blocks: catch up, then this step (gradient row = Y[k] * v_rgb, gather_grads.wgsl:186-222); two
    // chunks per lane are requested together (~25 visible splats x 12 chunks over 256 lanes at 1 M splats)
    const uint32_t total = nvis * kChunks;
    for (uint32_t q0 = threadIdx.x; q0 < total; q0 += 2 * kThreads) {
        float4 x[2], mo[2], vo[2];
        size_t e[2];
        uint32_t r[2], k0[2];
#pragma unroll
        for (uint32_t u = 0; u < 2; u++) {
            const uint32_t q = q0 + u * kThreads;
            if (q < total) {
                r[u] = q / kChunks, k0[u] = (q - r[u] * kChunks) * 4u;
                e[u] = (size_t)__float_as_uint(fac[r[u]][0]) * kRow + k0[u];
                x[u] = *reinterpret_cast<const float4 *>(af.sh + e[u]);
                mo[u] = *reinterpret_cast<const float4 *>(af.m1 + 11 * nn + e[u]);
                vo[u] = *reinterpret_cast<const float4 *>(af.m2 + 11 * nn + e[u]);
            }
        }
#pragma unroll
        for (uint32_t u = 0; u < 2; u++) {
            if (q0 + u * kThreads >= total) continue;
            const float *f = fac[r[u]];
            lazy_replay4(af.lazy, __float_as_uint(f[1]), k0[u], mo[u], vo[u], x[u]);
            const float4 gr = make_float4(f[5 + (k0[u] + 0) / 3] * f[2 + (k0[u] + 0) % 3], f[5 + (k0[u] + 1) / 3] * f[2 + (k0[u] + 1) % 3],
                                          f[5 + (k0[u] + 2) / 3] * f[2 + (k0[u] + 2) % 3], f[5 + (k0[u] + 3) / 3] * f[2 + (k0[u] + 3) % 3]);
            float4 st = adam_elem4(af, 11 * nn + e[u], gr, x[u], mo[u], vo[u], af.lr[4]);
            st.x = k0[u] + 0 >= 3 ? x[u].x * (1.0f - af.sh_lerp) + st.x * af.sh_lerp : st.x;
            st.y = k0[u] + 1 >= 3 ? x[u].y * (1.0f - af.sh_lerp) + st.y * af.sh_lerp : st.y;
            st.z = k0[u] + 2 >= 3 ? x[u].z * (1.0f - af.sh_lerp) + st.z * af.sh_lerp : st.z;
            st.w = k0[u] + 3 >= 3 ? x[u].w * (1.0f - af.sh_lerp) + st.w * af.sh_lerp : st.w;
            *reinterpret_cast<float4 *>(af.sh + e[u]) = st;
        }
    }
    if (g != kInvalid) af.lazy.sh_time[g] = af.lazy.now + 1u;
    // ---- 3. the small groups of the workgroup's splats, as 16-byte chunks in the order of the moment arrays
    const uint32_t nb = min(kThreads, n - b0);                      // splats owned (a multiple of 4)
    const uint32_t c3 = nb * 3u / 4u, cq = nb, co = nb / 4u;         // chunks of means (= log_scales), rotation, raw_opacity
    const uint32_t chunks = 2u * c3 + cq + co;
    constexpr uint32_t kPer = 3;                                     // 704 chunks over 256 lanes
    float4 x[kPer], mo[kPer], vo[kPer];
    float *p[kPer];
    size_t rel[kPer], e0[kPer];
    uint32_t rowf[kPer], field0[kPer];
    float lr[kPer];
#pragma unroll
    for (uint32_t u = 0; u < kPer; u++) {
        const uint32_t q = threadIdx.x + u * kThreads;
        if (q >= chunks) continue;
        if (q < c3) p[u] = af.means, rel[u] = (size_t)b0 * 3 + q * 4u, e0[u] = rel[u], lr[u] = af.lr[0], rowf[u] = 3, field0[u] = 0;
        else if (q < 2u * c3) p[u] = af.log_scales, rel[u] = (size_t)b0 * 3 + (q - c3) * 4u, e0[u] = 3 * nn + rel[u], lr[u] = af.lr[1], rowf[u] = 3, field0[u] = 3;
        else if (q < 2u * c3 + cq) p[u] = af.rotation, rel[u] = (size_t)b0 * 4 + (q - 2u * c3) * 4u, e0[u] = 6 * nn + rel[u], lr[u] = af.lr[2], rowf[u] = 4, field0[u] = 6;
        else p[u] = af.raw_opac, rel[u] = (size_t)b0 + (q - 2u * c3 - cq) * 4u, e0[u] = 10 * nn + rel[u], lr[u] = af.lr[3], rowf[u] = 1, field0[u] = 10;
        x[u] = nt_load4(p[u] + rel[u]);
        mo[u] = nt_load4(af.m1 + e0[u]);
        vo[u] = nt_load4(af.m2 + e0[u]);
    }
#pragma unroll
    for (uint32_t u = 0; u < kPer; u++) {
        const uint32_t q = threadIdx.x + u * kThreads;
        if (q >= chunks) continue;
        float gr[4];
#pragma unroll
        for (uint32_t i = 0; i < 4; i++) {
            const uint32_t gi = (uint32_t)((rel[u] + i) / rowf[u]);
            gr[i] = small_g[gi - b0][field0[u] + (uint32_t)((rel[u] + i) - (size_t)gi * rowf[u])];
        }
        float4 g4 = make_float4(gr[0], gr[1], gr[2], gr[3]);
        if (rowf[u] == 4) {  // one splat per chunk: the per-splat duties ride here
            const uint32_t gs = (uint32_t)(rel[u] / 4), l = gs - b0;
            const bool vis = small_g[l][12] != 0.0f;
            if (af.quat_vjp) {  // the op was fed rot/|rot| (gaussian_splats.rs:174-175); chain v_q to the raw parameter
                const float4 r = x[u];
                const float s2 = r.x * r.x + r.y * r.y + r.z * r.z + r.w * r.w;
                const float inv_s = 1.0f / sqrtf(s2);
                const float dot = (g4.x * r.x + g4.y * r.y + g4.z * r.z + g4.w * r.w) * (inv_s * inv_s * inv_s);
                g4 = make_float4(g4.x * inv_s - r.x * dot, g4.y * inv_s - r.y * dot, g4.z * inv_s - r.z * dot,
                                 g4.w * inv_s - r.w * dot);
            }
            if (af.grad_2d_accum && vis) {  // train.rs:284-316 (a splat the view does not see adds +0: left alone)
                af.grad_2d_accum[gs] += small_g[l][11] * af.stat_scale;
                af.xy_grad_counts[gs] += 1.0f;
            }
            if (!vis) reinterpret_cast<float2 *>(v_xy)[gs] = make_float2(0.f, 0.f);  // (visible: phase 1)
            const float4 st = adam_elem4(af, e0[u], g4, x[u], mo[u], vo[u], lr[u]);
            nt_store4(p[u] + rel[u], st);
            if (af.norm_rot_out) {  // what the next forward will be fed (gaussian_splats.rs:174-175)
                const float sn = sqrtf(st.x * st.x + st.y * st.y + st.z * st.z + st.w * st.w);
                reinterpret_cast<float4 *>(af.norm_rot_out)[gs] = make_float4(st.x / sn, st.y / sn, st.z / sn, st.w / sn);
            }
        } else {
            nt_store4(p[u] + rel[u], adam_elem4(af, e0[u], g4, x[u], mo[u], vo[u], lr[u]));
        }
    }
}

// ---- view-sharded data parallelism: per-view gradient records and their deterministic reduction ------------
//
// A view's parameter gradient is non-zero only for its visible splats and its SH row is rank one,
// v_sh[g] = Y(dir_view(g)) (x) v_rgb[g] (gather_grads.wgsl:186-222): 16 floats per VISIBLE splat describe it,
//   [gid | v_means(3) | v_scales(3) | v_quats(4) | v_opac | v_rgb(3) | |v_xy * (w/2, h/2)|]          (64 bytes)
// k_project_backward_records writes them in compact (depth) order straight from the compositing backward's sums
// (no dense 52+12C bytes/splat arrays at all); the ranks all-gather the records of every view and
// k_reduce_view_records, one lane per GLOBAL splat id, adds the <= W records of its splat in view order 0..W-1.
// No atomics: the sum is the same bit pattern on every rank and from run to run, so replicated parameters stay
// replicated.  The sums go through store_gradients_or_step: dense arrays, or straight into the Adam update.


__global__ __launch_bounds__(kThreads) void k_project_backward_records(
    ViewParams vp, const float *__restrict__ means, const float *__restrict__ log_scales,
    const float *__restrict__ quats, const float *__restrict__ raw_opac, const uint32_t *__restrict__ num_visible,
    const uint32_t *__restrict__ global_from_compact, const float *__restrict__ v_compact,
    float4 *__restrict__ records, uint32_t max_rows, float half_w, float half_h, DetSums det) {
    const uint32_t V = min(min(*num_visible, vp.total_splats), max_rows);
    for (uint32_t c = blockIdx.x * kThreads + threadIdx.x; c < V; c += gridDim.x * kThreads) {
        const uint32_t g = global_from_compact[c];
        float4 r0, r1, r2;
        load_compact_sums(v_compact, det, c, r0, r1, r2);
        const float vxy[2] = {r0.x, r0.y};
        const float vconic[3] = {r0.z, r0.w, r1.x};
        const float mean[3] = {means[(size_t)g * 3], means[(size_t)g * 3 + 1], means[(size_t)g * 3 + 2]};
        const float scale[3] = {det_expf(log_scales[(size_t)g * 3]), det_expf(log_scales[(size_t)g * 3 + 1]),
                                det_expf(log_scales[(size_t)g * 3 + 2])};
        const float4 q4 = reinterpret_cast<const float4 *>(quats)[g];
        const float quat[4] = {q4.x, q4.y, q4.z, q4.w};
        const float sg = det_sigmoid(raw_opac[g]);
        const float o_opac = r2.x * (sg * (1.0f - sg));  // gather_grads.wgsl:224-227
        float o_mean[3], o_scale[3], o_quat[4];
        splat_projection_vjp(vp, mean, scale, quat, vxy, vconic, o_mean, o_scale, o_quat);
        const float vx = vxy[0] * half_w, vy = vxy[1] * half_h;  // train.rs:300-302
        float4 *out = records + (size_t)c * (kRecFloats / 4);
        out[0] = make_float4(__uint_as_float(g), o_mean[0], o_mean[1], o_mean[2]);
        out[1] = make_float4(o_scale[0], o_scale[1], o_scale[2], o_quat[0]);
        out[2] = make_float4(o_quat[1], o_quat[2], o_quat[3], o_opac);
        out[3] = make_float4(r1.y, r1.z, r1.w, sqrtf(vx * vx + vy * vy));
    }
}

// index[v * n + gid] = row of splat gid in view v's records.  A reader validates an entry by checking
// row < view_rows[v] and records[v][row].gid == gid (a gid appears at most once per view), so stale or uninitialised
// entries are harmless; it clears the entries it consumes, so a buffer that started as all-ones stays clean and the
// check of an entry nobody wrote this step costs no gather.
__global__ __launch_bounds__(kThreads) void k_build_view_index(const float4 *__restrict__ records, uint32_t num_views,
                                                               uint32_t rows_per_view,
                                                               const uint32_t *__restrict__ view_rows,
                                                               const uint32_t *__restrict__ view_offsets, uint32_t n,
                                                               uint32_t *__restrict__ index) {
    // view_offsets == nullptr: view v owns rows [v * rows_per_view, + view_rows[v]); otherwise the views are packed,
    // view v owns rows [view_offsets[v], + view_rows[v]) of a buffer of rows_per_view rows in all.
    for (uint32_t v = 0; v < num_views; v++) {  // uniform: a handful of views
        const uint32_t first = view_offsets ? view_offsets[v] : v * rows_per_view;
        const uint32_t room = view_offsets ? (first < rows_per_view ? rows_per_view - first : 0u) : rows_per_view;
        const uint32_t cnt = min(view_rows[v], room);
        for (uint32_t r = blockIdx.x * kThreads + threadIdx.x; r < cnt; r += gridDim.x * kThreads) {
            const uint32_t gid = __float_as_uint(records[(size_t)(first + r) * (kRecFloats / 4)].x);
            if (gid < n) index[(size_t)v * n + gid] = r;
        }
    }
}

// The per-view sums of one splat (fixed view order: the same bits on every rank).  `add_sh(Y, v_rgb)` accumulates the
// splat's v_sh row wherever the caller keeps it.  A consumed index entry is cleared, so a buffer that was all-ones before
// its first use stays free of stale entries (the record is still checked: correctness never depends on that).
struct ViewSums {
    float mean[3] = {0.f, 0.f, 0.f}, scale[3] = {0.f, 0.f, 0.f}, quat[4] = {0.f, 0.f, 0.f, 0.f};
    float opac = 0.f, stat_norm = 0.f, stat_count = 0.f;
};
template <int DEG, typename AddSh>
__device__ __forceinline__ void sum_view_records(const float4 *__restrict__ records, uint32_t num_views,
                                                 uint32_t rows_per_view, const uint32_t *__restrict__ view_rows,
                                                 const uint32_t *__restrict__ view_offsets,
                                                 const float *__restrict__ campos, uint32_t *__restrict__ index,
                                                 const float *means, uint32_t n, uint32_t g, ViewSums &o, AddSh add_sh) {
    constexpr uint32_t ncoef = (DEG + 1) * (DEG + 1);
    constexpr uint32_t kChunk = 8;  // views whose index entries / record heads are in flight together
    const float mean[3] = {means[(size_t)g * 3], means[(size_t)g * 3 + 1], means[(size_t)g * 3 + 2]};
    for (uint32_t v0 = 0; v0 < num_views; v0 += kChunk) {
        // One memory phase for the index entries of up to 8 views, one for the heads of the records they point to
        // (round 2 walked the views one by one: two dependent loads per view, 117 us at 8 views where one view takes 43),
        // then the sums in view order: the same bits on every rank.
        uint32_t r[kChunk];
#pragma unroll
        for (uint32_t j = 0; j < kChunk; j++) r[j] = v0 + j < num_views ? index[(size_t)(v0 + j) * n + g] : kInvalid;
        const float4 *rec[kChunk];
        float4 a[kChunk];
#pragma unroll
        for (uint32_t j = 0; j < kChunk; j++) {
            const uint32_t v = v0 + j;
            rec[j] = nullptr;
            a[j] = make_float4(__uint_as_float(kInvalid), 0.f, 0.f, 0.f);
            if (v < num_views) {
                const uint32_t first = view_offsets ? view_offsets[v] : v * rows_per_view;
                const uint32_t room = view_offsets ? (first < rows_per_view ? rows_per_view - first : 0u) : rows_per_view;
                if (r[j] < min(view_rows[v], room)) {
                    rec[j] = records + ((size_t)first + r[j]) * (kRecFloats / 4);
                    a[j] = rec[j][0];
                }
            }
        }
#pragma unroll
        for (uint32_t j = 0; j < kChunk; j++) {
            const uint32_t v = v0 + j;
            if (rec[j] == nullptr || __float_as_uint(a[j].x) != g) continue;  // no entry / stale index entry
            index[(size_t)v * n + g] = kInvalid;
            const float4 b = rec[j][1], c = rec[j][2], d = rec[j][3];
            o.mean[0] += a[j].y, o.mean[1] += a[j].z, o.mean[2] += a[j].w;
            o.scale[0] += b.x, o.scale[1] += b.y, o.scale[2] += b.z;
            o.quat[0] += b.w, o.quat[1] += c.x, o.quat[2] += c.y, o.quat[3] += c.z;
            o.opac += c.w;
            o.stat_norm += d.w;
            o.stat_count += 1.0f;
            // gather_grads.wgsl:182-222 with this view's camera term (viewmat[3].xyz)
            float dir[3] = {mean[0] - campos[v * 3], mean[1] - campos[v * 3 + 1], mean[2] - campos[v * 3 + 2]};
            const float len = sqrtf(dir[0] * dir[0] + dir[1] * dir[1] + dir[2] * dir[2]);
            dir[0] = dir[0] / len, dir[1] = dir[1] / len, dir[2] = dir[2] / len;
            float Y[ncoef];
            sh_basis<ncoef>(DEG, dir, Y);
            add_sh(Y, d);
        }
    }
}

// Fused with Adam: the summed rows go through the per-wave LDS staging of store_gradients_or_step.
template <int DEG>
__global__ __launch_bounds__(kThreads) void k_reduce_view_records_adam(
    const float4 *__restrict__ records, uint32_t num_views, uint32_t rows_per_view,
    const uint32_t *__restrict__ view_rows, const uint32_t *__restrict__ view_offsets,
    const float *__restrict__ campos, uint32_t *__restrict__ index, const float *means, uint32_t n, AdamFuse af) {
    constexpr uint32_t ncoef = (DEG + 1) * (DEG + 1);
    constexpr uint32_t kRow = ncoef * 3, kRowPad = kRow | 1u;
    constexpr uint32_t kStageFloats = (kWave * kRowPad > 512u ? kWave * kRowPad : 512u);
    __shared__ float stage_all[kThreads / kWave][kStageFloats];
    const uint32_t wv = threadIdx.x / kWave, lane = threadIdx.x & (kWave - 1);
    float *stage = stage_all[wv];
    const uint32_t g0 = blockIdx.x * kThreads + wv * kWave;
    if (g0 >= n) return;  // wave-uniform; the kernel has no workgroup barrier
    const uint32_t g = g0 + lane;
    ViewSums o;
    float *row = stage + lane * kRowPad;
#pragma unroll
    for (uint32_t k = 0; k < kRow; k++) row[k] = 0.f;
    if (g < n)
        sum_view_records<DEG>(records, num_views, rows_per_view, view_rows, view_offsets, campos, index, means, n, g, o,
                              [&](const float *Y, const float4 &d) {
#pragma unroll
                                  for (uint32_t k = 0; k < ncoef; k++) {
                                      row[k * 3 + 0] += Y[k] * d.x;
                                      row[k * 3 + 1] += Y[k] * d.y;
                                      row[k * 3 + 2] += Y[k] * d.z;
                                  }
                              });
    const float zero2[2] = {0.f, 0.f}, zero3[3] = {0.f, 0.f, 0.f};
    // deferred SH: a splat some view saw (stat_count != 0) has its block caught up and stepped, the others wait
    __shared__ uint32_t row_t0_all[kThreads / kWave][kWave];
    uint32_t *row_t0 = row_t0_all[wv];
    const bool seen = af.lazy.on() && g < n && o.stat_count != 0.0f;
    if (af.lazy.on()) {
        row_t0[lane] = seen ? af.lazy.sh_time[g] : kInvalid;
        __builtin_amdgcn_wave_barrier();
    }
    store_gradients_or_step<DEG, true, true>(af, n, g0, lane, stage, row_t0, o.mean, o.scale, o.quat, o.opac, zero2,
                                             o.stat_norm, o.stat_count, nullptr, zero3, nullptr, nullptr, nullptr, nullptr,
                                             nullptr, nullptr);
    if (seen) af.lazy.sh_time[g] = af.lazy.now + 1u;
}

// Dense sum: like the dense backward, the zeros of the splats no view sees are stored straight from registers by the
// lanes that own the addresses, and a splat some view sees keeps its v_sh row in registers and writes its rows itself.
template <int DEG>
__global__ __launch_bounds__(kThreads) void k_reduce_view_records_dense(
    const float4 *__restrict__ records, uint32_t num_views, uint32_t rows_per_view,
    const uint32_t *__restrict__ view_rows, const uint32_t *__restrict__ view_offsets,
    const float *__restrict__ campos, uint32_t *__restrict__ index, const float *means, uint32_t n,
    float *__restrict__ v_means, float *__restrict__ v_scales,
    float *__restrict__ v_quats, float *__restrict__ v_sh, float *__restrict__ v_opac) {
    constexpr uint32_t ncoef = (DEG + 1) * (DEG + 1);
    constexpr uint32_t kRow = ncoef * 3;
    const uint32_t lane = threadIdx.x & (kWave - 1);
    const uint32_t g0 = blockIdx.x * kThreads + (threadIdx.x / kWave) * kWave;
    if (g0 >= n) return;  // wave-uniform; the kernel has no workgroup barrier
    const size_t g = (size_t)g0 + lane;
    ViewSums o;
    float row[kRow];
#pragma unroll
    for (uint32_t k = 0; k < kRow; k++) row[k] = 0.f;
    if (g < n)
        sum_view_records<DEG>(records, num_views, rows_per_view, view_rows, view_offsets, campos, index, means, n,
                              (uint32_t)g, o,
                              [&](const float *Y, const float4 &d) {
#pragma unroll
                                  for (uint32_t k = 0; k < ncoef; k++) {
                                      row[k * 3 + 0] += Y[k] * d.x;
                                      row[k * 3 + 1] += Y[k] * d.y;
                                      row[k * 3 + 2] += Y[k] * d.z;
                                  }
                              });
    const bool seen = o.stat_count != 0.0f;
    zero_invisible_rows<DEG>(n, g0, lane, __ballot(seen), v_means, nullptr, v_scales, v_quats, v_sh, v_opac);
    if (seen) {
        reinterpret_cast<float4 *>(v_quats)[g] = make_float4(o.quat[0], o.quat[1], o.quat[2], o.quat[3]);
        v_opac[g] = o.opac;
#pragma unroll
        for (int k = 0; k < 3; k++) v_means[g * 3 + k] = o.mean[k], v_scales[g * 3 + k] = o.scale[k];
        float *dst = v_sh + g * kRow;
        if constexpr (kRow % 4 == 0) {
#pragma unroll
            for (uint32_t j = 0; j < kRow / 4; j++)
                reinterpret_cast<float4 *>(dst)[j] = make_float4(row[4 * j], row[4 * j + 1], row[4 * j + 2], row[4 * j + 3]);
        } else {
#pragma unroll
            for (uint32_t e = 0; e < kRow; e++) dst[e] = row[e];
        }
    }
}

// brush_lazy_sh_flush: one lane per 16-byte chunk of the SH block; a chunk behind lazy.now replays its pending steps.
// sh_time is only read here (all chunks of a row read the same word); the launcher sets it to `now` afterwards.
__global__ __launch_bounds__(kThreads) void k_lazy_sh_flush(LazySh lazy, float *__restrict__ sh, uint32_t chunks_per_row,
                                                            uint64_t total_chunks) {
    const uint64_t i = (uint64_t)blockIdx.x * kThreads + threadIdx.x;
    if (i >= total_chunks) return;
    const uint32_t g = (uint32_t)(i / chunks_per_row), k0 = (uint32_t)(i - (uint64_t)g * chunks_per_row) * 4u;
    const uint32_t t0 = lazy.sh_time[g];
    if (t0 >= lazy.now) return;
    float4 x = nt_load4(sh + i * 4), m = nt_load4(lazy.m1 + i * 4), v = nt_load4(lazy.m2 + i * 4);
    lazy_replay4(lazy, t0, k0, m, v, x);
    nt_store4(sh + i * 4, x);
    nt_store4(lazy.m1 + i * 4, m);
    nt_store4(lazy.m2 + i * 4, v);
}
__global__ __launch_bounds__(kThreads) void k_fill_u32(uint32_t *__restrict__ dst, uint32_t value, uint32_t n) {
    const uint32_t i = blockIdx.x * kThreads + threadIdx.x;
    if (i < n) dst[i] = value;
}

}  // namespace

hipError_t launch_lazy_sh_flush(const LazySh &lazy, float *sh, uint32_t n, uint32_t row_floats, hipStream_t s) {
    const uint64_t chunks = (uint64_t)n * (row_floats / 4u);
    if (chunks == 0) return hipSuccess;
    if (chunks / kThreads >= 0x7FFFFFFFull) return hipErrorInvalidValue;
    hipLaunchKernelGGL(k_lazy_sh_flush, dim3((uint32_t)((chunks + kThreads - 1) / kThreads)), dim3(kThreads), 0, s, lazy, sh,
                       row_floats / 4u, chunks);
    hipLaunchKernelGGL(k_fill_u32, dim3(ceil_div(n, kThreads)), dim3(kThreads), 0, s, lazy.sh_time, lazy.now, n);
    return hipGetLastError();
}

hipError_t launch_zero_compact_grads(const uint32_t *num_visible, uint32_t n, float *v_compact, hipStream_t s) {
    const uint32_t grid = max(1u, min(ceil_div(n * 3u, kThreads), 1024u));
    hipLaunchKernelGGL(k_zero_compact_grads, dim3(grid), dim3(kThreads), 0, s, num_visible, n,
                       reinterpret_cast<float4 *>(v_compact));
    return hipGetLastError();
}

hipError_t launch_project_backward(const ViewParams &vp, const float *means, const float *log_scales,
                                   const float *quats, const float *raw_opac,
                                   const uint32_t *compact_from_global, const float *v_compact, float *v_means,
                                   float *v_xy, float *v_scales, float *v_quats, float *v_sh, float *v_opac,
                                   const AdamFuse *adam, const DetSumsArgs &dargs, bool prezeroed, hipStream_t s) {
    const uint32_t n = vp.total_splats;
    if (n == 0) return hipSuccess;
    const dim3 grid(ceil_div(n, kThreads)), block(kThreads);
    AdamFuse af{};
    if (adam) af = *adam;
    const DetSums det{dargs.cum_tiles_hit, dargs.num_intersections, dargs.partials, dargs.cap};
    if (adam && af.lazy.on()) {  // SH block under deferred Adam
        if (vp.sh_degree == 1)
            hipLaunchKernelGGL(k_project_backward_lazy<1>, grid, block, 0, s, vp, means, log_scales, quats, raw_opac,
                               compact_from_global, v_compact, v_xy, af, det);
        else
            hipLaunchKernelGGL(k_project_backward_lazy<3>, grid, block, 0, s, vp, means, log_scales, quats, raw_opac,
                               compact_from_global, v_compact, v_xy, af, det);
        return hipGetLastError();
    }
#define BRUSH_LAUNCH_PB(D)                                                                                      \
    if (adam)                                                                                                   \
        hipLaunchKernelGGL((k_project_backward<D, true>), grid, block, 0, s, vp, means, log_scales, quats,      \
                           raw_opac, compact_from_global, v_compact, v_means, v_xy, v_scales, v_quats, v_sh,    \
                           v_opac, af, det);                                                                    \
    else if (prezeroed)                                                                                         \
        hipLaunchKernelGGL((k_project_backward<D, false, true>), grid, block, 0, s, vp, means, log_scales,      \
                           quats, raw_opac, compact_from_global, v_compact, v_means, v_xy, v_scales, v_quats,   \
                           v_sh, v_opac, af, det);                                                              \
    else                                                                                                        \
        hipLaunchKernelGGL((k_project_backward<D, false>), grid, block, 0, s, vp, means, log_scales, quats,     \
                           raw_opac, compact_from_global, v_compact, v_means, v_xy, v_scales, v_quats, v_sh,    \
                           v_opac, af, det)
    switch (vp.sh_degree) {
        case 0: BRUSH_LAUNCH_PB(0); break;
        case 1: BRUSH_LAUNCH_PB(1); break;
        case 2: BRUSH_LAUNCH_PB(2); break;
        case 3: BRUSH_LAUNCH_PB(3); break;
        default: BRUSH_LAUNCH_PB(4); break;
    }
#undef BRUSH_LAUNCH_PB
    return hipGetLastError();
}

hipError_t launch_project_backward_records(const ViewParams &vp, const float *means, const float *log_scales,
                                           const float *quats, const float *raw_opac, const uint32_t *num_visible,
                                           const uint32_t *global_from_compact, const float *v_compact,
                                           float *records, uint32_t max_rows, const DetSumsArgs &dargs, hipStream_t s) {
    if (vp.total_splats == 0 || max_rows == 0) return hipSuccess;
    const uint32_t rows = min(vp.total_splats, max_rows);
    const DetSums det{dargs.cum_tiles_hit, dargs.num_intersections, dargs.partials, dargs.cap};
    hipLaunchKernelGGL(k_project_backward_records, dim3(min(ceil_div(rows, kThreads), 2048u)), dim3(kThreads), 0, s, vp,
                       means, log_scales, quats, raw_opac, num_visible, global_from_compact, v_compact,
                       reinterpret_cast<float4 *>(records), max_rows, (float)vp.img_size[0] / 2.0f,
                       (float)vp.img_size[1] / 2.0f, det);
    return hipGetLastError();
}

hipError_t launch_reduce_view_records(const float *records, uint32_t num_views, uint32_t rows_per_view,
                                      const uint32_t *view_rows, const uint32_t *view_offsets, const float *campos,
                                      const float *means, uint32_t n,
                                      uint32_t sh_degree, uint32_t *index, float *v_means, float *v_scales,
                                      float *v_quats, float *v_sh, float *v_opac, const AdamFuse *adam, hipStream_t s) {
    if (n == 0) return hipSuccess;
    const float4 *rec4 = reinterpret_cast<const float4 *>(records);
    if (num_views > 0 && rows_per_view > 0)  // (a u32 product would wrap to 0 at 8 views x 2^29 rows and skip the index)
        hipLaunchKernelGGL(k_build_view_index, dim3(min(ceil_div(rows_per_view, kThreads), 2048u)), dim3(kThreads), 0, s,
                           rec4, num_views, rows_per_view, view_rows, view_offsets, n, index);
    const dim3 grid(ceil_div(n, kThreads)), block(kThreads);
    AdamFuse af{};
    if (adam) af = *adam;
#define BRUSH_LAUNCH_RV(D)                                                                                         \
    if (adam)                                                                                                      \
        hipLaunchKernelGGL((k_reduce_view_records_adam<D>), grid, block, 0, s, rec4, num_views, rows_per_view,     \
                           view_rows, view_offsets, campos, index, means, n, af);                                  \
    else                                                                                                           \
        hipLaunchKernelGGL((k_reduce_view_records_dense<D>), grid, block, 0, s, rec4, num_views, rows_per_view,    \
                           view_rows, view_offsets, campos, index, means, n, v_means, v_scales, v_quats, v_sh, v_opac)
    switch (sh_degree) {
        case 0: BRUSH_LAUNCH_RV(0); break;
        case 1: BRUSH_LAUNCH_RV(1); break;
        case 2: BRUSH_LAUNCH_RV(2); break;
        case 3: BRUSH_LAUNCH_RV(3); break;
        default: BRUSH_LAUNCH_RV(4); break;
    }
#undef BRUSH_LAUNCH_RV
    return hipGetLastError();
}

hipError_t launch_sum_isect_rows(const float *rows, const uint32_t *num_intersections, const uint32_t *cum_tiles_hit,
                                 uint32_t cap, float *v_compact, float *partials, hipStream_t s) {
    if (cap == 0) return hipSuccess;
    const uint32_t chunks = ceil_div(cap, kWave);
    hipLaunchKernelGGL(k_sum_isect_rows, dim3(min(ceil_div(chunks, kThreads / kWave), 4096u)), dim3(kThreads), 0, s,
                       reinterpret_cast<const float4 *>(rows), num_intersections, cum_tiles_hit, cap,
                       reinterpret_cast<float4 *>(v_compact), reinterpret_cast<float4 *>(partials));
    return hipGetLastError();
}

}  // namespace brush
