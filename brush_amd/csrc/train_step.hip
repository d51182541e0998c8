// train_step.hip — the parts of the reference's training iteration that sit either side of the op
// (SURVEY §8(f) row 1): the image loss with its gradient, the optimizer step and the screen-space
// gradient statistics.  The reference expresses these as Burn tensor ops (≈60 launches); here each is
// one or two HBM-streaming kernels.
//
//   k_ssim_forward / k_ssim_backward : loss = L1*(1-w) - SSIM*w (train.rs:243-268) with the SSIM of
//       ssim.rs:42-101 (Gaussian window sigma 1.5, 11x11 by default, zero padding div_ceil(window,2), so the SSIM
//       map of an odd window is (h+2)x(w+2); variances clamped at 0) and d loss / d pred.  The window is outer(g, g), so
//       every blur is two 1-D passes (same linear operator, ssim.rs:17-32 notes it as a TODO): horizontal
//       through a per-wave LDS row buffer, vertical over a register ring while the wave marches down.
//   k_adam : Adam with the reference's five learning rates and the SH-rest lerp (train.rs:318-359) over
//       the gradient arrays the backward writes; update rule of burn 0.16 `Adam::step`
//       (m/(1-b1^t) / (sqrt(v/(1-b2^t)) + eps)), eps = 1e-15 (train.rs:184).  The single-view trainer uses
//       the fused form instead (project_bwd.hip, brush_render_backward_adam).
//   k_normalize_quats, k_refine_stats : gaussian_splats.rs:174-175, train.rs:284-316.
// Roofline: the blur kernels are VALU-issue bound (~200 / ~105 instructions per marched pixel-channel-row forward /
// backward, of which 143 / 62 are the taps themselves; the file is built without the SLP vectorizer, whose v_pk_* pairs
// cost more register moves than they save, and addresses are a wave-uniform row pointer plus a per-lane constant), Adam
// is an HBM stream (28 B/parameter).
#include "internal.hpp"

namespace brush {
namespace {

// The SSIM window: TrainConfig::ssim_window_size (train.rs:63, default 11); odd sizes 3..15 are compiled.  For an odd
// window 2m+1 the zero padding is div_ceil(window, 2) = m+1 (ssim.rs:49), so the SSIM map is (h+2) x (w+2) whatever
// the size.
constexpr int kMaxWin = 15;
template <int WIN>
struct Geo {
    static constexpr int kPad = (WIN + 1) / 2;      // div_ceil(WIN, 2)
    static constexpr int kOutCols = 64 - (WIN - 1);  // columns a wave produces: 64 lanes minus the halo
    static constexpr int kSegRows = 3 * WIN + 1;     // rows a block produces: + (WIN - 1) halo = 4 * WIN marched rows
    static constexpr int kOff = WIN - 1 - kPad;
};
constexpr int kRowBuf = 80;     // floats per LDS row buffer (lane + kMaxWin - 1 taps < 80)
constexpr float kC1 = 0.01f * 0.01f, kC2 = 0.03f * 0.03f;

struct Window {
    float g[kMaxWin];
};

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

// Sum over the 256 threads of a block in a fixed order (deterministic); valid in thread 0.
__device__ __forceinline__ float block_sum(float v, float *red) {
    v = wave_sum(v);
    if (lane_id() == 0) red[threadIdx.x / kWave] = v;
    __syncthreads();
    const float r = (red[0] + red[1]) + (red[2] + red[3]);
    __syncthreads();
    return r;
}

// Element at a wave-uniform base plus a per-lane BYTE offset below 4 GiB: global_load/store with an SGPR base and a
// 32-bit VGPR offset, no 64-bit vector address arithmetic.
__device__ __forceinline__ float ld_off(const float *base, uint32_t byte_off) {
    return *reinterpret_cast<const float *>(reinterpret_cast<const char *>(base) + byte_off);
}
__device__ __forceinline__ void st_off(float *base, uint32_t byte_off, float v) {
    *reinterpret_cast<float *>(reinterpret_cast<char *>(base) + byte_off) = v;
}

// LDS traffic inside one wave is in order; this only stops the compiler from moving accesses.
// A fence would also wait for the prefetched global loads, so this is a pure compiler barrier.
__device__ __forceinline__ void wave_lds_sync() {
    asm volatile("" ::: "memory");
    __builtin_amdgcn_wave_barrier();
    asm volatile("" ::: "memory");
}

// Both loss kernels march a wave down a strip of 64 columns for one colour channel (wave = channel,
// block = 3 waves).  Per marched row: the lane's value goes to a per-wave LDS row buffer, the
// 11-tap horizontal blur is read back from it (lane + k), and the vertical blur runs over a ring
// of the last 11 blurred rows kept in registers (the 11-fold unrolled loop makes the ring static).
// No block barrier, 2 KB of LDS per block: occupancy is bounded by registers only.

// SSIM map position (oy, ox) for oy in the block's 34 rows, ox in the wave's 54 columns: writes the
// three derivative maps (wrt blur(a), blur(a*a), blur(a*b); a = pred, b = gt) scaled by `coef`, and
// per-wave partial sums of the SSIM map and of |pred - gt| (each input pixel owned by the wave
// holding map position (iy+1, ix+1)).
template <int WIN>
__global__ __launch_bounds__(192) void k_ssim_forward(const float *__restrict__ pred, const float *__restrict__ gt,
                                                      uint32_t gt_channels, uint32_t w, uint32_t h, Window win,
                                                      float coef, float *__restrict__ dmaps,
                                                      float *__restrict__ partials) {
    using G = Geo<WIN>;
    __shared__ float rows[3][2][kRowBuf];
    const int ch = threadIdx.x / kWave, l = lane_id();
    float *ra = rows[ch][0], *rb = rows[ch][1];
    const int W2 = w + 2, H2 = h + 2;
    const int x0 = blockIdx.x * G::kOutCols, oy0 = blockIdx.y * G::kSegRows;
    const int ix = x0 - G::kPad + l;
    const bool col_ok = ix >= 0 && ix < (int)w;
    const int ox = x0 + l;
    const bool out_col = l < G::kOutCols && ox < W2;
    const bool own_col = l >= G::kPad - 1 && l < G::kPad - 1 + G::kOutCols;
    const uint32_t plane = (uint32_t)W2 * (uint32_t)H2;  // all element offsets fit 32 bits (checked by the host entry)
    float hq[WIN][5];
    float msum = 0.0f, l1 = 0.0f;
    // marched row r -> (a, b, alpha pair) of the lane's column, zero outside the image.  Loads are
    // unconditional (clamped address + select) and issued three rows ahead of their use, so the
    // vmcnt waits the compiler places leave the younger rows in flight.
    // Addresses: the row is wave-uniform (scalar pointer arithmetic), the column offset a per-lane constant.
    const bool alpha_on = ch == 0 && gt_channels == 4;
    const uint32_t ixc = (uint32_t)min(max(ix, 0), (int)w - 1);
    const uint32_t p_al = alpha_on ? 3u : (uint32_t)ch;
    const uint32_t pc_a = (ixc * 4u + (uint32_t)ch) * 4u, pc_al = (ixc * 4u + p_al) * 4u;  // pred column byte offsets
    const uint32_t gc_a = (ixc * gt_channels + (uint32_t)ch) * 4u, gc_al = (ixc * gt_channels + p_al) * 4u;  // gt
    struct Row {
        float a, b, pa, ga;
    };
    auto fetch = [&](int r) {
        const int iy = oy0 - G::kPad + r;  // uniform
        const bool row_ok = iy >= 0 && iy < (int)h;
        const uint32_t iyc = (uint32_t)min(max(iy, 0), (int)h - 1);
        const float *prow = pred + (size_t)iyc * w * 4u;
        const float *grow = gt + (size_t)iyc * w * gt_channels;
        const bool ok = col_ok && row_ok;
        Row v;
        v.a = ld_off(prow, pc_a);
        v.b = ld_off(grow, gc_a);
        v.pa = v.ga = 0.0f;
        if (alpha_on) {  // wave-uniform: the strided loads are what the kernel waits for (16 cache lines per instruction)
            v.pa = ld_off(prow, pc_al);
            v.ga = ld_off(grow, gc_al);
            v.pa = ok ? v.pa : 0.0f, v.ga = ok ? v.ga : 0.0f;
        }
        v.a = ok ? v.a : 0.0f, v.b = ok ? v.b : 0.0f;
        return v;
    };
    // three rows in flight; slot j % 3 of the ring is static inside the WIN-fold unrolled body, and the ring is turned
    // by WIN % 3 once per body instead of shifting it every row
    Row pf[3] = {fetch(0), fetch(1), fetch(2)};
    for (int r0 = 0; r0 < G::kSegRows + WIN - 1; r0 += WIN) {
#pragma unroll
        for (int j = 0; j < WIN; j++) {
            const int r = r0 + j;
            const Row c = pf[j % 3];
            ra[l] = c.a, rb[l] = c.b;
            if (own_col && r >= G::kPad - 1 && r < G::kPad - 1 + G::kSegRows) l1 += fabsf(c.a - c.b) + fabsf(c.pa - c.ga);
            pf[j % 3] = fetch(r + 3);
            wave_lds_sync();
            float sa, sb, saa, sbb, sab;
            {
                const float av = ra[l], bv = rb[l];
                const float ga = win.g[0] * av, gb = win.g[0] * bv;
                sa = ga, sb = gb, saa = ga * av, sbb = gb * bv, sab = ga * bv;
            }
#pragma unroll
            for (int k = 1; k < WIN; k++) {
                const float av = ra[l + k], bv = rb[l + k];
                const float ga = win.g[k] * av, gb = win.g[k] * bv;
                sa += ga, sb += gb, saa += ga * av, sbb += gb * bv, sab += ga * bv;
            }
            wave_lds_sync();
            hq[j][0] = sa, hq[j][1] = sb, hq[j][2] = saa, hq[j][3] = sbb, hq[j][4] = sab;
            const int oy = oy0 + r - (WIN - 1);
            if (r >= WIN - 1 && oy < H2) {  // ring slot of marched row r - (WIN - 1) + k is (j + 1 + k) % WIN
                float v[5];
#pragma unroll
                for (int q = 0; q < 5; q++) v[q] = win.g[0] * hq[(j + 1) % WIN][q];
#pragma unroll
                for (int k = 1; k < WIN; k++) {
#pragma unroll
                    for (int q = 0; q < 5; q++) v[q] += win.g[k] * hq[(j + 1 + k) % WIN][q];
                }
                if (out_col) {
                    const float mx = v[0], my = v[1];
                    const float mu_xx = mx * mx, mu_yy = my * my, mu_xy = mx * my;
                    const float sxx_raw = v[2] - mu_xx;
                    const float sxx = fmaxf(sxx_raw, 0.0f), syy = fmaxf(v[3] - mu_yy, 0.0f), sxy = v[4] - mu_xy;
                    const float A1 = mu_xy * 2.0f + kC1, A2 = sxy * 2.0f + kC2;
                    const float B1 = mu_xx + mu_yy + kC1, B2 = sxx + syy + kC2;
                    // B1 >= C1, B2 >= C2: v_rcp_f32 (1 ulp) twice instead of three IEEE divisions (10 instructions each)
                    const float i1 = __builtin_amdgcn_rcpf(B1), i2 = __builtin_amdgcn_rcpf(B2);
                    const float inv = i1 * i2;
                    const float m = A1 * A2 * inv;
                    msum += m;
                    const float d_eab = 2.0f * A1 * inv;
                    const float d_eaa = sxx_raw >= 0.0f ? -m * i2 : 0.0f;  // clamp_min(0) passes the gradient at >= 0
                    const float d_mu = 2.0f * my * (A2 - A1) * inv - 2.0f * mx * (m * i1) - 2.0f * mx * d_eaa;
                    float *drow = dmaps + (size_t)ch * plane + (size_t)oy * (uint32_t)W2;  // uniform
                    st_off(drow, (uint32_t)ox * 4u, d_mu * coef);
                    st_off(drow + (size_t)3u * plane, (uint32_t)ox * 4u, d_eaa * coef);
                    st_off(drow + (size_t)6u * plane, (uint32_t)ox * 4u, d_eab * coef);
                }
            }
        }
        if constexpr (WIN % 3 == 1) {
            const Row t = pf[0];
            pf[0] = pf[1], pf[1] = pf[2], pf[2] = t;
        } else if constexpr (WIN % 3 == 2) {
            const Row t = pf[2];
            pf[2] = pf[1], pf[1] = pf[0], pf[0] = t;
        }
    }
    msum = wave_sum(msum), l1 = wave_sum(l1);
    const uint32_t nwave = gridDim.x * gridDim.y * 3, wv = (blockIdx.y * gridDim.x + blockIdx.x) * 3 + ch;
    if (l == 0) partials[wv] = msum, partials[nwave + wv] = l1;
}

// Image position (py, px): T[X](p) = sum_j g[j] X[p - kOff + j] per axis (the transposed blur; g is
// symmetric) of the three derivative maps, combined with the L1 term into d loss / d pred.  Wave 0
// of block (0,0) also reduces the partial sums into the loss value.
template <int WIN>
__global__ __launch_bounds__(192) void k_ssim_backward(const float *__restrict__ pred, const float *__restrict__ gt,
                                                       uint32_t gt_channels, uint32_t w, uint32_t h, Window win,
                                                       const float *__restrict__ dmaps, float l1_coef,
                                                       float *__restrict__ v_pred, const float *__restrict__ partials,
                                                       uint32_t nwave_fwd, float l1_weight, float ssim_weight,
                                                       float inv_l1_count, float inv_ssim_count,
                                                       float *__restrict__ loss) {
    using G = Geo<WIN>;
    __shared__ float rows[3][3][kRowBuf];
    const int ch = threadIdx.x / kWave, l = lane_id();
    const int W2 = w + 2, H2 = h + 2;
    const int px0 = blockIdx.x * G::kOutCols, py0 = blockIdx.y * G::kSegRows;
    const int ox = px0 - G::kOff + l;
    const bool col_ok = ox >= 0 && ox < W2;
    const int px = px0 + l;
    const bool out_col = l < G::kOutCols && px < (int)w;
    const uint32_t plane = (uint32_t)W2 * (uint32_t)H2;  // all element offsets fit 32 bits (checked by the host entry)
    const float *d0 = dmaps + (0u * 3u + ch) * plane, *d1 = dmaps + (1u * 3u + ch) * plane, *d2 = dmaps + (2u * 3u + ch) * plane;
    // l1_coef * sign(d): the sign bit of d on l1_coef, zero at d == 0
    auto lsgn = [&](float d) { return d != 0.0f ? __builtin_copysignf(l1_coef, d) * 1.0f : 0.0f; };
    float hq[WIN][3];
    // marched row r: the three map values of the lane's column and (a, b, alpha pair) of the pixel the
    // iteration will emit (row py0 + r - (WIN - 1)); unconditional loads three rows ahead, as in the forward.
    const bool alpha_on = gt_channels == 4;
    const uint32_t oxc = (uint32_t)min(max(ox, 0), W2 - 1), pxc = (uint32_t)min(px, (int)w - 1);
    const uint32_t g_al = alpha_on ? 3u : 0u;
    const bool alpha_row = alpha_on && ch == 0;  // the wave that writes v_pred's alpha
    const uint32_t pc_a = (pxc * 4u + (uint32_t)ch) * 4u, pc_al = (pxc * 4u + 3u) * 4u;  // byte offsets
    const uint32_t gc_a = (pxc * gt_channels + (uint32_t)ch) * 4u, gc_al = (pxc * gt_channels + g_al) * 4u;
    const uint32_t oc = oxc * 4u;
    struct Row {
        float x0, x1, x2, a, b, pa, ga;
    };
    auto fetch = [&](int r) {  // rows are wave-uniform: scalar pointers, per-lane constant column offsets
        const int oy = py0 - G::kOff + r;
        const bool ok = col_ok && oy >= 0 && oy < H2;
        const size_t orow = (size_t)(uint32_t)min(max(oy, 0), H2 - 1) * (uint32_t)W2;
        const uint32_t pyc = (uint32_t)min(max(py0 + r - (WIN - 1), 0), (int)h - 1);
        const float *prow = pred + (size_t)pyc * w * 4u;
        const float *grow = gt + (size_t)pyc * w * gt_channels;
        Row v;
        v.x0 = ld_off(d0 + orow, oc), v.x1 = ld_off(d1 + orow, oc), v.x2 = ld_off(d2 + orow, oc);
        v.a = ld_off(prow, pc_a), v.b = ld_off(grow, gc_a);
        v.pa = v.ga = 0.0f;
        if (alpha_row) v.pa = ld_off(prow, pc_al), v.ga = ld_off(grow, gc_al);  // wave-uniform
        v.x0 = ok ? v.x0 : 0.0f, v.x1 = ok ? v.x1 : 0.0f, v.x2 = ok ? v.x2 : 0.0f;
        return v;
    };
    Row pf[3] = {fetch(0), fetch(1), fetch(2)};  // ring turned once per unrolled body, as in the forward
    for (int r0 = 0; r0 < G::kSegRows + WIN - 1; r0 += WIN) {
#pragma unroll
        for (int j = 0; j < WIN; j++) {
            const int r = r0 + j;
            const Row c = pf[j % 3];
            rows[ch][0][l] = c.x0, rows[ch][1][l] = c.x1, rows[ch][2][l] = c.x2;
            const float a = c.a, b = c.b, pa = c.pa, ga = c.ga;
            pf[j % 3] = fetch(r + 3);
            wave_lds_sync();
            float s0 = win.g[0] * rows[ch][0][l], s1 = win.g[0] * rows[ch][1][l], s2 = win.g[0] * rows[ch][2][l];
#pragma unroll
            for (int k = 1; k < WIN; k++) {
                s0 += win.g[k] * rows[ch][0][l + k];
                s1 += win.g[k] * rows[ch][1][l + k];
                s2 += win.g[k] * rows[ch][2][l + k];
            }
            wave_lds_sync();
            hq[j][0] = s0, hq[j][1] = s1, hq[j][2] = s2;
            const int py = py0 + r - (WIN - 1);
            if (r >= WIN - 1 && py < (int)h && out_col) {
                float t[3];
#pragma unroll
                for (int q = 0; q < 3; q++) t[q] = win.g[0] * hq[(j + 1) % WIN][q];
#pragma unroll
                for (int k = 1; k < WIN; k++) {
#pragma unroll
                    for (int q = 0; q < 3; q++) t[q] += win.g[k] * hq[(j + 1 + k) % WIN][q];
                }
                float *vrow = v_pred + (size_t)(uint32_t)py * w * 4u;  // uniform
                st_off(vrow, ((uint32_t)px * 4u + (uint32_t)ch) * 4u, t[0] + 2.0f * a * t[1] + b * t[2] + lsgn(a - b));
                if (ch == 0)  // alpha: compared only when the target has alpha (train.rs:248-252)
                    st_off(vrow, ((uint32_t)px * 4u + 3u) * 4u, alpha_on ? lsgn(pa - ga) : 0.0f);
            }
        }
        if constexpr (WIN % 3 == 1) {
            const Row t = pf[0];
            pf[0] = pf[1], pf[1] = pf[2], pf[2] = t;
        } else if constexpr (WIN % 3 == 2) {
            const Row t = pf[2];
            pf[2] = pf[1], pf[1] = pf[0], pf[0] = t;
        }
    }
    if (blockIdx.x == 0 && blockIdx.y == 0 && ch == 0) {
        float ms = 0.0f, ls = 0.0f;
        for (uint32_t i = l; i < nwave_fwd; i += kWave) ms += partials[i], ls += partials[nwave_fwd + i];
        ms = wave_sum(ms), ls = wave_sum(ls);
        if (l == 0) loss[0] = ls * inv_l1_count * l1_weight - ms * inv_ssim_count * ssim_weight;
    }
}

// L1-only form (ssim_weight == 0, train.rs:254-266): one thread per pixel.
__global__ __launch_bounds__(256) void k_l1_backward(const float4 *__restrict__ pred, const float *__restrict__ gt,
                                                     uint32_t gt_channels, uint32_t npix, float l1_coef,
                                                     float4 *__restrict__ v_pred, float *__restrict__ partials) {
    __shared__ float red[4];
    float l1 = 0.0f;
    for (uint32_t i = blockIdx.x * 256 + threadIdx.x; i < npix; i += gridDim.x * 256) {
        const float4 p = pred[i];
        const float *gp = gt + (size_t)i * gt_channels;
        auto sgn = [](float d) { return d > 0.0f ? 1.0f : (d < 0.0f ? -1.0f : 0.0f); };
        const float d0 = p.x - gp[0], d1 = p.y - gp[1], d2 = p.z - gp[2], d3 = gt_channels == 4 ? p.w - gp[3] : 0.0f;
        l1 += fabsf(d0) + fabsf(d1) + fabsf(d2) + fabsf(d3);
        v_pred[i] = make_float4(l1_coef * sgn(d0), l1_coef * sgn(d1), l1_coef * sgn(d2), l1_coef * sgn(d3));
    }
    const float s = block_sum(l1, red);
    if (threadIdx.x == 0) partials[blockIdx.x] = s;
}

__global__ __launch_bounds__(64) void k_l1_finalize(const float *__restrict__ partials, uint32_t nblk, float inv_count,
                                                    float *__restrict__ loss) {
    float s = 0.0f;
    for (uint32_t i = threadIdx.x; i < nblk; i += kWave) s += partials[i];
    s = wave_sum(s);
    if (threadIdx.x == 0) loss[0] = s * inv_count;
}

Window make_window(int n) {
    Window win;
    float sum = 0.0f;
    for (int i = 0; i < kMaxWin; i++) win.g[i] = 0.0f;
    for (int i = 0; i < n; i++) {
        const float d = (float)i - (float)(n / 2);
        win.g[i] = expf(-(d * d) / (2.0f * 1.5f * 1.5f));  // ssim.rs:7-14
        sum += win.g[i];
    }
    for (int i = 0; i < n; i++) win.g[i] /= sum;
    return win;
}

// Upper bound of the forward's workgroup count over the supported windows (the partial-sum buffer is sized without
// knowing the window): the narrowest column strip (window 15) times the shortest row segment (window 3).
inline uint32_t loss_blocks_max(uint32_t w, uint32_t h) {
    return ceil_div(w + 2, (uint32_t)Geo<kMaxWin>::kOutCols) * ceil_div(h + 2, (uint32_t)Geo<3>::kSegRows);
}
inline bool window_ok(uint32_t n) { return n >= 3 && n <= (uint32_t)kMaxWin && (n & 1u); }
constexpr uint32_t kL1Blocks = 1024;

}  // namespace
}  // namespace brush

using namespace brush;

extern "C" int brush_loss_workspace_size(uint32_t w, uint32_t h, size_t *bytes) {
    if (!bytes || w == 0 || h == 0) return BRUSH_ERR_INVALID_ARG;
    const size_t plane = (size_t)(w + 2) * (h + 2);
    const size_t nblk = std::max<size_t>((size_t)loss_blocks_max(w, h) * 3, kL1Blocks);
    *bytes = align_up(9 * plane * sizeof(float), 256) + align_up(2 * nblk * sizeof(float), 256);
    return BRUSH_OK;
}

extern "C" int brush_l1_ssim_loss(const float *pred, const float *gt, uint32_t w, uint32_t h, uint32_t gt_channels,
                                  float ssim_weight, uint32_t ssim_window, float grad_scale, float *loss,
                                  float *v_pred, void *workspace, size_t workspace_bytes, brush_stream_t stream) {
    if (!pred || !gt || !loss || !v_pred || !workspace || w == 0 || h == 0) return BRUSH_ERR_INVALID_ARG;
    if (gt_channels != 3 && gt_channels != 4) return BRUSH_ERR_INVALID_ARG;
    if (ssim_weight > 0.0f && !window_ok(ssim_window)) return BRUSH_ERR_INVALID_ARG;  // odd sizes 3..15
    if (9ull * (w + 2ull) * (h + 2ull) >= (1ull << 32)) return BRUSH_ERR_INVALID_ARG;   // 32-bit element offsets (477 M pixels)
    size_t need = 0;
    brush_loss_workspace_size(w, h, &need);
    if (workspace_bytes < need) return BRUSH_ERR_WORKSPACE_SMALL;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const size_t plane = (size_t)(w + 2) * (h + 2);
    float *dmaps = static_cast<float *>(workspace);
    float *partials = reinterpret_cast<float *>(static_cast<char *>(workspace) + align_up(9 * plane * sizeof(float), 256));
    const float inv_l1 = 1.0f / ((float)w * (float)h * (float)gt_channels);
    const float4 *pred4 = reinterpret_cast<const float4 *>(pred);
    float4 *v4 = reinterpret_cast<float4 *>(v_pred);
    if (!(ssim_weight > 0.0f)) {
        const uint32_t npix = w * h, nblk = std::min(ceil_div(npix, 256u), kL1Blocks);
        hipLaunchKernelGGL(k_l1_backward, dim3(nblk), dim3(256), 0, s, pred4, gt, gt_channels, npix, grad_scale * inv_l1,
                           v4, partials);
        hipLaunchKernelGGL(k_l1_finalize, dim3(1), dim3(64), 0, s, partials, nblk, inv_l1, loss);
        BRUSH_HIP_CHECK(hipGetLastError());
        return BRUSH_OK;
    }
    const Window win = make_window((int)ssim_window);
    const float inv_ssim = 1.0f / (3.0f * (float)plane);
#define BRUSH_SSIM(W)                                                                                                \
    do {                                                                                                             \
        using G = Geo<W>;                                                                                            \
        const dim3 gf(ceil_div(w + 2, (uint32_t)G::kOutCols), ceil_div(h + 2, (uint32_t)G::kSegRows));               \
        const dim3 gb(ceil_div(w, (uint32_t)G::kOutCols), ceil_div(h, (uint32_t)G::kSegRows));                       \
        hipLaunchKernelGGL(k_ssim_forward<W>, gf, dim3(192), 0, s, pred, gt, gt_channels, w, h, win,                 \
                           -ssim_weight * inv_ssim * grad_scale, dmaps, partials);                                   \
        hipLaunchKernelGGL(k_ssim_backward<W>, gb, dim3(192), 0, s, pred, gt, gt_channels, w, h, win, dmaps,         \
                           (1.0f - ssim_weight) * inv_l1 * grad_scale, v_pred, partials, gf.x * gf.y * 3,            \
                           1.0f - ssim_weight, ssim_weight, inv_l1, inv_ssim, loss);                                 \
    } while (0)
    switch (ssim_window) {
        case 3: BRUSH_SSIM(3); break;
        case 5: BRUSH_SSIM(5); break;
        case 7: BRUSH_SSIM(7); break;
        case 9: BRUSH_SSIM(9); break;
        case 11: BRUSH_SSIM(11); break;
        case 13: BRUSH_SSIM(13); break;
        default: BRUSH_SSIM(15); break;
    }
#undef BRUSH_SSIM
    BRUSH_HIP_CHECK(hipGetLastError());
    return BRUSH_OK;
}
