// rasterize.hip — per-tile front-to-back alpha compositing and its backward pass.
//
// Replaces:
//   Rasterize           crates/brush-render/src/shaders/rasterize.wgsl:20-115
//   RasterizeBackwards  crates/brush-render/src/shaders/rasterize_backwards.wgsl:140-304
//
// Both kernels: one 256-thread workgroup (4 wave64) per 16x16 tile, one pixel per lane; wave w
// owns pixel rows 4w..4w+3.  The tile's depth-sorted splat list is staged in LDS in batches of
// 256 records as two float4 + one float (broadcast ds_read_b128 x2 + ds_read_b32 per splat,
// conflict-free because all lanes read one address).
//
// Forward adds a workgroup-wide early exit once every pixel has saturated (the reference walks
// all batches, rasterize.wgsl:57-101; results are identical).
//
// Backward replaces the reference's LDS gradient queue + 9 software CAS loops per queued
// gradient (rasterize_backwards.wgsl:47-135,276-301) by: wave64 DPP/shuffle reduction of the 9
// components, LDS float atomics across the 4 waves into a per-batch [256][9] accumulator, and
// ONE hardware global_atomic_add_f32 per (tile, splat, component) when the batch retires.
// Waves in which no pixel is touched by a splat skip its reduction entirely.
//
// Roofline: these two kernels are fp32-VALU / v_exp_f32 / LDS-broadcast bound, not HBM bound
// (256 pixel evaluations per 40-byte intersection record); DESIGN.md states both ceilings.
#include "internal.hpp"

namespace brush {
namespace {

constexpr uint32_t kBatch = kTileSize;  // 256 splats per LDS batch

struct SplatLds {
    float4 a[kBatch];  // xy.x, xy.y, conic.x, conic.y
    float4 b[kBatch];  // conic.z, r, g, b
    float o[kBatch];   // opacity
};

__device__ __forceinline__ void stage_splat(SplatLds &lds, uint32_t slot, const float *__restrict__ p) {
    lds.a[slot] = make_float4(p[0], p[1], p[2], p[3]);
    lds.b[slot] = make_float4(p[4], p[5], p[6], p[7]);
    lds.o[slot] = p[8];
}

template <bool RASTER_U32>
__global__ __launch_bounds__(kTileSize) void k_rasterize(uint32_t w, uint32_t h, uint32_t tbx,
                                                         const uint32_t *__restrict__ gid_from_isect,
                                                         const uint32_t *__restrict__ tile_bins,
                                                         const float *__restrict__ projected,
                                                         void *__restrict__ out_img,
                                                         uint32_t *__restrict__ final_index) {
    __shared__ SplatLds lds;
    const uint32_t tile_id = blockIdx.x;
    const uint32_t tile_x = tile_id % tbx, tile_y = tile_id / tbx;
    const uint32_t tid = threadIdx.x;
    const uint32_t px = tile_x * kTileWidth + (tid % kTileWidth);
    const uint32_t py = tile_y * kTileWidth + (tid / kTileWidth);
    const bool inside = px < w && py < h;
    const float pcx = (float)px + 0.5f, pcy = (float)py + 0.5f;  // rasterize.wgsl:32
    bool done = !inside;

    const uint32_t r0 = tile_bins[tile_id * 2], r1 = tile_bins[tile_id * 2 + 1];
    float T = 1.0f;
    float cr = 0.0f, cg = 0.0f, cb = 0.0f;
    uint32_t final_idx = 0;

    for (uint32_t batch_start = r0; batch_start < r1; batch_start += kBatch) {
        // Workgroup-wide early exit; also the barrier that protects the LDS batch.
        if (__syncthreads_count(!done) == 0) break;
        const uint32_t remaining = min(kBatch, r1 - batch_start);
        if (tid < remaining) {
            const uint32_t cg_id = gid_from_isect[batch_start + tid];
            stage_splat(lds, tid, projected + (size_t)cg_id * BRUSH_PROJECTED_FLOATS);
        }
        __syncthreads();
        if (!done) {
            for (uint32_t t = 0; t < remaining; t++) {
                const float4 a = lds.a[t];
                const float4 b = lds.b[t];
                const float opac = lds.o[t];
                const float dx = a.x - pcx, dy = a.y - pcy;
                const float sigma = 0.5f * (a.z * dx * dx + b.x * dy * dy) + a.w * dx * dy;
                const float vis = __expf(-sigma);
                const float alpha = fminf(0.999f, opac * vis);
                if (sigma >= 0.0f && alpha >= 1.0f / 255.0f) {
                    const float next_T = T * (1.0f - alpha);
                    if (next_T <= 1e-4f) {
                        done = true;
                        break;
                    }
                    const float fac = alpha * T;
                    cr += b.y * fac;
                    cg += b.z * fac;
                    cb += b.w * fac;
                    T = next_T;
                    final_idx = batch_start + t;
                }
            }
        }
    }

    if (inside) {
        const size_t pix = (size_t)px + (size_t)py * w;
        const float a = 1.0f - T;
        if (RASTER_U32) {
            // rasterize.wgsl:106-109
            const uint32_t r8 = (uint32_t)fminf(fmaxf(cr * 255.0f, 0.0f), 255.0f);
            const uint32_t g8 = (uint32_t)fminf(fmaxf(cg * 255.0f, 0.0f), 255.0f);
            const uint32_t b8 = (uint32_t)fminf(fmaxf(cb * 255.0f, 0.0f), 255.0f);
            const uint32_t a8 = (uint32_t)fminf(fmaxf(a * 255.0f, 0.0f), 255.0f);
            static_cast<uint32_t *>(out_img)[pix] = r8 | (g8 << 8) | (b8 << 16) | (a8 << 24);
        } else {
            static_cast<float4 *>(out_img)[pix] = make_float4(cr, cg, cb, a);
            final_index[pix] = final_idx;
        }
    }
}

// ---- backward -----------------------------------------------------------------------------

constexpr uint32_t kGradComps = 9;  // v_xy(2) v_conic(3) v_rgb(3) v_opac(1)

__global__ __launch_bounds__(kTileSize) void k_rasterize_backward(
    uint32_t w, uint32_t h, uint32_t tbx, const uint32_t *__restrict__ gid_from_isect,
    const uint32_t *__restrict__ tile_bins, const float *__restrict__ projected,
    const uint32_t *__restrict__ final_index, const float *__restrict__ out_img,
    const float *__restrict__ v_out, float *__restrict__ v_xy, float *__restrict__ v_conics,
    float *__restrict__ v_colors) {
    __shared__ SplatLds lds;
    __shared__ uint32_t lds_gid[kBatch];
    __shared__ float acc[kBatch][kGradComps];

    const uint32_t tile_id = blockIdx.x;
    const uint32_t r0 = tile_bins[tile_id * 2], r1 = tile_bins[tile_id * 2 + 1];
    if (r1 <= r0) return;  // uniform per workgroup

    const uint32_t tile_x = tile_id % tbx, tile_y = tile_id / tbx;
    const uint32_t tid = threadIdx.x;
    const uint32_t px = tile_x * kTileWidth + (tid % kTileWidth);
    const uint32_t py = tile_y * kTileWidth + (tid / kTileWidth);
    const bool inside = px < w && py < h;
    const float pcx = (float)px + 0.5f, pcy = (float)py + 0.5f;
    const size_t pix = (size_t)px + (size_t)py * w;

    float T_final = 1.0f;
    uint32_t final_isect = 0;
    float4 vo = make_float4(0.f, 0.f, 0.f, 0.f);
    if (inside) {
        T_final = 1.0f - out_img[pix * 4 + 3];  // rasterize_backwards.wgsl:163
        final_isect = final_index[pix];
        vo = reinterpret_cast<const float4 *>(v_out)[pix];
    }
    float T = T_final;
    float bufr = 0.f, bufg = 0.f, bufb = 0.f;

    // Batches walk the list back to front (rasterize_backwards.wgsl:194-208).
    for (uint32_t batch_end = r1; batch_end > r0;) {
        const uint32_t remaining = min(kBatch, batch_end - r0);
        __syncthreads();  // previous batch fully consumed (LDS splats + acc flushed)
        if (tid < remaining) {
            const uint32_t cg_id = gid_from_isect[batch_end - 1u - tid];
            lds_gid[tid] = cg_id;
            stage_splat(lds, tid, projected + (size_t)cg_id * BRUSH_PROJECTED_FLOATS);
        }
        for (uint32_t i = tid; i < kBatch * kGradComps; i += kTileSize) (&acc[0][0])[i] = 0.0f;
        __syncthreads();

        for (uint32_t t = 0; t < remaining; t++) {
            const uint32_t isect_id = batch_end - 1u - t;
            float g[kGradComps];
#pragma unroll
            for (uint32_t k = 0; k < kGradComps; k++) g[k] = 0.0f;
            bool active = false;
            if (inside && isect_id <= final_isect) {
                const float4 a = lds.a[t];
                const float4 b = lds.b[t];
                const float opac = lds.o[t];
                const float dx = a.x - pcx, dy = a.y - pcy;
                const float sigma = 0.5f * (a.z * dx * dx + b.x * dy * dy) + a.w * dx * dy;
                const float vis = __expf(-sigma);
                const float alpha = fminf(0.99f, opac * vis);  // 0.99 here (rasterize_backwards.wgsl:239)
                if (sigma >= 0.0f && alpha >= 1.0f / 255.0f) {
                    active = true;
                    const float ra = 1.0f / (1.0f - alpha);
                    T *= ra;
                    const float fac = alpha * T;
                    float v_alpha = (b.y * T - bufr * ra) * vo.x + (b.z * T - bufg * ra) * vo.y +
                                    (b.w * T - bufb * ra) * vo.z;
                    v_alpha += T_final * ra * vo.w;
                    bufr += b.y * fac;
                    bufg += b.z * fac;
                    bufb += b.w * fac;
                    const float v_sigma = -opac * vis * v_alpha;
                    g[0] = v_sigma * (a.z * dx + a.w * dy);
                    g[1] = v_sigma * (a.w * dx + b.x * dy);
                    g[2] = 0.5f * v_sigma * dx * dx;
                    g[3] = v_sigma * dx * dy;
                    g[4] = 0.5f * v_sigma * dy * dy;
                    g[5] = fac * vo.x;
                    g[6] = fac * vo.y;
                    g[7] = fac * vo.z;
                    g[8] = vis * v_alpha;
                }
            }
            if (__ballot(active) != 0ull) {  // wave-uniform
#pragma unroll
                for (uint32_t k = 0; k < kGradComps; k++) g[k] = wave_sum(g[k]);
                if (lane_id() == 0) {
#pragma unroll
                    for (uint32_t k = 0; k < kGradComps; k++) atomicAdd(&acc[t][k], g[k]);
                }
            }
        }
        __syncthreads();
        // Flush: one hardware float atomic per (tile, splat, component).
        if (tid < remaining) {
            const uint32_t cg_id = lds_gid[tid];
            float s[kGradComps];
            bool any = false;
#pragma unroll
            for (uint32_t k = 0; k < kGradComps; k++) {
                s[k] = acc[tid][k];
                any |= s[k] != 0.0f;
            }
            if (any) {
                unsafeAtomicAdd(&v_xy[(size_t)cg_id * 2 + 0], s[0]);
                unsafeAtomicAdd(&v_xy[(size_t)cg_id * 2 + 1], s[1]);
                unsafeAtomicAdd(&v_conics[(size_t)cg_id * 3 + 0], s[2]);
                unsafeAtomicAdd(&v_conics[(size_t)cg_id * 3 + 1], s[3]);
                unsafeAtomicAdd(&v_conics[(size_t)cg_id * 3 + 2], s[4]);
                unsafeAtomicAdd(&v_colors[(size_t)cg_id * 4 + 0], s[5]);
                unsafeAtomicAdd(&v_colors[(size_t)cg_id * 4 + 1], s[6]);
                unsafeAtomicAdd(&v_colors[(size_t)cg_id * 4 + 2], s[7]);
                unsafeAtomicAdd(&v_colors[(size_t)cg_id * 4 + 3], s[8]);
            }
        }
        batch_end -= remaining;
    }
}

}  // namespace

hipError_t launch_rasterize(uint32_t w, uint32_t h, uint32_t tbx, uint32_t tby,
                            const uint32_t *compact_gid_from_isect, const uint32_t *tile_bins,
                            const float *projected, int raster_u32, void *out_img, uint32_t *final_index,
                            hipStream_t s) {
    const uint32_t tiles = tbx * tby;
    if (tiles == 0) return hipSuccess;
    if (raster_u32) {
        hipLaunchKernelGGL(k_rasterize<true>, dim3(tiles), dim3(kTileSize), 0, s, w, h, tbx, compact_gid_from_isect,
                           tile_bins, projected, out_img, final_index);
    } else {
        hipLaunchKernelGGL(k_rasterize<false>, dim3(tiles), dim3(kTileSize), 0, s, w, h, tbx,
                           compact_gid_from_isect, tile_bins, projected, out_img, final_index);
    }
    return hipGetLastError();
}

hipError_t launch_rasterize_backward(uint32_t w, uint32_t h, uint32_t tbx, uint32_t tby,
                                     const uint32_t *compact_gid_from_isect, const uint32_t *tile_bins,
                                     const float *projected, const uint32_t *final_index,
                                     const float *out_img, const float *v_out, float *v_xy_local,
                                     float *v_conics, float *v_colors, hipStream_t s) {
    const uint32_t tiles = tbx * tby;
    if (tiles == 0) return hipSuccess;
    hipLaunchKernelGGL(k_rasterize_backward, dim3(tiles), dim3(kTileSize), 0, s, w, h, tbx, compact_gid_from_isect,
                       tile_bins, projected, final_index, out_img, v_out, v_xy_local, v_conics, v_colors);
    return hipGetLastError();
}

}  // namespace brush
