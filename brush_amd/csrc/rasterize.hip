// rasterize.hip — per-tile front-to-back alpha compositing and its backward pass.
//
// Replaces:
//   Rasterize           crates/brush-render/src/shaders/rasterize.wgsl:20-115
//   RasterizeBackwards  crates/brush-render/src/shaders/rasterize_backwards.wgsl:140-304
//
// gfx950 layout (both kernels): ONE wave64 per 16x16 tile, FOUR horizontally adjacent pixels
// per lane (lane l -> row l/4, columns 4*(l%4)..+3).  A single-wave workgroup needs no
// s_barrier and no LDS atomics; the tile's depth-sorted splat list is staged in LDS in batches
// of 64 records (one gathered 36-byte record per lane) and read back as wave-uniform
// broadcasts (2 x ds_read_b128 + 1 x ds_read_b32 per splat, shared by 4 pixel evaluations per
// lane).  Workgroups are dealt to XCDs round-robin, so block ids are remapped to give every
// XCD a contiguous band of tiles: neighbouring tiles gather the same splat records from one L2.
//
// Forward: identical arithmetic to the reference per pixel; the wave leaves the list as soon as
// all of its 256 pixels have saturated (the reference walks every batch, rasterize.wgsl:57-101;
// same result).
//
// Backward: replaces the reference's LDS gradient queue + nine software CAS loops per queued
// gradient (rasterize_backwards.wgsl:47-135,276-301).  Each lane first sums the 9 gradient
// components over its 4 pixels (fused into the FMAs), then ONE wave64 DPP reduction per
// component (row_shr 1/2/4/8 + row_bcast 15/31, pure VALU) leaves the tile total in lane 63,
// which parks it in an LDS row.  When a batch retires, the [64][9] block is flushed with
// hardware global_atomic_add_f32 in a shape where consecutive lanes hit consecutive components
// of one splat (contiguous 36-byte segments).  Splats that touch no pixel of the tile skip the
// reduction.
//
// Roofline: both kernels are fp32-VALU / v_exp_f32 bound, not HBM bound (256 pixel evaluations
// per 40-byte intersection record); DESIGN.md states both ceilings.
#include "internal.hpp"

namespace brush {
namespace {

constexpr uint32_t kBatch = kWave;  // 64 splats per LDS batch, one per lane
constexpr uint32_t kPix = 4;        // pixels per lane

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

constexpr uint32_t kTilesPerBlock = 4;  // 4 independent wave64s per 256-thread workgroup
constexpr uint32_t kRasterThreads = kTilesPerBlock * kWave;

// Every XCD (blocks b, b+8, b+16, ... share one) gets a contiguous band of tile ids; wave `wv` of
// block `bid` takes one tile.  The grid has a multiple of 8 blocks.
__device__ __forceinline__ uint32_t xcd_tile(uint32_t bid, uint32_t nblocks, uint32_t wv) {
    const uint32_t per = nblocks >> 3;
    return ((bid & 7u) * per + (bid >> 3)) * kTilesPerBlock + wv;
}
// The waves of a workgroup never exchange data: LDS hand-offs are wave-local, the LDS queue of a
// wave is in order, so a compiler-level barrier is all that is needed (no s_barrier).
__device__ __forceinline__ void wave_sync() { __builtin_amdgcn_wave_barrier(); }

template <bool RASTER_U32>
__global__ __launch_bounds__(kRasterThreads) void k_rasterize(uint32_t w, uint32_t h, uint32_t tbx, uint32_t num_tiles,
                                                     const uint32_t *__restrict__ gid_from_isect,
                                                     const uint32_t *__restrict__ tile_bins,
                                                     const float *__restrict__ projected,
                                                     void *__restrict__ out_img,
                                                     uint32_t *__restrict__ final_index) {
    __shared__ SplatLds lds_all[kTilesPerBlock];
    const uint32_t wv = threadIdx.x / kWave;
    SplatLds &lds = lds_all[wv];
    const uint32_t tile_id = xcd_tile(blockIdx.x, gridDim.x, wv);
    if (tile_id >= num_tiles) return;
    const uint32_t tile_x = tile_id % tbx, tile_y = tile_id / tbx;
    const uint32_t lane = threadIdx.x & (kWave - 1);
    const uint32_t px0 = tile_x * kTileWidth + (lane & 3u) * kPix;
    const uint32_t py = tile_y * kTileWidth + (lane >> 2);
    const float pcy = (float)py + 0.5f;  // rasterize.wgsl:32
    const float pcx0 = (float)px0 + 0.5f;

    bool done[kPix];
    float T[kPix], cr[kPix], cg[kPix], cb[kPix];
    uint32_t fin[kPix];
#pragma unroll
    for (uint32_t j = 0; j < kPix; j++) {
        done[j] = !(px0 + j < w && py < h);
        T[j] = 1.0f;
        cr[j] = cg[j] = cb[j] = 0.0f;
        fin[j] = 0;
    }

    const uint32_t r0 = tile_bins[tile_id * 2], r1 = tile_bins[tile_id * 2 + 1];
    for (uint32_t batch_start = r0; batch_start < r1; batch_start += kBatch) {
        if (__ballot(!(done[0] && done[1] && done[2] && done[3])) == 0ull) break;
        const uint32_t remaining = min(kBatch, r1 - batch_start);
        wave_sync();
        if (lane < remaining) {
            const uint32_t cg_id = gid_from_isect[batch_start + lane];
            stage_splat(lds, lane, projected + (size_t)cg_id * BRUSH_PROJECTED_FLOATS);
        }
        wave_sync();
        for (uint32_t t = 0; t < remaining; t++) {
            const float4 a = lds.a[t];
            const float4 b = lds.b[t];
            const float opac = lds.o[t];
            const float dy = a.y - pcy;
            const float cdy2 = b.x * dy * dy;
            const float bdy = a.w * dy;
            const float dx0 = a.x - pcx0;
#pragma unroll
            for (uint32_t j = 0; j < kPix; j++) {
                // Branch-free form of rasterize.wgsl:80-99 (selects, no exec-mask juggling).
                const float dx = dx0 - (float)j;
                const float sigma = 0.5f * (a.z * dx * dx + cdy2) + bdy * dx;
                const float vis = __expf(-sigma);
                const float alpha = fminf(0.999f, opac * vis);
                const bool hit = !done[j] && sigma >= 0.0f && alpha >= 1.0f / 255.0f;
                const float next_T = T[j] * (1.0f - alpha);
                const bool stop = hit && next_T <= 1e-4f;  // :88-91: stop without adding this entry
                const bool add = hit && !stop;
                const float fac = alpha * T[j];
                cr[j] = add ? __builtin_fmaf(b.y, fac, cr[j]) : cr[j];
                cg[j] = add ? __builtin_fmaf(b.z, fac, cg[j]) : cg[j];
                cb[j] = add ? __builtin_fmaf(b.w, fac, cb[j]) : cb[j];
                T[j] = add ? next_T : T[j];
                fin[j] = add ? batch_start + t : fin[j];
                done[j] = done[j] || stop;
            }
            if (__ballot(!(done[0] && done[1] && done[2] && done[3])) == 0ull) break;
        }
    }

    if (py < h) {
#pragma unroll
        for (uint32_t j = 0; j < kPix; j++) {
            if (px0 + j < w) {
                const size_t pix = (size_t)(px0 + j) + (size_t)py * w;
                const float al = 1.0f - T[j];
                if (RASTER_U32) {
                    // rasterize.wgsl:106-109
                    const uint32_t r8 = (uint32_t)fminf(fmaxf(cr[j] * 255.0f, 0.0f), 255.0f);
                    const uint32_t g8 = (uint32_t)fminf(fmaxf(cg[j] * 255.0f, 0.0f), 255.0f);
                    const uint32_t b8 = (uint32_t)fminf(fmaxf(cb[j] * 255.0f, 0.0f), 255.0f);
                    const uint32_t a8 = (uint32_t)fminf(fmaxf(al * 255.0f, 0.0f), 255.0f);
                    static_cast<uint32_t *>(out_img)[pix] = r8 | (g8 << 8) | (b8 << 16) | (a8 << 24);
                } else {
                    static_cast<float4 *>(out_img)[pix] = make_float4(cr[j], cg[j], cb[j], al);
                    final_index[pix] = fin[j];
                }
            }
        }
    }
}

// ---- backward -----------------------------------------------------------------------------

constexpr uint32_t kGradComps = 9;  // v_xy(2) v_conic(3) v_rgb(3) v_opac(1)

// Wave64 sum on the VALU with DPP (no LDS traffic, unlike __shfl_xor = ds_bpermute):
// inclusive scan inside each row of 16 (row_shr 1/2/4/8), then row_bcast:15 and row_bcast:31
// carry the row totals up.  The full sum is valid in LANE 63 only.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_add(float v) {
    const int moved = __builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, ROW_MASK, 0xf, true);
    return v + __int_as_float(moved);
}
// Row-of-16 inclusive scan: the row total is valid in lane 15 of each row.
__device__ __forceinline__ float row_sum_lane15(float v) {
    v = dpp_add<0x111, 0xf>(v);  // row_shr:1
    v = dpp_add<0x112, 0xf>(v);  // row_shr:2
    v = dpp_add<0x114, 0xf>(v);  // row_shr:4
    v = dpp_add<0x118, 0xf>(v);  // row_shr:8
    return v;
}
// Transposing pair reductions with the gfx950 lane-swap instructions: one swap + one add fold two
// registers into one in which half of the lanes carry the pair sums of `a`, the other half of `b`.
//   swap32: lanes 0-31 <- a[l] + a[l+32],   lanes 32-63 <- b[l-32] + b[l]
//   swap16: even rows  <- a[row] + a[row+1], odd rows   <- b[row-1] + b[row]
__device__ __forceinline__ float fold_swap32(float a, float b) {
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(a), __float_as_uint(b), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
__device__ __forceinline__ float fold_swap16(float a, float b) {
    const auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(a), __float_as_uint(b), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
__device__ __forceinline__ float wave_sum_lane63(float v) {
    v = dpp_add<0x111, 0xf>(v);  // row_shr:1
    v = dpp_add<0x112, 0xf>(v);  // row_shr:2
    v = dpp_add<0x114, 0xf>(v);  // row_shr:4
    v = dpp_add<0x118, 0xf>(v);  // row_shr:8
    v = dpp_add<0x142, 0xa>(v);  // row_bcast:15 -> rows 1,3
    v = dpp_add<0x143, 0xc>(v);  // row_bcast:31 -> rows 2,3
    return v;
}

__global__ __launch_bounds__(kRasterThreads) void k_rasterize_backward(
    uint32_t w, uint32_t h, uint32_t tbx, uint32_t num_tiles, const uint32_t *__restrict__ gid_from_isect,
    const uint32_t *__restrict__ tile_bins, const float *__restrict__ projected,
    const uint32_t *__restrict__ final_index, const float *__restrict__ out_img,
    const float *__restrict__ v_out, float *__restrict__ v_compact) {
    __shared__ SplatLds lds_all[kTilesPerBlock];
    __shared__ uint32_t lds_gid_all[kTilesPerBlock][kBatch];
    __shared__ float acc_all[kTilesPerBlock][kBatch][12];  // 9 used; 48-byte rows keep b128 stores aligned

    const uint32_t wv = threadIdx.x / kWave;
    SplatLds &lds = lds_all[wv];
    uint32_t *lds_gid = lds_gid_all[wv];
    float(*acc)[12] = acc_all[wv];
    const uint32_t tile_id = xcd_tile(blockIdx.x, gridDim.x, wv);
    if (tile_id >= num_tiles) return;
    const uint32_t r0 = tile_bins[tile_id * 2], r1 = tile_bins[tile_id * 2 + 1];
    if (r1 <= r0) return;

    const uint32_t tile_x = tile_id % tbx, tile_y = tile_id / tbx;
    const uint32_t lane = threadIdx.x & (kWave - 1);
    const uint32_t px0 = tile_x * kTileWidth + (lane & 3u) * kPix;
    const uint32_t py = tile_y * kTileWidth + (lane >> 2);
    const float pcy = (float)py + 0.5f;
    const float pcx0 = (float)px0 + 0.5f;

    bool inside[kPix];
    // Per-pixel state.  The reference's running colour `buffer` (rasterize_backwards.wgsl:253-257)
    // only ever appears dotted with the pixel's constant v_out.rgb, so the scalar
    // D = sum_j fac_j * (c_j . v_rgb) carries the same information; K = T_final * v_out.a.
    float T[kPix], K[kPix], D[kPix];
    float4 vo[kPix];
    uint32_t fin[kPix];
#pragma unroll
    for (uint32_t j = 0; j < kPix; j++) {
        inside[j] = px0 + j < w && py < h;
        float T_final = 1.0f;
        fin[j] = 0;
        vo[j] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (inside[j]) {
            const size_t pix = (size_t)(px0 + j) + (size_t)py * w;
            T_final = 1.0f - out_img[pix * 4 + 3];  // rasterize_backwards.wgsl:163
            fin[j] = final_index[pix];
            vo[j] = reinterpret_cast<const float4 *>(v_out)[pix];
        }
        T[j] = T_final;
        K[j] = T_final * vo[j].w;
        D[j] = 0.0f;
    }

    // Batches walk the list back to front (rasterize_backwards.wgsl:194-208).
    for (uint32_t batch_end = r1; batch_end > r0;) {
        const uint32_t remaining = min(kBatch, batch_end - r0);
        wave_sync();  // previous batch fully flushed
        if (lane < remaining) {
            const uint32_t cg_id = gid_from_isect[batch_end - 1u - lane];
            lds_gid[lane] = cg_id;
            stage_splat(lds, lane, projected + (size_t)cg_id * BRUSH_PROJECTED_FLOATS);
        }
        {
            float4 *row = reinterpret_cast<float4 *>(&acc[lane][0]);
            row[0] = row[1] = row[2] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
        wave_sync();

        for (uint32_t t = 0; t < remaining; t++) {
            const uint32_t isect_id = batch_end - 1u - t;
            const float4 a = lds.a[t];
            const float4 b = lds.b[t];
            const float opac = lds.o[t];
            const float dy = a.y - pcy;
            const float cdy2 = b.x * dy * dy;
            const float bdy = a.w * dy;
            const float cdy = b.x * dy;
            const float dx0 = a.x - pcx0;
            float g[kGradComps];
#pragma unroll
            for (uint32_t k = 0; k < kGradComps; k++) g[k] = 0.0f;
            // Pass 1 (cheap, branch-free): alpha and the contribution mask of the 4 pixels.
            float vis[kPix], alpha[kPix];
            bool m[kPix];
#pragma unroll
            for (uint32_t j = 0; j < kPix; j++) {
                const float dx = dx0 - (float)j;
                const float sigma = 0.5f * (a.z * dx * dx + cdy2) + bdy * dx;
                vis[j] = __expf(-sigma);
                alpha[j] = fminf(0.99f, opac * vis[j]);  // 0.99 here (rasterize_backwards.wgsl:239)
                m[j] = inside[j] && isect_id <= fin[j] && sigma >= 0.0f && alpha[j] >= 1.0f / 255.0f;
            }
            const bool any = __ballot(m[0] || m[1] || m[2] || m[3]) != 0ull;
            // Pass 2: rasterize_backwards.wgsl:244-271 with selects; a pixel column no lane of the
            // wave contributes to is skipped by a wave-uniform (scalar) branch.
#pragma unroll
            for (uint32_t j = 0; j < kPix; j++) {
                if (__ballot(m[j]) == 0ull) continue;
                const float dx = dx0 - (float)j;
                const float adx = a.z * dx;
                // v_rcp_f32 (1 ulp) + one Newton step; 1 - alpha >= 0.01 so this is always finite.
                const float om = 1.0f - alpha[j];
                float ra = __builtin_amdgcn_rcpf(om);
                ra = __builtin_fmaf(__builtin_fmaf(-om, ra, 1.0f), ra, ra);
                const float Tn = T[j] * ra;
                const float fac = alpha[j] * Tn;
                // v_alpha = (c*T - buffer*ra) . v_rgb + T_final*ra*v_a  (rasterize_backwards.wgsl:253-254)
                //         = T*(c . v_rgb) + ra*(K - D)
                const float cv = b.y * vo[j].x + b.z * vo[j].y + b.w * vo[j].z;
                const float v_alpha = __builtin_fmaf(Tn, cv, ra * (K[j] - D[j]));
                T[j] = m[j] ? Tn : T[j];
                D[j] = m[j] ? __builtin_fmaf(fac, cv, D[j]) : D[j];
                const float vis_m = m[j] ? vis[j] * v_alpha : 0.0f;  // v_opac term
                const float v_sigma = -opac * vis_m;                  // 0 when masked
                const float fac_m = m[j] ? fac : 0.0f;
                g[0] = __builtin_fmaf(v_sigma, adx + bdy, g[0]);
                g[1] = __builtin_fmaf(v_sigma, a.w * dx + cdy, g[1]);
                g[2] = __builtin_fmaf(0.5f * v_sigma, dx * dx, g[2]);
                g[3] = __builtin_fmaf(v_sigma, dx * dy, g[3]);
                g[4] = __builtin_fmaf(0.5f * v_sigma, dy * dy, g[4]);
                g[5] = __builtin_fmaf(fac_m, vo[j].x, g[5]);
                g[6] = __builtin_fmaf(fac_m, vo[j].y, g[6]);
                g[7] = __builtin_fmaf(fac_m, vo[j].z, g[7]);
                g[8] += vis_m;
            }
            if (any) {  // wave-uniform: all 64 lanes take part in the reduction
                // 8 components: two transposing folds (lane-swap + add), then a row-of-16 scan of
                // the two survivors; component 8 takes the plain 6-step DPP sum.  26 VALU ops.
                const float u0 = fold_swap32(g[0], g[1]), u1 = fold_swap32(g[2], g[3]);
                const float u2 = fold_swap32(g[4], g[5]), u3 = fold_swap32(g[6], g[7]);
                const float w0 = row_sum_lane15(fold_swap16(u0, u1));
                const float w1 = row_sum_lane15(fold_swap16(u2, u3));
                const float s8 = wave_sum_lane63(g[8]);
                if ((lane & 15u) == 15u) {
                    // row r: w0 holds component ((r&1)<<1 | r>>1), w1 the same + 4
                    const uint32_t r = lane >> 4;
                    const uint32_t i0 = ((r & 1u) << 1) | (r >> 1);
                    acc[t][i0] = w0;
                    acc[t][4 + i0] = w1;
                    if (lane == 63) acc[t][8] = s8;
                }
            }
        }
        wave_sync();
        // Flush: one hardware float atomic per (tile, splat, component); consecutive lanes take
        // consecutive components of one splat (MI355X_MICROARCH.md, global float atomics).
        for (uint32_t f = lane; f < remaining * kGradComps; f += kWave) {
            const uint32_t t = f / kGradComps, k = f - t * kGradComps;
            const float v = acc[t][k];
            if (v != 0.0f) unsafeAtomicAdd(&v_compact[(size_t)lds_gid[t] * kCompactStride + k], v);
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
    const dim3 grid(ceil_div(ceil_div(tiles, kTilesPerBlock), 8u) * 8u), block(kRasterThreads);
    if (raster_u32) {
        hipLaunchKernelGGL(k_rasterize<true>, grid, block, 0, s, w, h, tbx, tiles, compact_gid_from_isect, tile_bins,
                           projected, out_img, final_index);
    } else {
        hipLaunchKernelGGL(k_rasterize<false>, grid, block, 0, s, w, h, tbx, tiles, compact_gid_from_isect, tile_bins,
                           projected, out_img, final_index);
    }
    return hipGetLastError();
}

hipError_t launch_rasterize_backward(uint32_t w, uint32_t h, uint32_t tbx, uint32_t tby,
                                     const uint32_t *compact_gid_from_isect, const uint32_t *tile_bins,
                                     const float *projected, const uint32_t *final_index,
                                     const float *out_img, const float *v_out, float *v_compact,
                                     hipStream_t s) {
    const uint32_t tiles = tbx * tby;
    if (tiles == 0) return hipSuccess;
    const dim3 grid(ceil_div(ceil_div(tiles, kTilesPerBlock), 8u) * 8u), block(kRasterThreads);
    hipLaunchKernelGGL(k_rasterize_backward, grid, block, 0, s, w, h, tbx, tiles, compact_gid_from_isect, tile_bins,
                       projected, final_index, out_img, v_out, v_compact);
    return hipGetLastError();
}

}  // namespace brush
