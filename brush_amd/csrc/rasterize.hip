// rasterize.hip — per-tile front-to-back alpha compositing and its backward pass.
//
// Replaces:
//   Rasterize           crates/brush-render/src/shaders/rasterize.wgsl:20-115
//   RasterizeBackwards  crates/brush-render/src/shaders/rasterize_backwards.wgsl:140-304
//
// gfx950 layout: pixels of an 8x8 QUADRANT of a 16x16 tile map to the 64 lanes of a wave (lane = x + 8 y), the tile's
// depth-sorted splat list is staged in LDS in batches of 64 records (one gathered 36-byte record per lane) and read back
// as wave-uniform broadcasts, and a (record, quadrant) pair whose alpha provably stays below 1/255 on the whole quadrant
// is skipped by a scalar branch (see "footprint-aware kernels" below).  Workgroups are dealt to XCDs round-robin, so
// block ids are remapped to give every XCD a contiguous band of tiles: neighbouring tiles gather the same splat records
// from one L2.  Waves of a workgroup never exchange data: no s_barrier, no LDS atomics.
//
// Forward: one wave per quadrant (4 waves = 1 tile per workgroup); identical arithmetic to the reference per pixel; the
// wave leaves the list as soon as all of its pixels have saturated (the reference walks every batch,
// rasterize.wgsl:57-101; same result).
//
// Backward: replaces the reference's LDS gradient queue + nine software CAS loops per queued gradient
// (rasterize_backwards.wgsl:47-135,276-301).  One wave per tile, one pixel per lane PER QUADRANT: a lane sums the 9
// gradient components over its quadrants in registers; the 64:1 sums are TRANSPOSED THROUGH LDS (the wave stores its
// partials as rows of 64 words, two lanes per row add half a row each with plain v_add_f32, three records per pass: see
// kStageRecs) because a cross-lane VALU add costs 6 SIMD cycles beside this kernel's arithmetic and a plain one 2.7.  The
// reducing lanes apply the per-record factors and flush with hardware global_atomic_add_f32, consecutive lanes on
// consecutive components of one splat's 64-byte compact row: the L2 executes float atomics line by line.  Records that
// touch no pixel of the tile skip reduction and flush.  (Deterministic mode keeps the round-2 form: one transposing
// wave64 reduction per record with v_permlane32/16_swap + DPP row sums, rows stored per intersection.)
//
// Roofline: both kernels are bound by fp32 VALU issue (and the backward by the L2's atomic rate), not by HBM;
// DESIGN.md states the ceilings and the measurements.
#include <stdlib.h>

#include "internal.hpp"
#include "trace.hpp"

namespace brush {
namespace {

constexpr uint32_t kBatch = kWave;  // 64 splats per LDS batch, one per lane
constexpr float kNegLog2e = -1.44269504088896341f;  // exp(-s) = exp2(kNegLog2e * s)

constexpr uint32_t kTilesPerBlock = 4;  // 4 independent wave64s per 256-thread workgroup
constexpr uint32_t kRasterThreads = kTilesPerBlock * kWave;

// The waves of a workgroup never exchange data: LDS hand-offs are wave-local, the LDS queue of a
// wave is in order, so a compiler-level barrier is all that is needed (no s_barrier).
__device__ __forceinline__ void wave_sync() { __builtin_amdgcn_wave_barrier(); }

// ---- footprint-aware kernels -----------------------------------------------------------------
//
// Measured on the headline scene (profiles/r02a_footprint_s1.json): only 26 % of the 256 pixel
// evaluations of a (tile, splat) record pass `sigma >= 0 && alpha >= 1/255`; at 8x8 granularity a
// record touches 2.1 of the tile's 4 quadrants on average.  An evaluation that fails the test
// changes nothing (rasterize.wgsl:80-87, rasterize_backwards.wgsl:229-242), so whole quadrants a
// splat provably cannot reach are skipped with wave-uniform (scalar) control flow:
//   * quad_may_pass(): EXACT minimum of the splat's quadratic form over the box of the quadrant's
//     pixel centres (for a positive-definite conic the constrained minimiser lies on the line through
//     the box face nearest to the mean in x or in y, see the derivation at the function), turned
//     into an upper bound of alpha with a slack far above the rounding error of the per-pixel
//     arithmetic.  A quadrant is skipped only when that bound is below 0.99/255; anything not
//     provably positive definite / finite is never skipped.
//   * one lane evaluates the bound for one staged record, a ballot gives the 64-bit hit mask of the
//     batch, and the compositing loop walks its set bits on the scalar unit.
// Lane mask of a predicate straight from the compare (HIP's __ballot() converts the bool to an int and
// compares it again: two VALU instructions per call).
__device__ __forceinline__ uint64_t ballot64(bool p) { return __builtin_amdgcn_ballot_w64(p); }

constexpr float kHalfNegLog2e = -0.72134752044448170f;  // 2 sigma -> exp2 argument (backward)

// Largest alpha the splat can reach at any point of the box [bx, bx+7] x [by, by+7] (pixel centres
// of one 8x8 quadrant) >= 1/255 ?  d = mean - pixel ranges over [dxl,dxh] x [dyl,dyh]; q(d) =
// 0.5 (a dx^2 + c dy^2) + b dx dy is convex with its minimum 0 at d = 0.  If the box does not contain 0
// the minimiser d* sits on the boundary, and (KKT + positive definiteness) at least one coordinate
// is at the bound NEAREST to 0 of an axis whose range excludes 0: a minimiser on a far face with
// the other coordinate free would need dx (a - b^2/c) <= 0, and both coordinates on far faces would
// need q(d*) <= 0.  So min q = min( min_dy q(ex, dy), min_dx q(dx, ey) ) with ex, ey the clamps of 0
// into the ranges; each 1-D problem is a clamped parabola vertex.  (If a range contains 0 its line
// runs through the box: a feasible point, so it can only raise that candidate, never the minimum.)
__device__ __forceinline__ bool quad_may_pass(float mx, float my, float ca, float cb, float cc, float opac,
                                              float bx, float by) {
    const float dxl = mx - (bx + 7.0f), dxh = mx - bx;
    const float dyl = my - (by + 7.0f), dyh = my - by;
    const float ex = fminf(fmaxf(0.0f, dxl), dxh), ey = fminf(fmaxf(0.0f, dyl), dyh);
    // parabola vertices with v_rcp_f32 (1 ulp): q is stationary there, so the error is second order
    const float y1 = fminf(fmaxf(-cb * ex * __builtin_amdgcn_rcpf(cc), dyl), dyh);
    const float x2 = fminf(fmaxf(-cb * ey * __builtin_amdgcn_rcpf(ca), dxl), dxh);
    const float s1 = 0.5f * (ca * ex * ex + cc * y1 * y1), c1 = cb * ex * y1;
    const float s2 = 0.5f * (ca * x2 * x2 + cc * ey * ey), c2 = cb * x2 * ey;
    const float qmin = fminf(s1 + c1, s2 + c2);
    // slack: 0.02 absolute plus 4e-6 of the magnitude of the terms (f32 rounding of the per-pixel
    // evaluation is ~1e-7 of the same terms)
    const float slack = 0.02f + 4e-6f * fmaxf(s1 + fabsf(c1), s2 + fabsf(c2));
    const float amax = opac * __builtin_amdgcn_exp2f((qmin - slack) * kNegLog2e);
    const bool pd = ca > 0.0f && cc > 0.0f && ca * cc > cb * cb;
    return !pd || !(amax < 0.99f / 255.0f);  // NaN anywhere -> keep
}

// v_min_f32 without the canonicalising v_max the IEEE-mode lowering of fminf() puts in front of it.
__device__ __forceinline__ float vmin(float a, float b) {
    float r;
    asm("v_min_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

// One staged record of the footprint-aware kernels: 48 bytes, read back as wave-uniform broadcasts.
struct QuadRec {
    float4 a;  // mean.x, mean.y, conic.x, conic.y
    float4 b;  // conic.z, r, g, b
    float4 c;  // opacity, -, -, -
};

// Forward: ONE wave64 per 8x8 quadrant, one pixel per lane; the four waves of a workgroup are the four
// quadrants of one tile (they gather the same records, so three of the four gathers hit L1/L2).
// A pixel that has saturated gets a NaN pixel centre: every later `power <= 0` test fails for it, so the
// per-record path carries no separate "live" predicate.
template <bool RASTER_U32>
__global__ __launch_bounds__(kRasterThreads) void k_rasterize_quad(
    uint32_t w, uint32_t h, uint32_t tbx, uint32_t num_tiles, const uint32_t *__restrict__ gid_from_isect,
    uint32_t *__restrict__ tile_bins, const uint32_t *__restrict__ bin_edges, const float *__restrict__ projected,
    void *__restrict__ out_img, uint32_t *__restrict__ final_index, uint32_t u32_pitch,
    float4 *__restrict__ zero_rows, const uint32_t *__restrict__ num_visible, uint32_t n_splats) {
    __shared__ QuadRec lds_all[kTilesPerBlock][kBatch];
    BRUSH_KTRACE(kTrRasterize, 0);
    if (zero_rows) {
        // BrushAux::bwd_accum: the backward's compact-order accumulator rows of this render, zeroed here so that the
        // backward needs no zero-fill launch (fire-and-forget stores beside a kernel that is bound by VALU issue)
        const uint32_t words = min(*num_visible, n_splats) * kCompactVec;
        for (uint32_t i = blockIdx.x * kRasterThreads + threadIdx.x; i < words; i += gridDim.x * kRasterThreads)
            zero_rows[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    const uint32_t q = __builtin_amdgcn_readfirstlane(threadIdx.x / kWave);
    QuadRec *lds = lds_all[q];
    const uint32_t tile_id = (blockIdx.x & 7u) * (gridDim.x >> 3) + (blockIdx.x >> 3);  // XCD-contiguous bands
    if (tile_id >= num_tiles) return;
    const uint32_t lane = threadIdx.x & (kWave - 1);
    const uint32_t qx0 = (tile_id % tbx) * kTileWidth + (q & 1u) * 8u, qy0 = (tile_id / tbx) * kTileWidth + (q >> 1) * 8u;
    if (qx0 >= w || qy0 >= h) return;  // quadrant entirely outside a ragged frame
    const uint32_t px = qx0 + (lane & 7u), py = qy0 + (lane >> 3);
    const float bx = (float)qx0 + 0.5f, by = (float)qy0 + 0.5f;
    const bool inside = px < w && py < h;
    const float pcy = (float)py + 0.5f;  // rasterize.wgsl:32
    float pcx = inside ? (float)px + 0.5f : __builtin_nanf("");

    const uint64_t inside_mask = ballot64(inside);
    uint64_t live = inside_mask;
    uint32_t walked = 0;
    float T = 1.0f, cr = 0.0f, cg = 0.0f, cb_ = 0.0f;
    uint32_t fin = 0;
    uint32_t r0, r1;
    if (bin_edges) {
        // GetTileBinEdges (get_tile_bin_edges.wgsl:15-42) without its launch: the tile sort's last pass left
        // (max ~start, max end) of this tile's run; (0, 0) = the tile id does not occur.  Quadrant 0 (always inside
        // the frame) publishes the decoded pair for the aux and the backward.
        const uint32_t ns = bin_edges[tile_id * 2];
        r1 = bin_edges[tile_id * 2 + 1];
        r0 = r1 ? ~ns : 0u;
        if (q == 0u && lane == 0u) {
            tile_bins[tile_id * 2] = r0;
            tile_bins[tile_id * 2 + 1] = r1;
        }
    } else {
        r0 = tile_bins[tile_id * 2], r1 = tile_bins[tile_id * 2 + 1];
    }
    for (uint32_t batch_start = r0; batch_start < r1 && live != 0ull; batch_start += kBatch) {
        const uint32_t remaining = min(kBatch, r1 - batch_start);
        bool hit = false;
        float rec[9];
        if (lane < remaining) {
            const float *p = projected + (size_t)gid_from_isect[batch_start + lane] * BRUSH_PROJECTED_FLOATS;
#pragma unroll
            for (int k = 0; k < 9; k++) rec[k] = p[k];
            hit = quad_may_pass(rec[0], rec[1], rec[2], rec[3], rec[4], rec[8], bx, by);
        }
        uint64_t mask = ballot64(hit);
        if (mask == 0ull) continue;
        wave_sync();  // the previous batch's broadcasts are done (LDS is in order per wave)
        if (hit) {
            lds[lane].a = make_float4(rec[0], rec[1], rec[2], rec[3]);
            lds[lane].b = make_float4(rec[4], rec[5], rec[6], rec[7]);
            lds[lane].c = make_float4(rec[8], 0.f, 0.f, 0.f);
        }
        wave_sync();
        while (mask != 0ull) {
            const uint32_t t = (uint32_t)__builtin_ctzll(mask);
            mask &= mask - 1ull;
            const float4 a = lds[t].a;
            const float4 b = lds[t].b;
            const float opac = lds[t].c.x;
            // rasterize.wgsl:80-99
            const float dx = a.x - pcx, dy = a.y - pcy;
            // The reference's association, 0.5 (a dx^2 + c dy^2) + b dx dy: folding -log2(e)/2 into the conic saves
            // three instructions but moves alpha by ~1e-5 relative on correlated conics, enough to flip
            // `alpha >= 1/255` outside the oracle's guard band on a 20 M-splat frame (measured: 3.7e-4 pixel error).
            const float sigma = fmaf(0.5f, fmaf(a.z * dx, dx, (b.x * dy) * dy), (a.w * dy) * dx);
            const float power = sigma * kNegLog2e;
            const float alpha_u = opac * __builtin_amdgcn_exp2f(power);
            if (power <= 0.0f && alpha_u >= 1.0f / 255.0f) {  // never true for a saturated pixel (NaN)
                const float alpha = vmin(0.999f, alpha_u);
                const float next_T = T * (1.0f - alpha);
                if (next_T <= 1e-4f) {  // :88-91: stop WITHOUT adding this entry
                    pcx = __builtin_nanf("");
                } else {
                    const float fac = alpha * T;
                    cr = fmaf(b.y, fac, cr);
                    cg = fmaf(b.z, fac, cg);
                    cb_ = fmaf(b.w, fac, cb_);
                    T = next_T;
                    fin = batch_start + t;
                }
            }
            // all pixels saturated?  one compare every 4th record (the lane mask comes straight out of it)
            if ((++walked & 3u) == 0u) {
                live = ~__builtin_amdgcn_fcmp(pcx, pcx, 8 /* FCMP_UNO */) & inside_mask;
                if (live == 0ull) break;
            }
        }
    }
    if (inside) {
        const float al = 1.0f - T;
        if (RASTER_U32) {
            // rasterize.wgsl:106-109; rows `u32_pitch` pixels apart (burn_texture.rs:17-26)
            const uint32_t r8 = (uint32_t)fminf(fmaxf(cr * 255.0f, 0.0f), 255.0f);
            const uint32_t g8 = (uint32_t)fminf(fmaxf(cg * 255.0f, 0.0f), 255.0f);
            const uint32_t b8 = (uint32_t)fminf(fmaxf(cb_ * 255.0f, 0.0f), 255.0f);
            const uint32_t a8 = (uint32_t)fminf(fmaxf(al * 255.0f, 0.0f), 255.0f);
            static_cast<uint32_t *>(out_img)[(size_t)px + (size_t)py * u32_pitch] = r8 | (g8 << 8) | (b8 << 16) | (a8 << 24);
        } else {
            const size_t pix = (size_t)px + (size_t)py * w;
            static_cast<float4 *>(out_img)[pix] = make_float4(cr, cg, cb_, al);
            final_index[pix] = fin;
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

// development / test-only hooks (identity in the product build): dev_vva, dev_flush, dev_skip_reduce, BRUSH_DEV_BWD_TRACE
#define BRUSH_DEV_SECTION 1
#include "rasterize_dev.inc"

// Staged records of the backward, one array per field (the forward's QuadRec pads the opacity to 16 bytes).
struct BwdRecs {
    float4 a[kBatch];    // mean.x, mean.y, conic.x, conic.y
    float4 b[kBatch];    // conic.z, r, g, b
    float opac[kBatch];
};
// Default-mode reduction of the per-record lane partials.  Cross-lane VALU adds (DPP, lane swaps) cost 6 SIMD cycles
// each beside this kernel's arithmetic and a plain v_add_f32 2.7 (tools/ubench/valu_rate.hip), so the 64:1 sums are
// TRANSPOSED through LDS instead: the wave stores the 9 partials of every lane as 9 rows of 64 words (plain LDS stores,
// not VALU work), and once kStageRecs records wait, 2 lanes per row read half a row each (8 ds_read_b128) and add it up
// with plain adds: 31 adds + one lane swap per kStageRecs records instead of 26 cross-lane adds per record.  Rows start
// kRowWords apart so that the 8 lanes the LDS serves per cycle read 8 different groups of 4 banks.
constexpr uint32_t kStageRecs = 3;
constexpr uint32_t kRowWords = 68;
constexpr uint32_t kStageRows = kStageRecs * kGradComps;
static_assert(kStageRows <= 32, "one row per lane pair");

// ---- zero-fill in passing (ZeroFill, internal.hpp) -------------------------------------------------------------------
// A wave owes `quota` consecutive KiB blocks of the launch-wide block sequence (the dense gradient arrays laid end to
// end); the cursor lives in SGPRs, one block is one fire-and-forget 16-byte store per lane.  Only the last block of an
// array looks at chunk counts (partial block, up to three trailing floats).
struct FillCursor {
    float4 *ptr;              // the next block of the current array
    uint32_t block;           // its index in the block sequence
    uint32_t seg, seg_left;   // current array; blocks left in it, this one included
    uint32_t quota;           // blocks this wave still owes
};
__device__ __forceinline__ void fill_seek(const ZeroFill &zf, FillCursor &c) {
    uint32_t seg = 0;
#pragma unroll
    for (uint32_t i = 1; i < kFillSegs; i++) seg = c.block >= zf.first_block[i] ? i : seg;  // empty arrays are passed over
    c.seg = seg;
    c.seg_left = zf.first_block[seg + 1] - c.block;
    c.ptr = reinterpret_cast<float4 *>(zf.base[seg]) + (size_t)(c.block - zf.first_block[seg]) * kWave;
}
__device__ __forceinline__ void fill_step(const ZeroFill &zf, FillCursor &c, uint32_t lane) {
    const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
    if (c.seg_left > 1u) {
        // streaming store: written once, far larger than the L2s (measured against ordinary stores at 1 M splats:
        // compositing backward 126 vs 133 us, the VJP kernel behind it 25 vs 34 us)
        typedef float v4f __attribute__((ext_vector_type(4)));
        const v4f nz = {0.f, 0.f, 0.f, 0.f};
        __builtin_nontemporal_store(nz, reinterpret_cast<v4f *>(c.ptr + lane));
    } else {
        const uint32_t chunk = (zf.first_block[c.seg + 1] - zf.first_block[c.seg] - 1u) * kWave + lane;
        const uint32_t full = zf.full[c.seg];
        if (chunk < full) {
            c.ptr[lane] = z4;
        } else if (chunk == full) {
            float *t = reinterpret_cast<float *>(c.ptr + lane);
            for (uint32_t d = 0; d < zf.tail[c.seg]; d++) t[d] = 0.0f;
        }
    }
    c.ptr += kWave;
    c.block++;
    c.quota--;
    if (--c.seg_left == 0u && c.quota != 0u) fill_seek(zf, c);
}

// Footprint-aware backward.  A wave owns NQ quadrants of one tile (NQ = 4: one wave per tile, NQ = 2:
// upper / lower half, NQ = 1: one quadrant), one pixel per lane PER QUADRANT, so a lane's gradient
// contributions of all its quadrants are summed in registers and the 9-component wave reduction runs
// once per (wave, record) exactly as before; quadrants the record cannot reach (quad_may_pass) are
// skipped by scalar branches.  Under the per-pixel `if` the updates are plain (exec-masked) moves,
// no selects.  sigma is evaluated as 0.5 (dx gx + dy gy) with gx = a dx + b dy, gy = b dx + c dy,
// which are the v_xy factors of rasterize_backwards.wgsl:260-263 as well.
//
// DET (deterministic mode, NQ = 4 only: one wave per tile, so every intersection has exactly one producer):
// instead of adding to the splat's compact row with float atomics, the wave STORES one 64-byte row per
// intersection, [9 sums | compact gid | 0 ...], at the position the intersection had before the tile sort
// (`unsorted_pos`, grouped by splat); intersections it does not walk get zero rows.  k_sum_isect_rows then adds a
// splat's rows in that fixed order.
template <uint32_t NQ, bool DET, uint32_t TPB>
__global__ __launch_bounds__(TPB * kWave) void k_rasterize_backward_quad(
    uint32_t w, uint32_t h, uint32_t tbx, uint32_t num_tiles, const uint32_t *__restrict__ gid_from_isect,
    const uint32_t *__restrict__ tile_bins, const float *__restrict__ projected,
    const uint32_t *__restrict__ final_index, const float *__restrict__ out_img,
    const float *__restrict__ v_out, float *__restrict__ v_compact, const uint32_t *__restrict__ unsorted_pos,
    float *__restrict__ rows, const ZeroFill zf) {
    static_assert(!DET || NQ == 4, "deterministic mode: one wave per tile");
    __shared__ uint32_t lds_pos_all[DET ? TPB : 1][kBatch];
    __shared__ BwdRecs lds_all[TPB];
    __shared__ uint32_t lds_gid_all[TPB][kBatch];
    __shared__ float acc_all[DET ? TPB : 1][kBatch][12];  // DET: 9 used; 48-byte rows keep b128 stores aligned
    __shared__ float stage_all[DET ? 1 : TPB][kStageRows * kRowWords];
    constexpr uint32_t kWavesPerTile = 4u / NQ;
    BRUSH_KTRACE(kTrRasterizeBwd, 0);

    const uint32_t wv = __builtin_amdgcn_readfirstlane(threadIdx.x / kWave);
    BwdRecs &lds = lds_all[wv];
    uint32_t *lds_gid = lds_gid_all[wv];
    uint32_t *lds_pos = lds_pos_all[DET ? wv : 0];
    float(*acc)[12] = acc_all[DET ? wv : 0];
    float *stage = stage_all[DET ? 0 : wv];
    const uint32_t unit = ((blockIdx.x & 7u) * (gridDim.x >> 3) + (blockIdx.x >> 3)) * TPB + wv;  // XCD-contiguous
    const uint32_t tile_id = unit / kWavesPerTile, sub = unit % kWavesPerTile;
    const uint32_t lane = threadIdx.x & (kWave - 1);
    // Zero-fill in passing: this wave's share of the block sequence (every launched wave has one, tile or not)
    FillCursor fc;
    fc.quota = 0u;
    if (zf.active()) {
        const uint32_t total = zf.first_block[kFillSegs], per = ceil_div(total, gridDim.x * TPB);
        fc.block = min(unit * per, total);
        fc.quota = min(per, total - fc.block);
        if (fc.quota != 0u) fill_seek(zf, fc);
    }
    auto fill_rest = [&]() {
        while (fc.quota != 0u) fill_step(zf, fc, lane);
    };
    if (tile_id >= num_tiles) return fill_rest();
    const uint32_t r0 = tile_bins[tile_id * 2], r1 = tile_bins[tile_id * 2 + 1];
    BRUSH_DEV_BWD_TRACE(blockIdx.x * TPB + wv, r1 > r0 ? r1 - r0 : 0u);
    if (r1 <= r0) return fill_rest();
    const uint32_t tx0 = (tile_id % tbx) * kTileWidth, ty0 = (tile_id / tbx) * kTileWidth;
    // DET: zero rows (carrying their gid) for the intersections [lo, hi) this wave does not walk
    auto zero_rows = [&](uint32_t lo, uint32_t hi) {
        for (uint32_t i = lo + lane; i < hi; i += kWave) {
            float4 *r = reinterpret_cast<float4 *>(rows + (size_t)unsorted_pos[i] * kCompactStride);
            r[0] = r[1] = r[3] = make_float4(0.f, 0.f, 0.f, 0.f);  // the whole 64-byte row
            r[2] = make_float4(0.f, __uint_as_float(gid_from_isect[i]), 0.f, 0.f);
        }
    };

    // Per-quadrant pixel state (see k_rasterize_backward for D and K); pixels outside the image get
    // fin = -1 and never contribute.
    float pcx[NQ], pcy[NQ], T[NQ], KD[NQ], vor[NQ], vog[NQ], vob[NQ];  // KD = K - D of k_rasterize_backward
    int32_t fin[NQ];
    int32_t max_fin = -1;
#pragma unroll
    for (uint32_t s = 0; s < NQ; s++) {
        const uint32_t qi = sub * NQ + s;
        const uint32_t px = tx0 + (qi & 1u) * 8u + (lane & 7u), py = ty0 + (qi >> 1) * 8u + (lane >> 3);
        pcx[s] = (float)px + 0.5f;
        pcy[s] = (float)py + 0.5f;
        float T_final = 1.0f;
        float4 vo = make_float4(0.f, 0.f, 0.f, 0.f);
        fin[s] = -1;
        if (px < w && py < h) {
            const size_t pix = (size_t)px + (size_t)py * w;
            T_final = 1.0f - out_img[pix * 4 + 3];  // rasterize_backwards.wgsl:163
            fin[s] = (int32_t)final_index[pix];
            vo = reinterpret_cast<const float4 *>(v_out)[pix];
        }
        T[s] = T_final, KD[s] = T_final * vo.w;
        vor[s] = vo.x, vog[s] = vo.y, vob[s] = vo.z;
        max_fin = max(max_fin, fin[s]);
    }
    // Entries behind the wave's largest final index fail `isect_id <= final_isect` for every pixel
    // (rasterize_backwards.wgsl:229): start the walk there.
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) max_fin = max(max_fin, __shfl_xor(max_fin, d, 64));
    // (wave-uniform by construction; said so, it stays on the scalar unit with everything derived from it)
    const uint32_t walk_end = __builtin_amdgcn_readfirstlane(min(r1, (uint32_t)(max_fin + 1)));
    if (DET) zero_rows(max(walk_end, r0), r1);
    if (walk_end <= r0) return fill_rest();
    // The wave's blocks are spread evenly over the records it walks (`fill_num` blocks per `fill_den` records, an
    // error-diffusion counter on the scalar unit per batch).  Measured (profiles/r04_zero_fill_in_passing.json): the
    // placement inside a wave's life hardly matters at 8160 tiles (all blocks behind the walk: the same within 3 us),
    // a burst at every batch start costs 5 us, and on small frames whose waves all start together only the even
    // spread overlaps at all (1 M splats @512x512: step 0.332 -> 0.316 ms even, 0.328 behind the walk).
    const uint32_t fill_num = fc.quota, fill_den = walk_end - r0;
    uint32_t fill_acc = 0u, fill_budget = 0u, fill_rate = 0u;

    for (uint32_t batch_end = walk_end; batch_end > r0;) {
        const uint32_t remaining = min(kBatch, batch_end - r0);
        bool hitq[NQ];
#pragma unroll
        for (uint32_t s = 0; s < NQ; s++) hitq[s] = false;
        float rec[9];
        uint32_t cg_id = 0;
        if (lane < remaining) {
            cg_id = gid_from_isect[batch_end - 1u - lane];
            const float *p = projected + (size_t)cg_id * BRUSH_PROJECTED_FLOATS;
#pragma unroll
            for (int k = 0; k < 9; k++) rec[k] = p[k];
#pragma unroll
            for (uint32_t s = 0; s < NQ; s++) {
                const uint32_t qi = sub * NQ + s;
                hitq[s] = quad_may_pass(rec[0], rec[1], rec[2], rec[3], rec[4], rec[8],
                                        (float)(tx0 + (qi & 1u) * 8u) + 0.5f, (float)(ty0 + (qi >> 1) * 8u) + 0.5f);
            }
        }
        uint64_t qm[NQ], todo = 0ull;
#pragma unroll
        for (uint32_t s = 0; s < NQ; s++) {
            qm[s] = ballot64(hitq[s]);
            todo |= qm[s];
        }
        if (fc.quota != 0u) {
            fill_budget = min(fc.quota, ceil_div(fill_num * remaining, fill_den));
            fill_rate = fill_budget, fill_acc = 0u;
        }
        if (todo == 0ull) {
            if (DET) zero_rows(batch_end - remaining, batch_end);
            for (; fill_budget != 0u; fill_budget--) fill_step(zf, fc, lane);
            batch_end -= remaining;
            continue;
        }
        wave_sync();  // previous batch fully flushed
        if (DET && lane < remaining) {
            lds_gid[lane] = cg_id;
            lds_pos[lane] = unsorted_pos[batch_end - 1u - lane];
        }
        if ((todo >> lane) & 1ull) {
            lds_gid[lane] = cg_id;
            lds.a[lane] = make_float4(rec[0], rec[1], rec[2], rec[3]);
            lds.b[lane] = make_float4(rec[4], rec[5], rec[6], rec[7]);
            lds.opac[lane] = rec[8];
        }
        if (DET) {
            float4 *row = reinterpret_cast<float4 *>(&acc[lane][0]);
            row[0] = row[1] = row[2] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
        wave_sync();
        const uint64_t flush_mask = todo;
        // Default mode: records whose partial sums wait in `stage` (slot s holds batch slot (staged_t >> 6 s) & 63)
        uint32_t staged = 0u;
        uint64_t staged_t = 0ull;
        // The transposed reduction of the staged records (see kStageRecs): lane (row, half) = (l & 31, l >> 5) adds half
        // of row `row` = (stage slot, component), the halves meet through one lane swap, and the lower lane applies the
        // per-record factor and issues the hardware float atomic: 9 consecutive lanes on the 9 consecutive words of one
        // splat's compact row, as the L2 executes float atomics line by line.
        auto reduce_stage = [&](const uint32_t cnt) {
            const uint32_t row = lane & 31u, half = lane >> 5;
            float sum = 0.0f;
            if (row < cnt * kGradComps) {
                const float4 *src = reinterpret_cast<const float4 *>(stage + row * kRowWords + half * 32u);
                float4 v[8];
#pragma unroll
                for (int k = 0; k < 8; k++) v[k] = src[k];
                float p[8];
#pragma unroll
                for (int k = 0; k < 8; k++) p[k] = (v[k].x + v[k].y) + (v[k].z + v[k].w);
                sum = ((p[0] + p[1]) + (p[2] + p[3])) + ((p[4] + p[5]) + (p[6] + p[7]));
            }
            sum = fold_swap32(sum, sum);  // every lane: lower + upper half of its row
            if (lane < cnt * kGradComps) {
                const uint32_t slot = (row * 57u) >> 9;  // row / 9 for row < 32
                const uint32_t k = row - slot * kGradComps;
                const uint32_t t = (uint32_t)(staged_t >> (6u * slot)) & 63u;
                // rasterize_backwards.wgsl:256-263: v_xy = -opac (sum vva gx, sum vva gy), v_conic = -opac (S2 / 2, S3,
                // S4 / 2), v_rgb, v_opac = S8
                const float nopac = -lds.opac[t];
                const float scale = k >= 5u ? 1.0f : ((k == 2u || k == 4u) ? 0.5f * nopac : nopac);
                dev_flush(&v_compact[(size_t)lds_gid[t] * kCompactStride + k], sum * scale);  // != 0: one float atomic
            }
        };
        // One record: its LDS row is read one record AHEAD (software pipeline, two register sets in turn), so the
        // broadcast's latency is covered by the previous record's arithmetic instead of stalling the wave.
        auto one_record = [&](const uint32_t t, const float4 a, const float4 b, const float opac) {
            const int32_t isect_id = (int32_t)(batch_end - 1u - t);
            if (fill_budget != 0u) {
                fill_acc += fill_rate;
                while (fill_acc >= remaining && fill_budget != 0u) {
                    fill_step(zf, fc, lane);
                    fill_acc -= remaining, fill_budget--;
                }
            }
            // Zeros the compiler cannot see through: every quadrant then accumulates in place under its
            // exec mask, instead of each path materialising its own set of nine zero registers.
            float g[kGradComps];
            {
                typedef float f2v __attribute__((ext_vector_type(2)));
                f2v z01, z23, z45, z67;
                asm volatile("v_mov_b64 %0, 0" : "=v"(z01));
                asm volatile("v_mov_b64 %0, 0" : "=v"(z23));
                asm volatile("v_mov_b64 %0, 0" : "=v"(z45));
                asm volatile("v_mov_b64 %0, 0" : "=v"(z67));
                g[0] = z01.x, g[1] = z01.y, g[2] = z23.x, g[3] = z23.y;
                g[4] = z45.x, g[5] = z45.y, g[6] = z67.x, g[7] = z67.y;
                asm volatile("v_mov_b32 %0, 0" : "=v"(g[8]));
            }
            bool contributed = false;
#pragma unroll
            for (uint32_t s = 0; s < NQ; s++) {
                if (((qm[s] >> t) & 1ull) == 0ull) continue;  // scalar: the record cannot reach quadrant s
                const float dx = a.x - pcx[s], dy = a.y - pcy[s];
                const float gx = fmaf(a.z, dx, a.w * dy);
                const float gy = fmaf(a.w, dx, b.x * dy);
                const float sig2 = fmaf(dx, gx, dy * gy);  // 2 sigma
                const float vis = __builtin_amdgcn_exp2f(sig2 * kHalfNegLog2e);
                const float alpha_u = opac * vis;
                if (isect_id <= fin[s] && sig2 >= 0.0f && alpha_u >= 1.0f / 255.0f) {
                    // rasterize_backwards.wgsl:239-271
                    const float alpha = vmin(0.99f, alpha_u);  // 0.99 here, 0.999 in the forward (:239)
                    const float om = 1.0f - alpha;
                    // v_rcp_f32 is good to 1 ulp and 1 - alpha >= 0.01 (always finite); WGSL's own division is
                    // specified to 2.5 ulp, so no refinement step
                    const float ra = __builtin_amdgcn_rcpf(om);
                    const float Tn = T[s] * ra;
                    const float fac = alpha * Tn;
                    const float cv = fmaf(b.w, vob[s], fmaf(b.z, vog[s], b.y * vor[s]));
                    // v_alpha = (c*T - buffer*ra) . v_rgb + T_final*ra*v_a = T (c . v_rgb) + ra (K - D)
                    const float v_alpha = fmaf(Tn, cv, ra * KD[s]);
                    T[s] = Tn;
                    KD[s] = fmaf(-fac, cv, KD[s]);
                    // v_sigma = -opac vis v_alpha; the factors that are the same for every pixel (-opac, the conic
                    // in gx / gy, the 1/2 of the conic terms) are applied once per record at the flush:
                    //   g0 = sum vva dx, g1 = sum vva dy (default mode: vva gx, vva gy), g2..4 = sum vva (dx dx, dx dy,
                    //   dy dy), g8 = sum vva
                    const float vva = dev_vva(vis, v_alpha);  // vis * v_alpha
                    const float wx = vva * dx, wy = vva * dy;
                    if (DET) {
                        g[0] += wx;
                        g[1] += wy;
                    } else {  // the conic factors of v_xy applied per pixel: the flush scales single sums only
                        g[0] = fmaf(vva, gx, g[0]);
                        g[1] = fmaf(vva, gy, g[1]);
                    }
                    g[2] = fmaf(wx, dx, g[2]);
                    g[3] = fmaf(wx, dy, g[3]);
                    g[4] = fmaf(wy, dy, g[4]);
                    g[5] = fmaf(fac, vor[s], g[5]);
                    g[6] = fmaf(fac, vog[s], g[6]);
                    g[7] = fmaf(fac, vob[s], g[7]);
                    g[8] += vva;
                    contributed = true;
                }
            }
            if (ballot64(contributed) != 0ull) {  // wave-uniform: all 64 lanes take part in the reduction
                if (dev_skip_reduce<DET>(g, v_compact, lane)) return;  // never in the product build
                if constexpr (!DET) {
                    // park the lane partials as 9 rows of the stage: plain LDS stores, no cross-lane VALU work
                    float *dst = stage + staged * (kGradComps * kRowWords) + lane;
#pragma unroll
                    for (uint32_t k = 0; k < kGradComps; k++) dst[k * kRowWords] = g[k];
                    staged_t |= (uint64_t)t << (6u * staged);
                    if (++staged == kStageRecs) {
                        reduce_stage(kStageRecs);
                        staged = 0u, staged_t = 0ull;
                    }
                    return;
                }
                const float u0 = fold_swap32(g[0], g[1]), u1 = fold_swap32(g[2], g[3]);
                const float u2 = fold_swap32(g[4], g[5]), u3 = fold_swap32(g[6], g[7]);
                const float w0 = row_sum_lane15(fold_swap16(u0, u1));
                const float w1 = row_sum_lane15(fold_swap16(u2, u3));
                const float s8 = wave_sum_lane63(g[8]);
                if ((lane & 15u) == 15u) {
                    const uint32_t r = lane >> 4;
                    const uint32_t i0 = ((r & 1u) << 1) | (r >> 1);
                    acc[t][i0] = w0;
                    acc[t][4 + i0] = w1;
                    if (lane == 63) acc[t][8] = s8;
                }
            }
        };
        {
            uint32_t tA = (uint32_t)__builtin_ctzll(todo), tB = tA;
            float4 aA = lds.a[tA], bA = lds.b[tA], aB, bB;
            float oA = lds.opac[tA], oB;
            for (;;) {
                todo &= todo - 1ull;
                tB = todo != 0ull ? (uint32_t)__builtin_ctzll(todo) : tA;
                aB = lds.a[tB], bB = lds.b[tB], oB = lds.opac[tB];
                one_record(tA, aA, bA, oA);
                if (todo == 0ull) break;
                todo &= todo - 1ull;
                tA = todo != 0ull ? (uint32_t)__builtin_ctzll(todo) : tB;
                aA = lds.a[tA], bA = lds.b[tA], oA = lds.opac[tA];
                one_record(tB, aB, bB, oB);
                if (todo == 0ull) break;
            }
        }
        for (; fill_budget != 0u; fill_budget--) fill_step(zf, fc, lane);  // (a batch with few hits)
        if constexpr (!DET) {
            if (staged != 0u) reduce_stage(staged);
            batch_end -= remaining;
            continue;
        }
        wave_sync();
        // DET only from here: flush the batch's rows.
        // acc holds the raw pixel sums; the per-record factors (rasterize_backwards.wgsl:256-263):
        //   v_xy = -opac (a S0 + b S1, b S0 + c S1), v_conic = -opac (S2 / 2, S3, S4 / 2), v_rgb, v_opac = S8
        auto finish = [&](uint32_t t, uint32_t k) -> float {
            const float v = acc[t][k];
            if (k >= 5u) return v;
            const float4 a = lds.a[t];
            const float nopac = -lds.opac[t];
            if (k >= 2u) return (k == 3u ? nopac : 0.5f * nopac) * v;
            const float other = acc[t][k ^ 1u];
            return nopac * (k == 0u ? fmaf(a.z, v, a.w * other) : fmaf(lds.b[t].x, v, a.w * other));
        };
        // one row per intersection of the batch (zeros where nothing contributed), 16 consecutive lanes per row
        for (uint32_t f = lane; f < remaining * kCompactStride; f += kWave) {
            const uint32_t t = f / kCompactStride, k = f - t * kCompactStride;
            float v = 0.0f;
            if (k < kGradComps) {
                if ((flush_mask >> t) & 1ull) v = finish(t, k);
            } else if (k == kGradComps) {
                v = __uint_as_float(lds_gid[t]);
            }
            rows[(size_t)lds_pos[t] * kCompactStride + k] = v;
        }
        batch_end -= remaining;
    }
    fill_rest();
}

}  // namespace

// Lays the given arrays end to end as a sequence of KiB blocks (64 lanes x 16 bytes).
bool make_zero_fill(ZeroFill *zf, float *const *arrays, const size_t *floats, uint32_t count) {
    *zf = ZeroFill{};
    if (count > kFillSegs) return false;
    uint64_t blocks = 0;
    for (uint32_t i = 0; i < kFillSegs; i++) {
        zf->first_block[i] = (uint32_t)blocks;
        if (i >= count || !arrays[i] || floats[i] == 0) continue;
        const uint64_t chunks = floats[i] / 4u;
        if ((reinterpret_cast<uintptr_t>(arrays[i]) & 15u) != 0 || chunks >= (1ull << 32) - 64u) {
            *zf = ZeroFill{};
            return false;
        }
        zf->base[i] = arrays[i];
        zf->full[i] = (uint32_t)chunks;
        zf->tail[i] = (uint32_t)(floats[i] & 3u);
        blocks += (chunks + (zf->tail[i] ? 1u : 0u) + kWave - 1u) / kWave;
    }
    if (blocks >= (1ull << 31)) {
        *zf = ZeroFill{};
        return false;
    }
    zf->first_block[kFillSegs] = (uint32_t)blocks;
    return true;
}

hipError_t launch_rasterize(uint32_t w, uint32_t h, uint32_t tbx, uint32_t tby,
                            const uint32_t *compact_gid_from_isect, uint32_t *tile_bins, const uint32_t *bin_edges,
                            const float *projected, int raster_u32, uint32_t u32_pitch, void *out_img,
                            uint32_t *final_index, float *zero_rows, const uint32_t *num_visible, uint32_t n,
                            hipStream_t s) {
    const uint32_t tiles = tbx * tby;
    if (tiles == 0) return hipSuccess;
    // one workgroup (4 quadrant waves) per tile
    const dim3 grid(ceil_div(tiles, 8u) * 8u), block(kRasterThreads);
    if (raster_u32)
        hipLaunchKernelGGL(k_rasterize_quad<true>, grid, block, 0, s, w, h, tbx, tiles, compact_gid_from_isect,
                           tile_bins, bin_edges, projected, out_img, final_index, u32_pitch,
                           reinterpret_cast<float4 *>(zero_rows), num_visible, n);
    else
        hipLaunchKernelGGL(k_rasterize_quad<false>, grid, block, 0, s, w, h, tbx, tiles, compact_gid_from_isect,
                           tile_bins, bin_edges, projected, out_img, final_index, u32_pitch,
                           reinterpret_cast<float4 *>(zero_rows), num_visible, n);
    return hipGetLastError();
}

// Quadrants per wave of the backward: fewer waves per tile mean less repeated per-record work (staging, set-up and
// reduction run once per wave and record), more waves per tile fill the chip when the frame has few tiles.  Thresholds
// from a sweep of frame sizes with the LDS-transposed reduction (backward kernel, us, 1 / 2 / 4 quadrants per wave;
// profiles/r03_bwd_occupancy_schedule_experiment.json): 972 tiles 42 / 48 / 73, 1200: 56 / 46 / 66, 1728: 56 / 47 / 58,
// 2040: 60 / 45 / 48, 2500: 70 / 68 / 66 (dense scene 283 / 177 / 129), 3072: 78 / 59 / 52, 3600: 97 / 61 / 47,
// 4096: 141 / 121 / 112.  (With round 2's reduction the switch to one wave per tile paid only from 6144 tiles.)
static uint32_t backward_quadrants_per_wave(uint32_t tiles) {
    return tiles >= 2304u ? 4u : (tiles >= 1100u ? 2u : 1u);
}

hipError_t launch_rasterize_backward(uint32_t w, uint32_t h, uint32_t tbx, uint32_t tby,
                                     const uint32_t *compact_gid_from_isect, const uint32_t *tile_bins,
                                     const float *projected, const uint32_t *final_index,
                                     const float *out_img, const float *v_out, float *v_compact,
                                     const uint32_t *unsorted_pos, float *rows, const ZeroFill &fill, hipStream_t s) {
    const uint32_t tiles = tbx * tby;
    if (tiles == 0) return hipSuccess;
    // Waves per SIMD.  The kernel is bound by VALU issue once a SIMD holds 3+ waves, every wave lives for its whole
    // tile and the tiles' lists are about equally long, so the launch proceeds in rounds of (SIMDs x k) waves and a
    // partly filled last round costs as much as a full one: k in {3, 4, 5} is chosen to waste the least of the last
    // round (1080p: 8160 waves on 1024 SIMDs, k = 4 -> 2 rounds, 152 us; k = 5 -> 1.6 rounds, 164 us; k = 3: 177 us).
    // Registers and static LDS allow 4; fewer are enforced with unused dynamic LDS per workgroup.  Workgroups of 4 waves
    // (4 tiles in a row): 1, 2 and 8 measured slower (154 / 157 / 172 vs 142 us).
    static std::atomic<uint32_t> simds_of[kMaxDevices];  // per device (0 = not queried yet)
    const int slot = current_device_slot();
    uint32_t simds = simds_of[slot].load(std::memory_order_relaxed);
    if (simds == 0u || slot == kMaxDevices - 1) {
        int dev = 0, cus = 256;
        if (hipGetDevice(&dev) != hipSuccess ||
            hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0)
            cus = 256;
        simds = (uint32_t)cus * 4u;
        simds_of[slot].store(simds, std::memory_order_relaxed);
    }
    // Dynamic LDS per workgroup that admits exactly k workgroups (of 4 waves) per CU beside `static_lds` bytes of static LDS.
    constexpr uint32_t kLdsPerCu = 160u * 1024u;
    auto lds_pad_for = [&](uint32_t units, uint32_t static_lds, uint32_t max_k_regs) -> uint32_t {
        const uint32_t max_k = min(max_k_regs, kLdsPerCu / static_lds);
        uint32_t best_k = max_k, best_cost = 0xFFFFFFFFu;
        for (uint32_t k = max_k; k >= 3u; k--) {
            const uint32_t cost = ceil_div(units, simds * k) * k;  // in wave-rounds per SIMD
            if (cost < best_cost) best_cost = cost, best_k = k;
        }
        if (best_k >= max_k_regs) return 0u;  // the registers stop the (k+1)-th workgroup
        // halfway between "k + 1 fit" and "k fit": sized to the last KB (160 KB / k) the CU admitted one workgroup fewer
        // than intended (per-wave timeline: 2 resident waves per SIMD instead of 3)
        const uint32_t per_wg = ((kLdsPerCu / (best_k + 1u) + kLdsPerCu / best_k) / 2u) & ~1023u;
        return per_wg > static_lds ? per_wg - static_lds : 0u;
    };
#define BRUSH_RASTER_BWD(NQ, DET, UNSORTED, ROWS)                                                                     \
    hipLaunchKernelGGL((k_rasterize_backward_quad<NQ, DET, kTilesPerBlock>),                                          \
                       dim3(ceil_div(ceil_div(units, kTilesPerBlock), 8u) * 8u), dim3(kRasterThreads), lds_pad, s, w, \
                       h, tbx, tiles, compact_gid_from_isect, tile_bins, projected, final_index, out_img, v_out,      \
                       v_compact, UNSORTED, ROWS, fill)
    // static LDS per workgroup as the compiler lays it out (unused arrays of the other mode are dropped); registers:
    // 110-122 VGPRs -> 4 waves per SIMD
    constexpr uint32_t kStaticLdsDet = kTilesPerBlock * (sizeof(BwdRecs) + kBatch * (4u + 4u + 48u));
    constexpr uint32_t kStaticLdsDefault = kTilesPerBlock * (sizeof(BwdRecs) + kBatch * 4u + kStageRows * kRowWords * 4u);
    if (rows) {  // deterministic mode: one wave per tile, one stored row per intersection
        const uint32_t units = tiles;
        const uint32_t lds_pad = lds_pad_for(units, kStaticLdsDet, 4u);
        BRUSH_RASTER_BWD(4, true, unsorted_pos, rows);
        return hipGetLastError();
    }
    const uint32_t nq = backward_quadrants_per_wave(tiles);
    const uint32_t units = tiles * (4u / nq);
    const uint32_t lds_pad = lds_pad_for(units, kStaticLdsDefault, 4u);
    if (nq == 4) BRUSH_RASTER_BWD(4, false, nullptr, nullptr);
    else if (nq == 2) BRUSH_RASTER_BWD(2, false, nullptr, nullptr);
    else BRUSH_RASTER_BWD(1, false, nullptr, nullptr);
#undef BRUSH_RASTER_BWD
    return hipGetLastError();
}

}  // namespace brush

#define BRUSH_DEV_SECTION 2
#include "rasterize_dev.inc"
