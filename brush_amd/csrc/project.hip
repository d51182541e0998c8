// project.hip — per-splat forward stages: cull + depth key, order-preserving compaction,
// visible-splat projection (conic / xy / SH colour / opacity / exact tile count), tile
// intersection emission and tile bin edges.
//
// Replaces (paths relative to the reference checkout):
//   ProjectSplats            crates/brush-render/src/shaders/project_forward.wgsl:15-68
//   ProjectVisible           .../project_visible.wgsl:163-258
//   MapGaussiansToIntersect  .../map_gaussian_to_intersects.wgsl:10-48
//   GetTileBinEdges          .../get_tile_bin_edges.wgsl:15-42
//   CreateDispatchBuffer     crates/brush-kernel/src/shaders/wg.wgsl:15-40 (not needed: kernels
//                            read the device-side counts themselves and grid-stride)
//
// This translation unit is compiled with -ffp-contract=off: every integer decision (cull, tile
// counts, tile lists) is a function of correctly rounded f32 operations in a fixed order, so
// the visible set, depth order and per-tile lists are reproducible bit-for-bit.
//
// Differences from the reference by design:
//   * compaction is order-preserving (ascending global id) instead of an atomicAdd slot
//     (project_forward.wgsl:65), which makes equal-depth order deterministic;
//   * a compact_from_global map is produced so the backward can write dense gradients once,
//     coalesced, instead of zero-filling and scattering.
// All kernels are HBM-streaming (roofline: HBM).
#include "internal.hpp"
#include "lazy_sh.hpp"
#include "splat_math.hpp"
#include "trace.hpp"

#pragma clang fp contract(off)

namespace brush {
namespace {

constexpr uint32_t kThreads = 256;
constexpr uint32_t kCullPerThread = 4;                     // splats per lane in the cull / compact kernels
constexpr uint32_t kCullBlock = kThreads * kCullPerThread;  // splats per cull workgroup
// Tile walks.  A splat's bbox holds 1 .. tiles_x*tiles_y candidate tiles and the exact
// can_be_visible test costs ~300 VALU instructions, so the walk is split by size:
//   * bboxes of <= kSmallArea (16) tiles are walked inside project_visible, but not lane by lane:
//     the candidate tiles of the wave's 64 splats are flattened into one list (wave prefix sum of
//     the bbox areas), every lane tests one candidate per step after finding its owner splat with
//     a 6-step shuffle binary search, and each owner harvests its hit bits from the step's ballot
//     (count + 64-bit hit mask).  All lanes do useful tests regardless of how uneven the areas are;
//   * larger ones are cut into chunks of kChunkTiles (64) tiles and queued as (splat, chunk) work items
//     (one atomicAdd per wave reserves consecutive slots).  A second launch consumes the queue
//     4 or 16 items per wave (the lanes fetch the items' geometry in one memory phase, then one
//     64-tile step per item), so a whole-screen splat is spread over many waves.
// If the queue is full the lane falls back to walking its bbox inline (slow, still correct).
// The hit masks are kept so that the emission pass never repeats the exact test.
constexpr uint32_t kSmallArea = 16;      // few visible splats (latency-bound launch): short inline walks
constexpr uint32_t kSmallAreaMany = 64;
constexpr uint32_t kHalfWaveSplats = 1u << 18;  // up to this many visible splats a ProjectVisible wave takes 32 of them  // many visible splats (throughput-bound): everything one hit mask can hold
constexpr uint32_t kChunkTiles = 64;   // one 64-bit hit mask per queue item
constexpr uint32_t kWalkGroupMax = 64;  // queue items a consumer wave takes at a time when the queue is very long
// Few items: small groups (more waves, shorter serial chains); many items: amortise the memory phase and the per-item
// set-up (record gather, log / sqrt / divisions of the tile test and the walk rectangle: ~300 instructions that only
// the group's lanes execute, so a group of 16 runs them at a quarter of the wave).
__device__ __forceinline__ uint32_t walk_group(uint32_t n_items) {
    return n_items <= 16384u ? 4u : (n_items <= (1u << 18) ? 16u : kWalkGroupMax);
}

struct WalkQueue {
    uint32_t *counter;      // [1] items reserved so far (zeroed by the cull kernel)
    uint2 *items;           // [capacity] (compact gid, chunk index)
    uint32_t *chunk_count;  // [capacity] tiles hit inside the chunks up to and including this one, counted from the
                            //     start of the item's group of walk_group(n) items (group-local inclusive prefix)
    uint64_t *chunk_mask;   // [capacity] hit bitmask of the chunk's 64 tiles (count pass -> emit pass)
    uint32_t *slot_of;      // [N] queued splat: first item slot (< 2^31); inline splat: kInlineFlag
                            //     (hit mask in inline_mask) or kInlineRetest
    uint64_t *inline_mask;  // [N] hit mask of an inline splat's <= 64 bbox tiles (row-major)
    uint32_t capacity;
};
// The count pass records WHICH tiles passed, so the emit pass never repeats the exact test.
constexpr uint32_t kInlineFlag = 0x80000000u;
constexpr uint32_t kInlineRetest = 0xFFFFFFFFu;  // walked inline because the queue was full

// Serial walk of one bbox by its own lane (queue-full fallback only).
__device__ __forceinline__ uint32_t walk_inline_count(const uint32_t bb[4], const TileTest &tt, const float xy[2]) {
    uint32_t cnt = 0;
    for (uint32_t ty = bb[1]; ty < bb[3]; ty++)
        for (uint32_t tx = bb[0]; tx < bb[2]; tx++)
            if (can_be_visible(tt, tx, ty, xy)) cnt++;
    return cnt;
}

// Row / column of row-major index i in a rectangle `bw` tiles wide: i = row * bw + col.  The u32 division the compiler
// emits is ~25 instructions; here one v_rcp_f32 estimate (i < 2^24 is exact in f32, so the estimate is off by at most
// one) and an integer fix-up make it exact for every input the walks can produce.
__device__ __forceinline__ void row_col(uint32_t i, uint32_t bw, uint32_t &row, uint32_t &col) {
    int32_t q = (int32_t)(((float)i + 0.5f) * __builtin_amdgcn_rcpf((float)bw));
    int32_t r = (int32_t)i - q * (int32_t)bw;
    if (r < 0) q -= 1, r += (int32_t)bw;
    else if (r >= (int32_t)bw) q += 1, r -= (int32_t)bw;
    row = (uint32_t)q, col = (uint32_t)r;
}

// ---- two-phase tile test (splat_math.hpp: tile_test_head / tile_test_tail) --------------------------------------------
// Every candidate tile gets the cheap head at once; the ~18 % whose head returns kTileEdge wait in a per-wave LDS ring
// with their geometry until 64 of them are there, then the expensive tail runs on a full wave and its hits are OR-ed
// into the owner's late-hit words.  The owner adds them to its hit mask when the walk is over.
constexpr uint32_t kLateRing = 128;  // < 64 waiting + <= 64 pushed per step
struct LateRing {
    float4 a[kLateRing];  // q0 q1 q2 centre.x
    float4 b[kLateRing];  // centre.y | tx + (ty << 16) | owner lane + (bit << 8) | -
    uint32_t lo[kWave], hi[kWave];  // late hits of the mask owned by lane l
};
struct LateState {
    uint32_t head, count;  // wave-uniform
};
__device__ __forceinline__ void late_reset(LateRing &R, LateState &st) {
    R.lo[lane_id()] = 0u;
    R.hi[lane_id()] = 0u;
    st.head = st.count = 0u;
    __builtin_amdgcn_wave_barrier();
}
__device__ __forceinline__ void late_run(LateRing &R, uint32_t first, uint32_t n) {
    const uint32_t lane = lane_id();
    if (lane < n) {
        const uint32_t e = (first + lane) & (kLateRing - 1u);
        const float4 a = R.a[e], b = R.b[e];
        const float q[3] = {a.x, a.y, a.z};
        const float c[2] = {a.w, b.x};
        const uint32_t t = __float_as_uint(b.y), dst = __float_as_uint(b.z);
        if (tile_test_tail(q, t & 0xFFFFu, t >> 16, c)) {
            const uint32_t owner = dst & 0xFFu, bit = dst >> 8;
            atomicOr(bit < 32u ? &R.lo[owner] : &R.hi[owner], 1u << (bit & 31u));
        }
    }
    __builtin_amdgcn_wave_barrier();
}
// Must be called by all 64 lanes.  `edge`: this lane's candidate needs the tail.
__device__ __forceinline__ void late_push(LateRing &R, LateState &st, bool edge, const float q[3], const float c[2],
                                          uint32_t tx, uint32_t ty, uint32_t owner, uint32_t bit) {
    const uint64_t m = __ballot(edge);
    if (m == 0ull) return;  // wave-uniform
    if (edge) {
        const uint32_t e = (st.head + st.count + __popcll(m & lanemask_lt())) & (kLateRing - 1u);
        R.a[e] = make_float4(q[0], q[1], q[2], c[0]);
        R.b[e] = make_float4(c[1], __uint_as_float(tx | (ty << 16)), __uint_as_float(owner | (bit << 8)), 0.0f);
    }
    st.count += (uint32_t)__popcll(m);
    __builtin_amdgcn_wave_barrier();
    if (st.count >= kWave) {
        late_run(R, st.head, kWave);
        st.head = (st.head + kWave) & (kLateRing - 1u);
        st.count -= kWave;
    }
}
// Runs what is left and returns this lane's late hits.
__device__ __forceinline__ uint64_t late_flush(LateRing &R, LateState &st) {
    if (st.count) late_run(R, st.head, st.count);
    st.head = (st.head + st.count) & (kLateRing - 1u);
    st.count = 0u;
    __builtin_amdgcn_wave_barrier();
    const uint64_t late = ((uint64_t)R.hi[lane_id()] << 32) | R.lo[lane_id()];
    __builtin_amdgcn_wave_barrier();
    return late;
}

// Wave-flattened walk of the wave's small bboxes.  `area` = this lane's bbox tile count (0 if the
// lane has no small bbox).  Must be called by all 64 lanes.  Returns this lane's hit count and
// its row-major hit mask.
// `first` = row-major index (inside the lane's bbox) of the lane's first candidate: 0 for a whole
// small bbox, k * kChunkTiles for chunk k of a queued one.
__device__ __forceinline__ void walk_flat(uint32_t area, const uint32_t bb[4], const TileTest &tt, const float xy[2],
                                          uint32_t first, uint32_t &cnt, uint64_t &mask, LateRing &ring) {
    const uint32_t lane = lane_id();
    const uint32_t bw = bb[2] - bb[0];
    cnt = 0;
    mask = 0;
    const TileReach reach = make_tile_reach(tt);
    LateState st;
    late_reset(ring, st);
    const uint32_t incl = wave_inclusive_scan(area);
    const uint32_t excl = incl - area;
    const uint32_t total = wave_bcast(incl, 63u);
    for (uint32_t base = 0; base < total; base += kWave) {  // wave-uniform
        const uint32_t j = base + lane;
        // owner = number of lanes whose inclusive prefix is <= j (prefixes are non-decreasing)
        uint32_t own = 0;
#pragma unroll
        for (uint32_t step = 32; step > 0; step >>= 1)
            if (__shfl(incl, own + step - 1, 64) <= j) own += step;
        own = min(own, kWave - 1);
        TileTest ot;
        ot.q[0] = __shfl(tt.q[0], own, 64);
        ot.q[1] = __shfl(tt.q[1], own, 64);
        ot.q[2] = __shfl(tt.q[2], own, 64);
        ot.any = __shfl((int)tt.any, own, 64) != 0;
        const float oxy[2] = {__shfl(xy[0], own, 64), __shfl(xy[1], own, 64)};
        const uint32_t ob0 = __shfl(bb[0], own, 64), ob1 = __shfl(bb[1], own, 64);
        const uint32_t obw = __shfl(bw, own, 64), oexcl = __shfl(excl, own, 64), ofirst = __shfl(first, own, 64);
        TileReach orr;
        orr.rx = __shfl(reach.rx, own, 64);
        orr.ry = __shfl(reach.ry, own, 64);
        uint32_t cls = kTileMiss, tx = 0, ty = 0;
        if (j < total) {
            const uint32_t li = ofirst + (j - oexcl);
            uint32_t row, col;
            row_col(li, obw, row, col);
            tx = ob0 + col, ty = ob1 + row;
            cls = tile_test_head(ot, orr, tx, ty, oxy);
        }
        const uint64_t bal = __ballot(cls == kTileHit);
        late_push(ring, st, cls == kTileEdge, ot.q, oxy, tx, ty, own, j - oexcl);
        // harvest: this lane's candidates occupy [excl, incl) of the flattened list
        const uint32_t lo = max(excl, base), hi = min(incl, base + kWave);
        if (lo < hi) {
            const uint32_t len = hi - lo;
            const uint64_t seg = (bal >> (lo - base)) & (len == 64 ? ~0ull : ((1ull << len) - 1ull));
            mask |= seg << (lo - excl);
        }
    }
    mask |= late_flush(ring, st);
    cnt = (uint32_t)__popcll(mask);
}

__device__ __forceinline__ void walk_inline_emit(const uint32_t bb[4], const TileTest &tt, const float xy[2],
                                                 uint32_t c, uint32_t isect, uint32_t tbx, uint32_t cap,
                                                 uint32_t *__restrict__ tile_ids, uint32_t *__restrict__ gids) {
    for (uint32_t ty = bb[1]; ty < bb[3]; ty++)
        for (uint32_t tx = bb[0]; tx < bb[2]; tx++)
            if (can_be_visible(tt, tx, ty, xy) && isect < cap) {
                tile_ids[isect] = tx + ty * tbx;
                gids[isect] = c;
                isect++;
            }
}

// Geometry of one queued splat, rebuilt from its ProjectedSplat record.
struct SplatWalk {
    float xy[2];
    TileTest tt;
    TileReach reach;
    uint32_t b0, b1, bw, area;  // the walk rectangle (splat_math.hpp: walk_rect)
};
__device__ __forceinline__ SplatWalk load_walk(const ViewParams &vp, const float *__restrict__ projected, uint32_t c) {
    const float *p = projected + (size_t)c * BRUSH_PROJECTED_FLOATS;
    SplatWalk s;
    s.xy[0] = p[0];
    s.xy[1] = p[1];
    const float conic[3] = {p[2], p[3], p[4]};
    uint32_t bb[4];
    s.tt = make_tile_test(conic, p[8]);
    s.reach = make_tile_reach(s.tt);
    walk_rect(s.xy, conic, s.tt, s.reach, vp.tile_bounds, bb);
    s.b0 = bb[0];
    s.b1 = bb[1];
    s.bw = bb[2] - bb[0];
    s.area = s.bw * (bb[3] - bb[1]);
    return s;
}

// ---- ProjectSplats: cull + depth key (+ the survivors' ProjectedSplat record) -------------------
// project_forward.wgsl:15-68 and project_visible.wgsl:163-258.  Being the first launch of the forward
// pass it also publishes the uniforms buffer and clears the counters and tile_bins
// (render.rs:102-116,241-244) so that no separate init launch is needed.
// SH -> colour with the WGSL expression tree (project_visible.wgsl:51-147,232-241).
template <int DEG>
__device__ __forceinline__ void sh_colour(const ViewParams &vp, const float mean[3], const float *__restrict__ sh,
                                          float rgb[3]) {
    constexpr uint32_t ncoef = (DEG + 1) * (DEG + 1);  // compile-time: all SH loads issue together
    float dir[3];
    view_dir(vp, mean, dir);
    float Y[ncoef];
    sh_basis<ncoef>(DEG, dir, Y);
#pragma unroll
    for (int ch = 0; ch < 3; ch++) {
        float col = Y[0] * sh[ch];
        if (DEG >= 1) {
            const float inner = ((-dir[1]) * sh[1 * 3 + ch] + dir[2] * sh[2 * 3 + ch]) - dir[0] * sh[3 * 3 + ch];
            col = col + 0.48860251190292f * inner;
        }
        if (DEG >= 2) {
            float acc = Y[4] * sh[4 * 3 + ch];
#pragma unroll
            for (int k = 5; k < 9; k++) acc = acc + Y[k] * sh[k * 3 + ch];
            col = col + acc;
        }
        if (DEG >= 3) {
            float acc = Y[9] * sh[9 * 3 + ch];
#pragma unroll
            for (int k = 10; k < 16; k++) acc = acc + Y[k] * sh[k * 3 + ch];
            col = col + acc;
        }
        if (DEG >= 4) {
            float acc = Y[16] * sh[16 * 3 + ch];
#pragma unroll
            for (int k = 17; k < 25; k++) acc = acc + Y[k] * sh[k * 3 + ch];
            col = col + acc;
        }
        rgb[ch] = col + 0.5f;
    }
}

// The splat's ProjectedSplat record (project_visible.wgsl:163-258) is produced HERE, for every splat
// that passes the cull, while its parameters stream through in global-id order: the reference (and
// an earlier version of this file) recomputes it after the depth sort from seven gathers by global id,
// which on MI355X is bound by address-translation misses (one page per lane per array), not by
// bandwidth or arithmetic.  The record goes to a global-id-indexed staging row of 48 bytes;
// k_project_visible then needs ONE gather per splat.
// LAZY (BrushAux::lazy_sh): the SH block is under deferred Adam (lazy_sh.hpp): a visible splat's colour is evaluated
// from its stored coefficients with the pending zero-gradient steps replayed in registers; nothing is written back.
// (four waves per SIMD: the launch is 4 waves per SIMD at 1 M splats, one round; the LAZY degree-3 form would otherwise
// take 130 registers and a second round)
template <int DEG, bool LAZY = false>
__global__ __launch_bounds__(kThreads) __attribute__((amdgpu_waves_per_eu(4))) void k_project_cull(ViewParams vp, BrushUniforms u,
                                                           const float *__restrict__ means,
                                                           const float *__restrict__ log_scales,
                                                           const float *__restrict__ quats,
                                                           const float *__restrict__ sh_coeffs,
                                                           const float *__restrict__ raw_opac,
                                                           float4 *__restrict__ proj_global,
                                                           uint32_t *__restrict__ key_all,
                                                           uint32_t *__restrict__ compact_from_global,
                                                           uint32_t *__restrict__ block_counts,
                                                           uint32_t *__restrict__ uniforms_buffer,
                                                           uint32_t *__restrict__ num_intersections,
                                                           uint32_t *__restrict__ overflow,
                                                           uint32_t *__restrict__ tile_bins, uint32_t num_bin_words,
                                                           uint32_t *__restrict__ bin_edges,
                                                           uint32_t *__restrict__ walk_counter, LazySh lazy) {
    __shared__ uint32_t wave_cnt[kThreads / kWave];
    __shared__ uint32_t vis_list[kThreads / kWave][kCullPerThread * kWave];  // per wave: global ids that passed
    BRUSH_KTRACE(kTrCull, 0);
    // (Measured and rejected, round 3: streaming the raw opacities with Phase A's coalesced loads instead of gathering
    // them in Phase B, where every candidate's 4 bytes cost a 128-byte request: +1.5 us at 1 M splats, no gain at 21 M.)
    const uint32_t gt = blockIdx.x * kThreads + threadIdx.x;
    if (gt < kUniformWords) uniforms_buffer[gt] = reinterpret_cast<const uint32_t *>(&u)[gt];
    if (gt == 0) {
        *num_intersections = 0;
        *overflow = 0;
        *walk_counter = 0;
    }
    for (uint32_t i = gt; i < num_bin_words; i += gridDim.x * kThreads) {
        tile_bins[i] = 0;  // render.rs:241-244
        bin_edges[i] = 0;  // accumulators of the tile sort's last pass (sort_launch: edges)
    }

    // Phase A — every splat of the wave's 4 x 64 (round r covers 256 consecutive splats): the two cheap
    // rejections (behind the camera :32, and a conservative screen-bounds test that never changes a
    // decision).  Only ~1 splat in 8 survives them, so the survivors' ids are queued per wave and the
    // expensive exact projection (Phase B) runs on full waves instead of once per round at 12 % lane
    // occupancy.  The loads of all four rounds are issued up front (one memory phase per wave).
    float mean_r[kCullPerThread][3], smax_r[kCullPerThread];
    const uint32_t last = vp.total_splats ? vp.total_splats - 1 : 0;
#pragma unroll
    for (uint32_t r = 0; r < kCullPerThread; r++) {
        const size_t g = min(blockIdx.x * kCullBlock + r * kThreads + threadIdx.x, last);
        if (vp.total_splats) {  // uniform: an empty cloud has no row 0 to clamp to
#pragma unroll
            for (int k = 0; k < 3; k++) mean_r[r][k] = means[g * 3 + k];
            smax_r[r] = fmaxf(log_scales[g * 3], fmaxf(log_scales[g * 3 + 1], log_scales[g * 3 + 2]));
        } else {
            mean_r[r][0] = mean_r[r][1] = mean_r[r][2] = 0.0f;
            smax_r[r] = 0.0f;
        }
    }
    BRUSH_KTRACE_MARK(1, mean_r[0][0]);
    BRUSH_KTRACE_MARK(2, smax_r[kCullPerThread - 1]);
    uint32_t n_cand = 0;
    uint32_t *list = vis_list[threadIdx.x / kWave];
#pragma unroll
    for (uint32_t r = 0; r < kCullPerThread; r++) {
        const uint32_t g = blockIdx.x * kCullBlock + r * kThreads + threadIdx.x;
        bool maybe = false;
        if (g < vp.total_splats) {
            const float mean[3] = {mean_r[r][0], mean_r[r][1], mean_r[r][2]};
            float p_view[3];
            to_view(vp, mean, p_view);
            maybe = p_view[2] > 0.01f;  // :32
            if (maybe) {
                // radius <= 3*sqrt(lambda_max + quirk slack) + 1 with lambda_max(cov2d) <= trace <=
                // s_max^2 * cull_k / z^2 + 0.6; if even that radius leaves the tile bbox empty, the exact
                // test (:54-62) would reject as well.
                const float smax = det_expf(smax_r[r]) * 1.001f;
                const float rz = 1.0f / p_view[2];
                const float lam = smax * smax * vp.cull_k * rz * rz + 1.0f;
                const float rb = 3.0f * sqrtf(lam) + 2.0f;
                const float cxp = p_view[0] * rz * vp.focal[0] + vp.pixel_center[0];
                const float cyp = p_view[1] * rz * vp.focal[1] + vp.pixel_center[1];
                const float wpx = (float)(vp.tile_bounds[0] * kTileWidth), hpx = (float)(vp.tile_bounds[1] * kTileWidth);
                if (cxp + rb < -1.0f || cxp - rb > wpx + 1.0f || cyp + rb < -1.0f || cyp - rb > hpx + 1.0f) maybe = false;
            }
            if (!maybe) key_all[g] = kInvalid;
            compact_from_global[g] = kInvalid;
        }
        const uint64_t bal = __ballot(maybe);
        if (maybe) list[n_cand + __popcll(bal & lanemask_lt())] = g;
        n_cand += __popcll(bal);
    }
    __builtin_amdgcn_wave_barrier();

    // Phase B — exact cull (project_forward.wgsl:36-66) and, for the splats that pass, the whole
    // ProjectedSplat record (project_visible.wgsl:163-258).
    BRUSH_KTRACE_MARK(3, n_cand);
    uint32_t block_visible = 0;
    for (uint32_t base = 0; base < n_cand; base += kWave) {  // wave-uniform
        const uint32_t i = base + lane_id();
        bool visible = false;
        if (i < n_cand) {
            const uint32_t g = list[i];
            const float mean[3] = {means[(size_t)g * 3], means[(size_t)g * 3 + 1], means[(size_t)g * 3 + 2]};
            const float scale[3] = {det_expf(log_scales[(size_t)g * 3]), det_expf(log_scales[(size_t)g * 3 + 1]),
                                    det_expf(log_scales[(size_t)g * 3 + 2])};
            const float4 q4 = reinterpret_cast<const float4 *>(quats)[g];
            const float quat[4] = {q4.x, q4.y, q4.z, q4.w};
            const float ro = raw_opac[g];
            const float *sh = sh_coeffs + (size_t)g * ((DEG + 1) * (DEG + 1)) * 3;
            float p_view[3], cov2d[3];
            to_view(vp, mean, p_view);
            calc_cov2d(vp, p_view, scale, quat, cov2d);
            const float det = cov2d[0] * cov2d[2] - cov2d[1] * cov2d[1];
            if (!(det == 0.0f)) {  // :43
                float conic[3], xy[2];
                cov_to_conic(cov2d, conic);
                project_pix(vp, p_view, xy);
                const uint32_t radius = radius_from_conic(conic);
                uint32_t bb[4];
                get_tile_bbox(xy, radius, vp.tile_bounds, bb);
                if ((bb[2] - bb[0]) != 0u && (bb[3] - bb[1]) != 0u) {  // :60
                    visible = true;
                    float rgb[3];
                    if constexpr (LAZY) {
                        constexpr uint32_t kRow = (DEG + 1) * (DEG + 1) * 3;
                        float cur[kRow];
                        lazy_current_row<kRow>(lazy, sh_coeffs, g, cur);
                        sh_colour<DEG>(vp, mean, cur, rgb);
                    } else {
                        sh_colour<DEG>(vp, mean, sh, rgb);
                    }
                    float4 *row = proj_global + (size_t)g * 3;
                    row[0] = make_float4(xy[0], xy[1], conic[0], conic[1]);
                    row[1] = make_float4(conic[2], det_sigmoid(ro), 0.0f, 0.0f);
                    row[2] = make_float4(rgb[0], rgb[1], rgb[2], 0.0f);
                }
            }
            key_all[g] = visible ? __float_as_uint(p_view[2]) : kInvalid;
        }
        block_visible += __popcll(__ballot(visible));
    }
    BRUSH_KTRACE_MARK(4, block_visible);
    if (lane_id() == 0) wave_cnt[threadIdx.x / kWave] = block_visible;
    __syncthreads();
    if (threadIdx.x == 0) block_counts[blockIdx.x] = wave_cnt[0] + wave_cnt[1] + wave_cnt[2] + wave_cnt[3];
}

// Exclusive scan of block_counts in place; total -> num_visible and uniforms_buffer[25].
__global__ __launch_bounds__(1024) void k_cull_scan(uint32_t *__restrict__ block_counts, uint32_t num_blocks,
                                                    uint32_t *__restrict__ num_visible,
                                                    uint32_t *__restrict__ uniforms_buffer) {
    __shared__ uint32_t wave_tot[16];
    __shared__ uint32_t carry_s;
    if (threadIdx.x == 0) carry_s = 0;
    __syncthreads();
    for (uint32_t base = 0; base < num_blocks; base += 1024) {
        const uint32_t i = base + threadIdx.x;
        const uint32_t v = i < num_blocks ? block_counts[i] : 0u;
        const uint32_t incl = wave_inclusive_scan(v);
        if (lane_id() == 63) wave_tot[threadIdx.x / kWave] = incl;
        __syncthreads();
        uint32_t off = carry_s;
        for (uint32_t w = 0; w < threadIdx.x / kWave; w++) off += wave_tot[w];
        if (i < num_blocks) block_counts[i] = off + incl - v;
        __syncthreads();
        if (threadIdx.x == 1023) carry_s = off + incl;
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        *num_visible = carry_s;
        uniforms_buffer[kNumVisibleWord] = carry_s;
    }
}

// Order-preserving compaction of (depth key, global id); same 1024-splat partition as the cull.
// SELF_SCAN: block_offsets holds the raw per-block counts and every block sums the counts of the
// blocks before it (<= kSelfScanBlocks of them), the last block publishing the total; this saves the
// k_cull_scan launch for clouds up to 2M splats.
template <bool SELF_SCAN>
__global__ __launch_bounds__(kThreads) void k_compact(uint32_t n, const uint32_t *__restrict__ key_all,
                                                      const uint32_t *__restrict__ block_offsets,
                                                      uint32_t *__restrict__ keys, uint32_t *__restrict__ gids,
                                                      uint32_t *__restrict__ num_visible,
                                                      uint32_t *__restrict__ uniforms_buffer) {
    __shared__ uint32_t wave_cnt[kCullPerThread][kThreads / kWave];
    __shared__ uint32_t pre_s[kThreads / kWave];
    const uint32_t wid = threadIdx.x / kWave;
    BRUSH_KTRACE(kTrCompact, 0);
    // the keys do not depend on the block offsets: requested first, so both loads are in flight together
    uint32_t key[kCullPerThread];
#pragma unroll
    for (uint32_t r = 0; r < kCullPerThread; r++) {
        const uint32_t g = blockIdx.x * kCullBlock + r * kThreads + threadIdx.x;
        key[r] = g < n ? key_all[g] : kInvalid;
    }
    uint32_t before = 0;
    if (SELF_SCAN) {
        for (uint32_t i = threadIdx.x; i < blockIdx.x; i += kThreads) before += block_offsets[i];
        BRUSH_KTRACE_MARK(1, before);
        before = wave_sum(before);
        if (lane_id() == 0) pre_s[wid] = before;
    }
    uint64_t bal[kCullPerThread];
#pragma unroll
    for (uint32_t r = 0; r < kCullPerThread; r++) {
        bal[r] = __ballot(key[r] != kInvalid);
        if (lane_id() == 0) wave_cnt[r][wid] = __popcll(bal[r]);
    }
    BRUSH_KTRACE_MARK(2, key[kCullPerThread - 1]);
    __syncthreads();
    uint32_t off;
    if (SELF_SCAN) {
        off = pre_s[0] + pre_s[1] + pre_s[2] + pre_s[3];
        if (blockIdx.x == gridDim.x - 1 && threadIdx.x == 0) {
            const uint32_t total = off + block_offsets[blockIdx.x];
            *num_visible = total;
            uniforms_buffer[kNumVisibleWord] = total;
        }
    } else {
        off = block_offsets[blockIdx.x];
    }
#pragma unroll
    for (uint32_t r = 0; r < kCullPerThread; r++) {
        uint32_t mine = off + __popcll(bal[r] & lanemask_lt());
        for (uint32_t w = 0; w < kThreads / kWave; w++) {
            const uint32_t c = wave_cnt[r][w];
            if (w < wid) mine += c;
            off += c;
        }
        if (key[r] != kInvalid) {
            keys[mine] = key[r];
            gids[mine] = blockIdx.x * kCullBlock + r * kThreads + threadIdx.x;
        }
    }
}

// ---- ProjectVisible ----------------------------------------------------------------------
// project_visible.wgsl:163-258.  One visible splat per lane (compact = depth order): fetches the
// record the cull kernel staged under its global id, writes ProjectedSplat (36 B) in compact
// order, the exact tile count, and the inverse map.  Lanes c >= V clear the tail of
// global_from_compact_gid.
__global__ __launch_bounds__(kThreads) void k_project_visible(
    ViewParams vp, const float4 *__restrict__ proj_global, const uint32_t *__restrict__ num_visible,
    uint32_t *__restrict__ global_from_compact, uint32_t *__restrict__ compact_from_global,
    float *__restrict__ projected, uint32_t *__restrict__ tiles_hit, WalkQueue q, uint32_t small_switch) {
    BRUSH_KTRACE(kTrProjectVisible, 0);
    const uint32_t V = *num_visible;
    BRUSH_KTRACE_MARK(1, V);
    const uint32_t n = vp.total_splats;
    const uint32_t small_area = V > small_switch ? kSmallAreaMany : kSmallArea;
    // Tail of global_from_compact_gid (never written by the sort) := 0.
    for (uint32_t i = V + blockIdx.x * kThreads + threadIdx.x; i < n; i += gridDim.x * kThreads)
        global_from_compact[i] = 0;
    // Few visible splats: the launch is bound by the latency of one wave's chain (gather, queue reservation, walk
    // rounds), so a wave takes 32 splats instead of 64 (twice the waves, half the walk rounds each).
    const uint32_t spw = V <= kHalfWaveSplats ? 32u : kWave;
    // Block-uniform trip count: the queue reservation below is a workgroup-level collective.
    __shared__ uint32_t wave_chunks[kThreads / kWave], block_base_s;
    __shared__ LateRing rings[kThreads / kWave];
    const uint32_t wv = threadIdx.x / kWave;
    const uint32_t per_block = (kThreads / kWave) * spw;
    for (uint32_t bbase = blockIdx.x * per_block; bbase < V; bbase += gridDim.x * per_block) {
        const uint32_t c = bbase + wv * spw + lane_id();
        const bool active = c < V && lane_id() < spw;
        float xy[2] = {0.f, 0.f}, conic[3] = {0.f, 0.f, 0.f}, rgb[3] = {0.f, 0.f, 0.f};
        float opac = 0.f;
        uint32_t bb[4] = {0, 0, 0, 0};
        TileTest tt;
        tt.q[0] = tt.q[1] = tt.q[2] = 0.f;
        tt.any = false;
        if (active) {
            const uint32_t g = global_from_compact[c];
            compact_from_global[g] = c;
            const float4 *row = proj_global + (size_t)g * 3;
            const float4 r0 = row[0], r1 = row[1], r2 = row[2];
            xy[0] = r0.x, xy[1] = r0.y;
            conic[0] = r0.z, conic[1] = r0.w, conic[2] = r1.x;
            opac = r1.y;
            rgb[0] = r2.x, rgb[1] = r2.y, rgb[2] = r2.z;
            BRUSH_KTRACE_MARK(2, r0.x + r2.z);
            tt = make_tile_test(conic, opac);
            walk_rect(xy, conic, tt, make_tile_reach(tt), vp.tile_bounds, bb);
        }
        // exact tile count (project_visible.wgsl:244-250): inline for small bboxes, queued otherwise.
        // Queue slots are reserved by a wave scan of the chunk counts ...
        const uint32_t bbox_tiles = active ? (bb[2] - bb[0]) * (bb[3] - bb[1]) : 0u;
        const uint32_t nchunks = bbox_tiles > small_area ? (bbox_tiles + kChunkTiles - 1) / kChunkTiles : 0u;
        const uint32_t incl = wave_inclusive_scan(nchunks);
        const uint32_t wave_total = wave_bcast(incl, 63u);
        // ... and ONE atomicAdd per workgroup: the returning atomics of all workgroups hit one address and are
        // executed one after the other (~88 per us: 800 workgroups at the headline scene take 9 us to drain).  The
        // atomic is issued here and its result is first read after the inline walk below, which covers that wait
        // (profiles/r04_small_kernel_timeline.json: 2.1 us of every wave's 10 us stood between the two).
        if (lane_id() == 0) wave_chunks[wv] = wave_total;
        __syncthreads();
        uint32_t block_base = 0;
        if (threadIdx.x == 0) {
            const uint32_t tot = (wave_chunks[0] + wave_chunks[1]) + (wave_chunks[2] + wave_chunks[3]);
            if (tot) block_base = atomicAdd(q.counter, tot);
        }
        uint32_t wave_before = 0;
        for (uint32_t w2 = 0; w2 < wv; w2++) wave_before += wave_chunks[w2];
        // small bboxes: flattened across the wave
        const bool small = active && bbox_tiles <= small_area;
        uint32_t flat_cnt;
        uint64_t flat_mask;
        walk_flat(small ? bbox_tiles : 0u, bb, tt, xy, 0u, flat_cnt, flat_mask, rings[wv]);
        BRUSH_KTRACE_MARK(4, flat_cnt);
        if (threadIdx.x == 0) block_base_s = block_base;
        __syncthreads();
        const uint32_t wave_base = block_base_s + wave_before;
        BRUSH_KTRACE_MARK(3, wave_base);
        __syncthreads();  // the next trip overwrites wave_chunks and block_base_s
        uint32_t area = 0, slot = kInvalid;
        if (nchunks) {
            const uint32_t first = wave_base + incl - nchunks;
            if (first + nchunks <= q.capacity) {
                slot = first;
                for (uint32_t k = 0; k < nchunks; k++) q.items[first + k] = make_uint2(c, k);
            } else {
                // Does not fit: walk inline.  Reservations are disjoint, so at most one of them
                // straddles the capacity; its in-range slots get a sentinel the consumers skip.
                for (uint32_t k = first; k < q.capacity; k++) q.items[k] = make_uint2(kInvalid, 0u);
            }
        }
        if (small) {
            area = flat_cnt;
            slot = kInlineFlag;
            q.inline_mask[c] = flat_mask;
        } else if (active && slot == kInvalid) {  // queue full
            area = walk_inline_count(bb, tt, xy);
            slot = kInlineRetest;
        }
        if (active) q.slot_of[c] = slot;
        if (active) {
            float *p = projected + (size_t)c * BRUSH_PROJECTED_FLOATS;
            p[0] = xy[0];
            p[1] = xy[1];
            p[2] = conic[0];
            p[3] = conic[1];
            p[4] = conic[2];
            p[5] = rgb[0];
            p[6] = rgb[1];
            p[7] = rgb[2];
            p[8] = opac;
            tiles_hit[c] = area;
        }
    }
}

__device__ __forceinline__ uint32_t bcast(uint32_t v, uint32_t src_lane) {  // src_lane wave-uniform
    return (uint32_t)__builtin_amdgcn_readlane((int)v, (int)src_lane);
}
__device__ __forceinline__ float bcastf(float v, uint32_t src_lane) {
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), (int)src_lane));
}

// Second half of the tile count.  A wave takes walk_group(n) consecutive queue items: those lanes fetch the
// items and their splats' geometry in ONE memory phase, then the wave spends one step per item with
// the 64 lanes on the chunk's 64 tiles (geometry broadcast with v_readlane).  Besides the per-splat
// total (atomicAdd into tiles_hit) it records each item's hit mask and the running hit count inside
// the group, from which the emit pass derives every item's output offset without a scan.
__global__ __launch_bounds__(kThreads) void k_walk_count(ViewParams vp, const float *__restrict__ projected,
                                                         WalkQueue q, uint32_t *__restrict__ tiles_hit) {
    __shared__ LateRing rings[kThreads / kWave];
    LateRing &ring = rings[threadIdx.x / kWave];
    BRUSH_KTRACE(kTrWalkCount, 0);
    const uint32_t n_items = min(*q.counter, q.capacity);
    BRUSH_KTRACE_MARK(1, n_items);
    const uint32_t G = walk_group(n_items);
    const uint32_t n_groups = (n_items + G - 1) / G;
    const uint32_t lane = lane_id();
    const uint32_t waves = gridDim.x * (kThreads / kWave);
    for (uint32_t grp = blockIdx.x * (kThreads / kWave) + threadIdx.x / kWave; grp < n_groups; grp += waves) {
        const uint32_t it = grp * G + lane;
        const bool mine = lane < G && it < n_items;
        uint2 item = make_uint2(kInvalid, 0u);
        if (mine) item = q.items[it];
        const bool valid = mine && item.x != kInvalid;  // hole left by a reservation that straddled the capacity
        SplatWalk s;
        s.xy[0] = s.xy[1] = 0.f;
        s.tt.q[0] = s.tt.q[1] = s.tt.q[2] = 0.f;
        s.tt.any = false;
        s.reach.rx = s.reach.ry = 0.f;
        s.b0 = s.b1 = s.area = 0u;
        s.bw = 1u;
        if (valid) s = load_walk(vp, projected, item.x);
        BRUSH_KTRACE_MARK(2, s.area + item.x);
        const uint32_t first = item.y * kChunkTiles;
        const uint32_t len = (valid && first < s.area) ? min(s.area - first, kChunkTiles) : 0u;
        uint64_t my_mask = 0;
        const TileReach reach = s.reach;
        LateState st;
        late_reset(ring, st);
        const uint32_t in_group = min(G, n_items - grp * G);
        for (uint32_t qi = 0; qi < in_group; qi++) {  // wave-uniform
            const uint32_t qlen = bcast(len, qi);
            if (qlen == 0) continue;
            TileTest t;
            t.q[0] = bcastf(s.tt.q[0], qi), t.q[1] = bcastf(s.tt.q[1], qi), t.q[2] = bcastf(s.tt.q[2], qi);
            t.any = bcast((uint32_t)s.tt.any, qi) != 0u;
            TileReach r;
            r.rx = bcastf(reach.rx, qi), r.ry = bcastf(reach.ry, qi);
            const float xy[2] = {bcastf(s.xy[0], qi), bcastf(s.xy[1], qi)};
            const uint32_t b0 = bcast(s.b0, qi), b1 = bcast(s.b1, qi), bw = bcast(s.bw, qi);
            const uint32_t i = bcast(first, qi) + lane;
            uint32_t row, col;
            row_col(i, bw, row, col);
            const uint32_t tx = b0 + col, ty = b1 + row;
            const uint32_t cls = lane < qlen ? tile_test_head(t, r, tx, ty, xy) : kTileMiss;
            const uint64_t bal = __ballot(cls == kTileHit);
            if (lane == qi) my_mask = bal;
            late_push(ring, st, cls == kTileEdge, t.q, xy, tx, ty, qi, lane);
        }
        my_mask |= late_flush(ring, st);
        const uint32_t my_cnt = (uint32_t)__popcll(my_mask);
        BRUSH_KTRACE_MARK(3, my_cnt);
        const uint32_t pre = wave_inclusive_scan(lane < G ? my_cnt : 0u);
        if (mine) {
            q.chunk_count[it] = pre;
            q.chunk_mask[it] = my_mask;
            if (my_cnt) atomicAdd(&tiles_hit[item.x], my_cnt);
        }
    }
}

// ---- MapGaussiansToIntersect ---------------------------------------------------------------
// map_gaussian_to_intersects.wgsl:10-48: splats walked inline by project_visible emit here inline;
// queued splats are emitted by the queue role of the same launch.
// Index of the k-th set bit (k = 0 is the lowest) of a 64-bit mask that has more than k bits set.
__device__ __forceinline__ uint32_t kth_set_bit(uint64_t m, uint32_t k) {
    uint32_t w = (uint32_t)m, base = 0;
    const uint32_t c0 = __popc(w);
    if (k >= c0) k -= c0, w = (uint32_t)(m >> 32), base = 32;
#pragma unroll
    for (uint32_t s = 16; s >= 1; s >>= 1) {
        const uint32_t c = __popc(w & ((1u << s) - 1u));
        if (k >= c) k -= c, w >>= s, base += s;
    }
    return base;
}

// Inline splats, two forms chosen from the visible count (block-uniform):
//  * up to kFlatEmitMin visible splats: one lane per splat replays its recorded hit mask into its own output run (the
//    launch is bound by its dependent-load chain, the short serial loops are free);
//  * beyond: the wave's 64 consecutive inline splats own ONE contiguous output range (their offsets are consecutive
//    values of the scan), so the emission is flattened like the count walk: lane l of step s writes entry 64 s + l of
//    that range, finding its splat by a shuffle binary search over the running hit counts and its tile as the k-th set
//    bit of that splat's hit mask.  Consecutive lanes, consecutive addresses: at 2 M visible splats / 18 M
//    intersections 139 -> 79 us for the launch (lane-private runs: a stride of ~9 entries between neighbouring lanes);
//    at 100 k visible splats the flat form is 2 us SLOWER (same-box A/B), hence the switch.
constexpr uint32_t kFlatEmitMin = 1u << 19;
__device__ __forceinline__ void map_inline_role(uint32_t bid, uint32_t nblocks, const ViewParams &vp,
                                                const float *__restrict__ projected,
                                                const uint32_t *__restrict__ cum_tiles_hit,
                                                const uint32_t *__restrict__ num_visible, uint32_t cap,
                                                uint32_t *__restrict__ tile_ids, uint32_t *__restrict__ gids,
                                                const WalkQueue &q) {
    const uint32_t V = *num_visible;
    const uint32_t lane = lane_id();
    const bool flat = V >= kFlatEmitMin;
    const uint32_t wave_stride = nblocks * kThreads;
    for (uint32_t cbase = bid * kThreads + (threadIdx.x / kWave) * kWave; cbase < V; cbase += wave_stride) {  // wave-uniform
        const uint32_t c = cbase + lane;
        const uint32_t code = c < V ? q.slot_of[c] : 0u;
        const bool replay = (code & kInlineFlag) && code != kInlineRetest;
        uint32_t bb[4] = {0, 0, 0, 0};
        uint32_t start = 0;
        uint64_t mask = 0;
        float xy[2] = {0.f, 0.f};
        TileTest tt;
        tt.q[0] = tt.q[1] = tt.q[2] = 0.f;
        tt.any = false;
        if (code & kInlineFlag) {  // inline splat (replayed or re-tested): its rectangle and its first output slot
            const float *p = projected + (size_t)c * BRUSH_PROJECTED_FLOATS;
            xy[0] = p[0], xy[1] = p[1];
            const float conic[3] = {p[2], p[3], p[4]};
            tt = make_tile_test(conic, p[8]);
            walk_rect(xy, conic, tt, make_tile_reach(tt), vp.tile_bounds, bb);
            start = c > 0 ? cum_tiles_hit[c - 1] : 0u;
            if (replay) mask = q.inline_mask[c];
        }
        const uint32_t bw = bb[2] - bb[0];
        if (!flat) {
            uint32_t isect = start;
            while (mask) {  // row-major over the walk rectangle
                const uint32_t i = __ffsll((long long)mask) - 1;
                mask &= mask - 1;
                if (isect < cap) {
                    uint32_t row, col;
                    row_col(i, bw, row, col);
                    tile_ids[isect] = (bb[0] + col) + (bb[1] + row) * vp.tile_bounds[0];
                    gids[isect] = c;
                    isect++;
                }
            }
        } else {
            const uint32_t cnt = (uint32_t)__popcll(mask);
            const uint32_t incl = wave_inclusive_scan(cnt), excl = incl - cnt;
            const uint32_t total = wave_bcast(incl, 63u);
            const uint32_t mlo = (uint32_t)mask, mhi = (uint32_t)(mask >> 32);
            for (uint32_t base = 0; base < total; base += kWave) {  // wave-uniform
                const uint32_t j = base + lane;
                uint32_t own = 0;
#pragma unroll
                for (uint32_t step = 32; step > 0; step >>= 1)
                    if (__shfl(incl, own + step - 1, 64) <= j) own += step;
                own = min(own, kWave - 1);
                const uint64_t omask = ((uint64_t)__shfl(mhi, own, 64) << 32) | __shfl(mlo, own, 64);
                const uint32_t ob0 = __shfl(bb[0], own, 64), ob1 = __shfl(bb[1], own, 64), obw = __shfl(bw, own, 64);
                const uint32_t k = j - __shfl(excl, own, 64);
                const uint32_t pos = __shfl(start, own, 64) + k;
                if (j < total && pos < cap) {
                    uint32_t row, col;
                    row_col(kth_set_bit(omask, k), obw, row, col);
                    tile_ids[pos] = (ob0 + col) + (ob1 + row) * vp.tile_bounds[0];
                    gids[pos] = cbase + own;
                }
            }
        }
        // queue-full fallback: the splat was counted by an inline walk and is walked again here (rare)
        if (code == kInlineRetest) walk_inline_emit(bb, tt, xy, c, start, vp.tile_bounds[0], cap, tile_ids, gids);
    }
}

// Emission for the queued splats, walk_group(n) items per wave like the count pass.  Item (c, k) writes
// after the k earlier chunks of its splat, which are the k items before it in the queue: their hit
// total is a difference of the group-local running counts (at most 128 / G + 2 loads), so
// the entries of a splat land in [cum[c-1], cum[c]) in row-major bbox order, as an inline walk
// writes them.
__device__ __forceinline__ void map_queue_role(uint32_t bid, uint32_t nblocks, const ViewParams &vp,
                                               const float *__restrict__ projected,
                                               const uint32_t *__restrict__ cum_tiles_hit, uint32_t cap,
                                               uint32_t *__restrict__ tile_ids, uint32_t *__restrict__ gids,
                                               const WalkQueue &q) {
    const uint32_t n_items = min(*q.counter, q.capacity);
    const uint32_t G = walk_group(n_items);
    const uint32_t n_groups = (n_items + G - 1) / G;
    const uint32_t lane = lane_id();
    const uint64_t lt = lanemask_lt();
    const uint32_t waves = nblocks * (kThreads / kWave);
    for (uint32_t grp = bid * (kThreads / kWave) + threadIdx.x / kWave; grp < n_groups; grp += waves) {
        const uint32_t g_first = grp * G;
        const uint32_t it = g_first + lane;
        const bool mine = lane < G && it < n_items;
        uint2 item = make_uint2(kInvalid, 0u);
        if (mine) item = q.items[it];
        const bool valid = mine && item.x != kInvalid;
        uint32_t b0 = 0, b1 = 0, bw = 1, base = 0, first = 0;
        uint64_t mask = 0;
        if (valid) {
            const uint32_t c = item.x, k = item.y;
            mask = q.chunk_mask[it];
            // hits of the k preceding items = [it - k, it): whole groups by their last running count,
            // the two partial groups by differences
            uint32_t before = 0;
            const uint32_t lo = it - k;  // first item of this splat
            if (lo >= g_first) {         // all in this group
                before = (lane > 0 ? q.chunk_count[it - 1] : 0u) - (lo > g_first ? q.chunk_count[lo - 1] : 0u);
            } else {
                before = lane > 0 ? q.chunk_count[it - 1] : 0u;                       // this group's part
                const uint32_t lo_grp = lo / G;
                for (uint32_t g2 = lo_grp + 1; g2 < grp; g2++) before += q.chunk_count[g2 * G + G - 1];
                const uint32_t lg_last = lo_grp * G + G - 1;                             // the splat's first group
                before += q.chunk_count[lg_last] - (lo > lo_grp * G ? q.chunk_count[lo - 1] : 0u);
            }
            const float *pp = projected + (size_t)c * BRUSH_PROJECTED_FLOATS;
            const float xy[2] = {pp[0], pp[1]};
            const float conic[3] = {pp[2], pp[3], pp[4]};
            uint32_t bb[4];
            const TileTest tt = make_tile_test(conic, pp[8]);
            walk_rect(xy, conic, tt, make_tile_reach(tt), vp.tile_bounds, bb);
            b0 = bb[0], b1 = bb[1], bw = bb[2] - bb[0];
            first = k * kChunkTiles;
            base = (c > 0 ? cum_tiles_hit[c - 1] : 0u) + before;
        }
        const uint32_t mlo = (uint32_t)mask, mhi = (uint32_t)(mask >> 32);
        const uint32_t in_group = min(G, n_items - g_first);
        for (uint32_t qi = 0; qi < in_group; qi++) {  // wave-uniform
            const uint64_t bal = ((uint64_t)bcast(mhi, qi) << 32) | bcast(mlo, qi);
            if (bal == 0ull) continue;
            const uint32_t qb0 = bcast(b0, qi), qb1 = bcast(b1, qi), qbw = bcast(bw, qi);
            const uint32_t i = bcast(first, qi) + lane;
            const uint32_t pos = bcast(base, qi) + __popcll(bal & lt);
            const uint32_t qc = bcast(item.x, qi);
            if (((bal >> lane) & 1ull) && pos < cap) {
                uint32_t row, col;
                row_col(i, qbw, row, col);
                tile_ids[pos] = (qb0 + col) + (qb1 + row) * vp.tile_bounds[0];
                gids[pos] = qc;
            }
        }
    }
}

// One launch, two roles: the first `inline_blocks` workgroups emit the inline splats, the rest consume
// the queue (they write disjoint ranges of the same arrays, so neither waits for the other).
__global__ __launch_bounds__(kThreads) void k_map_intersects(ViewParams vp, const float *__restrict__ projected,
                                                             const uint32_t *__restrict__ cum_tiles_hit,
                                                             const uint32_t *__restrict__ num_visible, uint32_t cap,
                                                             uint32_t *__restrict__ tile_ids,
                                                             uint32_t *__restrict__ gids, WalkQueue q,
                                                             uint32_t inline_blocks) {
    BRUSH_KTRACE(kTrMap, blockIdx.x < inline_blocks ? 0u : (1u << 24) | 1u);
    if (blockIdx.x < inline_blocks)
        map_inline_role(blockIdx.x, inline_blocks, vp, projected, cum_tiles_hit, num_visible, cap, tile_ids, gids, q);
    else
        map_queue_role(blockIdx.x - inline_blocks, gridDim.x - inline_blocks, vp, projected, cum_tiles_hit, cap,
                       tile_ids, gids, q);
}

// ---- GetTileBinEdges -----------------------------------------------------------------------
// get_tile_bin_edges.wgsl:15-42
// perm != nullptr (deterministic mode): the tile sort carried the PRE-SORT positions as values; the compact gid
// of sorted intersection i is then gathered from the unsorted list here.
__global__ __launch_bounds__(kThreads) void k_tile_bin_edges(const uint32_t *__restrict__ sorted_tile_ids,
                                                             const uint32_t *__restrict__ num_intersections,
                                                             uint32_t *__restrict__ tile_bins,
                                                             const uint32_t *__restrict__ perm,
                                                             const uint32_t *__restrict__ gid_unsorted,
                                                             uint32_t *__restrict__ gid_sorted) {
    const uint32_t I = *num_intersections;
    for (uint32_t i = blockIdx.x * kThreads + threadIdx.x; i < I; i += gridDim.x * kThreads) {
        const uint32_t cur = sorted_tile_ids[i];
        if (perm) gid_sorted[i] = gid_unsorted[perm[i]];
        if (i == I - 1) tile_bins[cur * 2 + 1] = I;
        if (i == 0) {
            tile_bins[cur * 2 + 0] = 0;
        } else {
            const uint32_t prev = sorted_tile_ids[i - 1];
            if (prev != cur) {
                tile_bins[prev * 2 + 1] = i;
                tile_bins[cur * 2 + 0] = i;
            }
        }
    }
}

uint32_t stride_grid(uint32_t work_items) { return max(1u, min(ceil_div(work_items, kThreads), 2048u)); }
WalkQueue make_queue(const WalkWs &w) {
    WalkQueue q;
    q.counter = w.counter;
    q.items = reinterpret_cast<uint2 *>(w.items);
    q.chunk_count = w.chunk_count;
    q.chunk_mask = reinterpret_cast<uint64_t *>(w.chunk_mask);
    q.slot_of = w.slot_of;
    q.inline_mask = reinterpret_cast<uint64_t *>(w.inline_mask);
    q.capacity = w.capacity;
    return q;
}

}  // namespace

constexpr uint32_t kSelfScanBlocks = 2048;

size_t cull_block_count(uint32_t n) { return ceil_div(n ? n : 1, kCullBlock); }

hipError_t launch_project_cull(const ViewParams &vp, const BrushUniforms &u, const BrushAux &aux,
                               uint32_t num_tiles, const float *means, const float *log_scales,
                               const float *quats, const float *sh, const float *raw_opac, float *proj_global,
                               uint32_t *key_all, uint32_t *block_counts, uint32_t *keys,
                               uint32_t *gids, uint32_t *bin_edges, const WalkWs &walk, const LazySh &lazy,
                               hipStream_t s) {
    const uint32_t n = vp.total_splats;
    const uint32_t blocks = (uint32_t)cull_block_count(n);
    uint32_t *compact_from_global = aux.compact_from_global_gid;
    uint32_t *num_visible = aux.num_visible;
    uint32_t *uniforms_buffer = aux.uniforms_buffer;
#define BRUSH_LAUNCH_CULL(D, L)                                                                                  \
    hipLaunchKernelGGL((k_project_cull<D, L>), dim3(blocks), dim3(kThreads), 0, s, vp, u, means, log_scales, quats, \
                       sh, raw_opac, reinterpret_cast<float4 *>(proj_global), key_all, compact_from_global,      \
                       block_counts, uniforms_buffer, aux.num_intersections, aux.overflow, aux.tile_bins,        \
                       num_tiles * 2, bin_edges, walk.counter, lazy)
    if (lazy.on()) {  // rows of whole 16-byte chunks only (make_lazy_sh)
        if (vp.sh_degree == 1) BRUSH_LAUNCH_CULL(1, true);
        else BRUSH_LAUNCH_CULL(3, true);
    } else {
        switch (vp.sh_degree) {
            case 0: BRUSH_LAUNCH_CULL(0, false); break;
            case 1: BRUSH_LAUNCH_CULL(1, false); break;
            case 2: BRUSH_LAUNCH_CULL(2, false); break;
            case 3: BRUSH_LAUNCH_CULL(3, false); break;
            default: BRUSH_LAUNCH_CULL(4, false); break;
        }
    }
#undef BRUSH_LAUNCH_CULL
    if (blocks <= kSelfScanBlocks) {
        hipLaunchKernelGGL(k_compact<true>, dim3(blocks), dim3(kThreads), 0, s, n, key_all, block_counts, keys, gids,
                           num_visible, uniforms_buffer);
    } else {
        hipLaunchKernelGGL(k_cull_scan, dim3(1), dim3(1024), 0, s, block_counts, blocks, num_visible,
                           uniforms_buffer);
        hipLaunchKernelGGL(k_compact<false>, dim3(blocks), dim3(kThreads), 0, s, n, key_all, block_counts, keys, gids,
                           num_visible, uniforms_buffer);
    }
    return hipGetLastError();
}

hipError_t launch_project_visible(const ViewParams &vp, const float *proj_global, const uint32_t *num_visible,
                                  uint32_t *global_from_compact, uint32_t *compact_from_global, float *projected,
                                  uint32_t *tiles_hit, const WalkWs &walk, hipStream_t s) {
    const WalkQueue q = make_queue(walk);
    const dim3 grid(stride_grid(vp.total_splats)), block(kThreads);
    // visible-splat count above which bboxes up to 64 tiles are walked inline (S3, 2 M visible: 470 -> 259 us; at
    // 100 k visible the short inline walks win: 28.5 vs 50.4 us)
    const uint32_t small_switch = 1u << 19;
    hipLaunchKernelGGL(k_project_visible, grid, block, 0, s, vp, reinterpret_cast<const float4 *>(proj_global),
                       num_visible, global_from_compact, compact_from_global, projected, tiles_hit, q, small_switch);
    hipLaunchKernelGGL(k_walk_count, dim3(1024), dim3(kThreads), 0, s, vp, projected, q, tiles_hit);
    return hipGetLastError();
}

hipError_t launch_map_intersects(const ViewParams &vp, const float *projected, const uint32_t *cum_tiles_hit,
                                 const uint32_t *num_visible, uint32_t cap, uint32_t *tile_ids, uint32_t *gids,
                                 const WalkWs &walk, hipStream_t s) {
    const WalkQueue q = make_queue(walk);
    const uint32_t inline_blocks = stride_grid(vp.total_splats);
    hipLaunchKernelGGL(k_map_intersects, dim3(inline_blocks + 1024u), dim3(kThreads), 0, s, vp, projected,
                       cum_tiles_hit, num_visible, cap, tile_ids, gids, q, inline_blocks);
    return hipGetLastError();
}

hipError_t launch_tile_bin_edges(const uint32_t *sorted_tile_ids, const uint32_t *num_intersections,
                                 uint32_t cap, uint32_t *tile_bins, const uint32_t *perm,
                                 const uint32_t *gid_unsorted, uint32_t *gid_sorted, hipStream_t s) {
    hipLaunchKernelGGL(k_tile_bin_edges, dim3(stride_grid(cap)), dim3(kThreads), 0, s, sorted_tile_ids,
                       num_intersections, tile_bins, perm, gid_unsorted, gid_sorted);
    return hipGetLastError();
}

}  // namespace brush
