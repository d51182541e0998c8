// render.hip — host orchestration of the forward and backward passes behind the C ABI.
//
// Replaces render_forward (crates/brush-render/src/render.rs:55-323) and
// RenderBackwards::backward (render.rs:465-626).  Like the reference it only enqueues work on
// the caller's stream and never reads a count back to the host; unlike the reference it
// performs no allocation (the caller passes one workspace) and needs no indirect-dispatch
// helper kernels (kernels read the device-side counts and grid-stride).
#include <stdlib.h>
#include <string.h>

#include "internal.hpp"
#include "splat_math.hpp"
#include "trace.hpp"

namespace brush {

static thread_local int g_last_hip_error = 0;
void set_last_hip_error(int e) { g_last_hip_error = e; }

}  // namespace brush

// Opt-in stage timing: events [0] = forward start, [1 + stage] = end of forward stage,
// [kBwdStart] = backward start, then one per backward stage.
struct BrushProfiler {
    static constexpr int kFwdStages = BRUSH_STAGE_RASTERIZE + 1;
    static constexpr int kBwdStages = BRUSH_NUM_STAGES - kFwdStages;
    hipEvent_t fwd[kFwdStages + 1];
    hipEvent_t bwd[kBwdStages + 1];
    bool fwd_recorded = false, bwd_recorded = false;
    uint32_t fwd_mask = 0, bwd_mask = 0;  // events recorded by the last pass (a stage without a launch records none)
    int stop_after = -1;  // brush_profiler_stop_after: the pass ends behind this stage, no events
};

namespace brush {
static thread_local BrushProfiler *g_prof = nullptr;

// true: the attached profiler asks the pass to end behind `stage` (prefix timing)
static inline bool stop_behind(int stage) { return g_prof && g_prof->stop_after == stage; }

static inline void mark_fwd(hipStream_t s, int idx) {
    if (g_prof && g_prof->stop_after < 0) {
        if (idx == 0) g_prof->fwd_mask = 0;
        (void)hipEventRecord(g_prof->fwd[idx], s);
        g_prof->fwd_mask |= 1u << idx;
        if (idx == BrushProfiler::kFwdStages) g_prof->fwd_recorded = true;
    }
}
static inline void mark_bwd(hipStream_t s, int idx) {
    if (g_prof && g_prof->stop_after < 0) {
        if (idx == 0) g_prof->bwd_mask = 0;
        (void)hipEventRecord(g_prof->bwd[idx], s);
        g_prof->bwd_mask |= 1u << idx;
        if (idx == BrushProfiler::kBwdStages) g_prof->bwd_recorded = true;
    }
}

namespace {

struct FwdWs {
    uint32_t *key_all;       // [N] depth bits or 0xFFFFFFFF
    uint32_t *block_counts;  // [ceil(N/256)]
    uint32_t *pre_keys;      // [N] compacted, unsorted
    uint32_t *pre_gids;      // [N]
    uint32_t *sorted_keys;   // [N]
    uint32_t *tiles_hit;     // [N]
    float *proj_global;      // [N][12] projected record staged by the cull kernel under the global id
    uint32_t *tile_unsorted; // [cap]
    uint32_t *gid_unsorted;  // [cap]
    uint32_t *tile_sorted;   // [cap]
    uint32_t *bin_edges;     // [tiles][2] (~start, end) accumulated by the tile sort's last pass
    void *scan_ws;
    void *sort_ws;           // sized for max(N, cap)
    WalkWs walk;             // (splat, chunk) queue of the tile walks
    size_t bytes;
};

FwdWs carve_fwd(void *ws, uint32_t n, uint32_t cap, uint32_t num_tiles) {
    FwdWs f;
    Carver c(ws);
    const size_t nn = n ? n : 1, cc = cap ? cap : 1;
    f.key_all = c.take<uint32_t>(nn);
    f.block_counts = c.take<uint32_t>(cull_block_count(n));
    f.pre_keys = c.take<uint32_t>(nn);
    f.pre_gids = c.take<uint32_t>(nn);
    f.sorted_keys = c.take<uint32_t>(nn);
    f.tiles_hit = c.take<uint32_t>(nn);
    f.proj_global = c.take<float>(nn * 12);
    f.tile_unsorted = c.take<uint32_t>(cc);
    f.gid_unsorted = c.take<uint32_t>(cc);
    f.tile_sorted = c.take<uint32_t>(cc);
    f.bin_edges = c.take<uint32_t>((size_t)(num_tiles ? num_tiles : 1) * 2);
    f.scan_ws = c.take<char>(scan_workspace_bytes(n));
    f.sort_ws = c.take<char>(sort_workspace_bytes(n > cap ? n : cap));
    f.walk.capacity = (uint32_t)nn;
    f.walk.counter = c.take<uint32_t>(1);
    f.walk.items = c.take<uint32_t>(nn * 2);
    f.walk.chunk_count = c.take<uint32_t>(nn);
    f.walk.chunk_mask = c.take<uint32_t>(nn * 8);
    f.walk.slot_of = c.take<uint32_t>(nn);
    f.walk.inline_mask = c.take<uint32_t>(nn * 2);
    f.bytes = c.bytes();
    return f;
}

struct BwdWs {
    float *v_compact;  // [N, kCompactStride] compact-order gradient rows (first V used)
    float *rows;       // deterministic mode: [cap, kCompactStride] one row per intersection, emission order
    float *partials;   // deterministic mode: [ceil(cap / 64), 2, kCompactStride] chunk-border partial sums
    size_t bytes;
};

BwdWs carve_bwd(void *ws, uint32_t n, uint32_t cap, bool det) {
    BwdWs b;
    Carver c(ws);
    const size_t nn = n ? n : 1;
    b.v_compact = c.take<float>(nn * kCompactStride);
    b.rows = b.partials = nullptr;
    if (det) {
        const size_t cc = cap ? cap : 1;
        b.rows = c.take<float>(cc * kCompactStride);
        b.partials = c.take<float>((size_t)ceil_div((uint32_t)cc, kWave) * 2 * kCompactStride);
    }
    b.bytes = c.bytes();
    return b;
}

inline bool aux_det(const BrushAux &a) { return (a.flags & BRUSH_AUX_DETERMINISTIC) != 0; }

bool aux_ok(const BrushAux *a, bool need_final_index) {
    if (a && (a->flags & ~(BRUSH_AUX_DETERMINISTIC | BRUSH_AUX_ACCUM_ZEROED)) != 0) return false;  // unknown flag bits
    if (a && aux_det(*a) && !a->isect_unsorted_pos) return false;
    return a && a->projected_splats && a->uniforms_buffer && a->num_intersections && a->num_visible &&
           (a->final_index || !need_final_index) && a->cum_tiles_hit && a->tile_bins &&
           a->compact_gid_from_isect && a->global_from_compact_gid && a->compact_from_global_gid && a->overflow;
}

bool uniforms_ok(const BrushUniforms *u) {
    if (!u || u->sh_degree > 4) return false;  // render.rs:44-52
    if (u->tile_bounds[0] != ceil_div(u->img_size[0], kTileWidth) ||
        u->tile_bounds[1] != ceil_div(u->img_size[1], kTileWidth))
        return false;  // render.rs:82-85
    return true;
}

}  // namespace
}  // namespace brush

using namespace brush;

#ifndef BRUSH_BUILD_TAG
#define BRUSH_BUILD_TAG ""  // the test / development twins of the library name themselves here (Makefile)
#endif
extern "C" const char *brush_version(void) { return "brush_amd 0.4.0 (gfx950)" BRUSH_BUILD_TAG; }

extern "C" const char *brush_status_string(int status) {
    switch (status) {
        case BRUSH_OK: return "ok";
        case BRUSH_ERR_INVALID_ARG: return "invalid argument";
        case BRUSH_ERR_WORKSPACE_SMALL: return "workspace too small";
        case BRUSH_ERR_HIP: return "HIP runtime error";
        case BRUSH_ERR_NO_DEVICE: return "no HIP device";
        default: return "unknown status";
    }
}

extern "C" int brush_last_hip_error(void) { return g_last_hip_error; }

extern "C" uint32_t brush_default_max_intersects(uint32_t n, uint32_t w, uint32_t h) {
    const uint64_t tiles = (uint64_t)ceil_div(w, kTileWidth) * ceil_div(h, kTileWidth);
    uint64_t m = (uint64_t)n * tiles;  // render.rs:204-206 (saturating_mul)
    if (m > 128ull * 65535ull) m = 128ull * 65535ull;
    return m ? (uint32_t)m : 1u;
}

extern "C" int brush_fwd_workspace_size(uint32_t n, uint32_t w, uint32_t h, uint32_t sh_degree,
                                        uint32_t max_intersects, size_t *bytes) {
    if (!bytes || sh_degree > 4) return BRUSH_ERR_INVALID_ARG;
    *bytes = carve_fwd(nullptr, n, max_intersects, ceil_div(w, kTileWidth) * ceil_div(h, kTileWidth)).bytes;
    return BRUSH_OK;
}

// Host-side default only (not cached, not read by any render entry point): see brush_hip.h.
extern "C" int brush_deterministic(void) {
    const char *e = getenv("BRUSH_DETERMINISTIC");
    return (e && e[0] == '1') ? 1 : 0;
}

extern "C" int brush_bwd_workspace_size_flags(uint32_t n, uint32_t w, uint32_t h, uint32_t sh_degree,
                                              uint32_t max_intersects, uint32_t flags, size_t *bytes) {
    (void)w;
    (void)h;
    if (!bytes || sh_degree > 4 || (flags & ~BRUSH_AUX_DETERMINISTIC) != 0) return BRUSH_ERR_INVALID_ARG;
    *bytes = carve_bwd(nullptr, n, max_intersects, (flags & BRUSH_AUX_DETERMINISTIC) != 0).bytes;
    return BRUSH_OK;
}

extern "C" int brush_bwd_workspace_size_ex(uint32_t n, uint32_t w, uint32_t h, uint32_t sh_degree,
                                           uint32_t max_intersects, size_t *bytes) {
    return brush_bwd_workspace_size_flags(n, w, h, sh_degree, max_intersects, 0u, bytes);
}

extern "C" int brush_bwd_workspace_size(uint32_t n, uint32_t w, uint32_t h, uint32_t sh_degree, size_t *bytes) {
    return brush_bwd_workspace_size_ex(n, w, h, sh_degree, brush_default_max_intersects(n, w, h), bytes);
}

static int render_forward_impl(const BrushUniforms *h_uniforms, const float *means, const float *log_scales,
                               const float *quats, const float *sh_coeffs, const float *raw_opacity, uint32_t n,
                               int raster_u32, uint32_t u32_pitch, void *out_img, const BrushAux *h_aux,
                               void *workspace, size_t workspace_bytes, brush_stream_t stream) {
    if (!uniforms_ok(h_uniforms) || !aux_ok(h_aux, !raster_u32) || !out_img || !workspace)
        return BRUSH_ERR_INVALID_ARG;
    if (n > 0 && (!means || !log_scales || !quats || !sh_coeffs || !raw_opacity)) return BRUSH_ERR_INVALID_ARG;
    const BrushAux &aux = *h_aux;
    const uint32_t cap = aux.max_intersects;
    if (cap == 0) return BRUSH_ERR_INVALID_ARG;
    const FwdWs ws = carve_fwd(workspace, n, cap, h_uniforms->tile_bounds[0] * h_uniforms->tile_bounds[1]);
    if (workspace_bytes < ws.bytes) return BRUSH_ERR_WORKSPACE_SMALL;
    hipStream_t s = static_cast<hipStream_t>(stream);

    BrushUniforms u = *h_uniforms;
    u.num_visible = 0;
    u.total_splats = n;
    u.padding = 0;
    const ViewParams vp = make_view_params(u, n);
    const uint32_t w = u.img_size[0], h = u.img_size[1];
    const uint32_t tbx = u.tile_bounds[0], tby = u.tile_bounds[1];
    const uint32_t num_tiles = tbx * tby;

    LazySh lazy;  // deferred Adam of the SH block (BrushAux::lazy_sh): colours from the caught-up coefficients
    if (!make_lazy_sh(aux.lazy_sh, u.sh_degree, &lazy)) return BRUSH_ERR_INVALID_ARG;
    mark_fwd(s, 0);
    // uniforms buffer, counters, tile_bins = 0; ProjectSplats + order-preserving compaction
    // (render.rs:102-142)
    BRUSH_HIP_CHECK(launch_project_cull(vp, u, aux, num_tiles, means, log_scales, quats, sh_coeffs, raw_opacity,
                                        ws.proj_global, ws.key_all, ws.block_counts, ws.pre_keys, ws.pre_gids,
                                        ws.bin_edges, ws.walk, lazy, s));
    mark_fwd(s, 1 + BRUSH_STAGE_PROJECT_CULL);
    if (stop_behind(BRUSH_STAGE_PROJECT_CULL)) return BRUSH_OK;
    // DepthSort: keys = f32 depth bits, all 32 bits (render.rs:151-156)
    BRUSH_HIP_CHECK(sort_launch(ws.pre_keys, ws.pre_gids, ws.sorted_keys, aux.global_from_compact_gid,
                                aux.num_visible, n, 32, ws.sort_ws, s));
    mark_fwd(s, 1 + BRUSH_STAGE_DEPTH_SORT);
    if (stop_behind(BRUSH_STAGE_DEPTH_SORT)) return BRUSH_OK;
    // ProjectVisible (render.rs:161-184)
    BRUSH_HIP_CHECK(launch_project_visible(vp, ws.proj_global, aux.num_visible, aux.global_from_compact_gid,
                                           aux.compact_from_global_gid, aux.projected_splats, ws.tiles_hit,
                                           ws.walk, s));
    mark_fwd(s, 1 + BRUSH_STAGE_PROJECT_VISIBLE);
    if (stop_behind(BRUSH_STAGE_PROJECT_VISIBLE)) return BRUSH_OK;
    // PrefixSum over all N, tail treated as 0 (render.rs:186-192); total -> num_intersections
    BRUSH_HIP_CHECK(scan_launch(ws.tiles_hit, aux.cum_tiles_hit, n, aux.num_visible, aux.num_intersections, cap,
                                aux.overflow, ws.scan_ws, s));
    mark_fwd(s, 1 + BRUSH_STAGE_PREFIX_SUM);
    if (stop_behind(BRUSH_STAGE_PREFIX_SUM)) return BRUSH_OK;
    // MapGaussiansToIntersect (render.rs:211-223)
    BRUSH_HIP_CHECK(launch_map_intersects(vp, aux.projected_splats, aux.cum_tiles_hit, aux.num_visible, cap,
                                          ws.tile_unsorted, ws.gid_unsorted, ws.walk, s));
    mark_fwd(s, 1 + BRUSH_STAGE_MAP_INTERSECTS);
    if (stop_behind(BRUSH_STAGE_MAP_INTERSECTS)) return BRUSH_OK;
    // Tile sort on bits = 32 - clz(num_tiles) (render.rs:227-237)
    uint32_t bits = 0;
    while (bits < 32 && (num_tiles >> bits) != 0) bits++;
    // Deterministic mode: the sort carries the pre-sort positions (aux.isect_unsorted_pos) and the gids follow by
    // a gather in the bin-edge kernel.  Default mode: GetTileBinEdges (render.rs:239-262) has no launch of its own —
    // the sort's last pass sees every key next to its neighbours in final order and records the edges of the runs
    // (ws.bin_edges), the compositing kernel decodes them and writes aux.tile_bins.
    const bool det = aux_det(aux);
    // The sorted tile ids are read by nobody once the edges come out of the sort: not written in the default mode.  (With
    // three or more passes, i.e. more than 65 535 tiles, the ping-pong also routes an intermediate pass through the
    // output buffer, so it is kept then.)
    const bool drop_keys = !det && bits <= 16;
    BRUSH_HIP_CHECK(sort_launch(ws.tile_unsorted, det ? nullptr : ws.gid_unsorted, drop_keys ? nullptr : ws.tile_sorted,
                                det ? aux.isect_unsorted_pos : aux.compact_gid_from_isect, aux.num_intersections, cap,
                                bits, ws.sort_ws, s, det ? nullptr : ws.bin_edges, num_tiles));
    mark_fwd(s, 1 + BRUSH_STAGE_TILE_SORT);
    if (stop_behind(BRUSH_STAGE_TILE_SORT)) return BRUSH_OK;
    if (det) {
        BRUSH_HIP_CHECK(launch_tile_bin_edges(ws.tile_sorted, aux.num_intersections, cap, aux.tile_bins,
                                              aux.isect_unsorted_pos, ws.gid_unsorted, aux.compact_gid_from_isect, s));
        mark_fwd(s, 1 + BRUSH_STAGE_TILE_BINS);
    }  // default mode: the stage has no launch (the edges come out of the tile sort), so it records no event either
    if (stop_behind(BRUSH_STAGE_TILE_BINS)) return BRUSH_OK;
    // Rasterize (render.rs:267-307)
    // (default mode, aux.bwd_accum given: the kernel also zeroes the backward's accumulator rows, BRUSH_AUX_ACCUM_ZEROED)
    BRUSH_HIP_CHECK(launch_rasterize(w, h, tbx, tby, aux.compact_gid_from_isect, aux.tile_bins,
                                     det ? nullptr : ws.bin_edges, aux.projected_splats, raster_u32,
                                     u32_pitch ? u32_pitch : w, out_img, aux.final_index,
                                     det ? nullptr : aux.bwd_accum, aux.num_visible, n, s));
    mark_fwd(s, 1 + BRUSH_STAGE_RASTERIZE);
    return BRUSH_OK;
}

extern "C" int brush_render_forward(const BrushUniforms *h_uniforms, const float *means, const float *log_scales,
                                    const float *quats, const float *sh_coeffs, const float *raw_opacity,
                                    uint32_t n, int raster_u32, void *out_img, const BrushAux *h_aux,
                                    void *workspace, size_t workspace_bytes, brush_stream_t stream) {
    return render_forward_impl(h_uniforms, means, log_scales, quats, sh_coeffs, raw_opacity, n, raster_u32, 0, out_img,
                               h_aux, workspace, workspace_bytes, stream);
}

extern "C" int brush_render_forward_rgba8(const BrushUniforms *h_uniforms, const float *means,
                                          const float *log_scales, const float *quats, const float *sh_coeffs,
                                          const float *raw_opacity, uint32_t n, uint32_t *out_img,
                                          uint32_t row_pitch_pixels, const BrushAux *h_aux, void *workspace,
                                          size_t workspace_bytes, brush_stream_t stream) {
    if (!h_uniforms || row_pitch_pixels < h_uniforms->img_size[0]) return BRUSH_ERR_INVALID_ARG;
    return render_forward_impl(h_uniforms, means, log_scales, quats, sh_coeffs, raw_opacity, n, 1, row_pitch_pixels,
                               out_img, h_aux, workspace, workspace_bytes, stream);
}

extern "C" uint32_t brush_rgba8_row_pitch(uint32_t width) { return (width + 63u) / 64u * 64u; }

// RasterizeBackwards (render.rs:505-532): the compact-order sums of every visible splat.  Default: zero the
// accumulators, add with hardware float atomics.  Deterministic mode: one stored row per intersection, then the
// fixed-order per-splat sums (no zero-fill, no atomics).  Fills `det` for the consumer kernel.
// `fill` (may be inactive): zero-fill the compositing kernel carries in passing; *filled says whether it ran.
static int composite_backward(const BrushUniforms &u, const BrushAux &aux, const float *out_img, const float *v_out,
                              uint32_t n, const BwdWs &ws, DetSumsArgs *det, const ZeroFill &fill, bool *filled,
                              hipStream_t s) {
    const uint32_t w = u.img_size[0], h = u.img_size[1], tbx = u.tile_bounds[0], tby = u.tile_bounds[1];
    *filled = fill.active() && tbx * tby != 0u;  // (no tiles: no launch)
    if (ws.rows) {
        // no zero-fill stage in this mode: no launch, no event
        BRUSH_HIP_CHECK(launch_rasterize_backward(w, h, tbx, tby, aux.compact_gid_from_isect, aux.tile_bins,
                                                  aux.projected_splats, aux.final_index, out_img, v_out, ws.v_compact,
                                                  aux.isect_unsorted_pos, ws.rows, fill, s));
        BRUSH_HIP_CHECK(launch_sum_isect_rows(ws.rows, aux.num_intersections, aux.cum_tiles_hit, aux.max_intersects,
                                              ws.v_compact, ws.partials, s));
        mark_bwd(s, 2);
        det->cum_tiles_hit = aux.cum_tiles_hit;
        det->num_intersections = aux.num_intersections;
        det->partials = ws.partials;
        det->cap = aux.max_intersects;
        return BRUSH_OK;
    }
    // compact-order accumulators are atomically added to: zero the first V rows (render.rs:505-507) — unless the
    // forward of this render already did (BrushAux::bwd_accum + BRUSH_AUX_ACCUM_ZEROED)
    if (!(aux.flags & BRUSH_AUX_ACCUM_ZEROED)) {
        BRUSH_HIP_CHECK(launch_zero_compact_grads(aux.num_visible, n, ws.v_compact, s));
        mark_bwd(s, 1);
    }
    if (stop_behind(BRUSH_STAGE_BWD_ZERO)) return BRUSH_OK;
    BRUSH_HIP_CHECK(launch_rasterize_backward(w, h, tbx, tby, aux.compact_gid_from_isect, aux.tile_bins,
                                              aux.projected_splats, aux.final_index, out_img, v_out, ws.v_compact,
                                              nullptr, nullptr, fill, s));
    mark_bwd(s, 2);
    return BRUSH_OK;
}

static int render_backward_impl(const BrushUniforms *h_uniforms, const BrushAux *h_aux, const float *means,
                                const float *log_scales, const float *quats, const float *raw_opacity, uint32_t n,
                                const float *out_img, const float *v_out, float *v_means, float *v_xy,
                                float *v_scales, float *v_quats, float *v_sh, float *v_opac, const AdamFuse *adam,
                                void *workspace, size_t workspace_bytes, brush_stream_t stream) {
    if (!uniforms_ok(h_uniforms) || !aux_ok(h_aux, true) || !out_img || !v_out || !workspace)
        return BRUSH_ERR_INVALID_ARG;
    if (n > 0 && (!means || !log_scales || !quats || !raw_opacity || !v_xy)) return BRUSH_ERR_INVALID_ARG;
    if (n > 0 && !adam && (!v_means || !v_scales || !v_quats || !v_sh || !v_opac)) return BRUSH_ERR_INVALID_ARG;
    const BrushAux &aux = *h_aux;
    const BwdWs ws = carve_bwd(workspace, n, aux.max_intersects, aux_det(aux));
    if (workspace_bytes < ws.bytes) return BRUSH_ERR_WORKSPACE_SMALL;
    if ((aux.flags & BRUSH_AUX_ACCUM_ZEROED) && !aux_det(aux) && aux.bwd_accum != ws.v_compact) return BRUSH_ERR_INVALID_ARG;
    hipStream_t s = static_cast<hipStream_t>(stream);

    BrushUniforms u = *h_uniforms;
    u.total_splats = n;
    const ViewParams vp = make_view_params(u, n);

    // Dense gradients (render.rs:539-547,573-575 zero-fill them with launches of their own): the VALU-bound compositing
    // kernel stores the zeros in passing, the VJP kernel behind it writes the visible splats' rows only.
    ZeroFill fill{};
    if (!adam && n > 0) {
        const size_t nn = n, C = (size_t)(u.sh_degree + 1) * (u.sh_degree + 1);
        float *const arrays[kFillSegs] = {v_sh, v_means, v_scales, v_quats, v_opac, v_xy};
        const size_t floats[kFillSegs] = {nn * C * 3, nn * 3, nn * 3, nn * 4, nn, nn * 2};
        (void)make_zero_fill(&fill, arrays, floats, kFillSegs);  // (unaligned / oversized arrays: zeros stay in the VJP kernel)
        // a frame of a few tiles has a few waves: more than 1 MiB of zeros per tile would turn the compositing launch
        // into a slow fill (1 M splats on a 64 x 64 frame); the all-in-one VJP kernel writes them at stream rate
        const uint64_t tiles = (uint64_t)u.tile_bounds[0] * u.tile_bounds[1];
        if (fill.first_block[kFillSegs] > 1024ull * (tiles ? tiles : 1ull)) fill = ZeroFill{};
    }
    mark_bwd(s, 0);
    DetSumsArgs det;
    bool filled = false;
    if (const int rc = composite_backward(u, aux, out_img, v_out, n, ws, &det, fill, &filled, s)) return rc;
    if (stop_behind(BRUSH_STAGE_BWD_ZERO) || stop_behind(BRUSH_STAGE_RASTERIZE_BWD)) return BRUSH_OK;
    // GatherGrads + ProjectBackwards fused, dense outputs written once (render.rs:534-594)
    BRUSH_HIP_CHECK(launch_project_backward(vp, means, log_scales, quats, raw_opacity, aux.compact_from_global_gid,
                                            ws.v_compact, v_means, v_xy, v_scales, v_quats, v_sh, v_opac, adam, det,
                                            filled, s));
    mark_bwd(s, 3);
    return BRUSH_OK;
}

extern "C" int brush_lazy_sh_fill_table(float beta1, float beta2, float lr_coeffs_dc, float sh_rest_lerp, uint32_t base,
                                        uint32_t capacity, float *host_rows) {
    if (!host_rows || base > 0xFFFFFFFFu - capacity) return BRUSH_ERR_INVALID_ARG;
    for (uint32_t i = 0; i < capacity; i++) {
        adam_bias_corrections(beta1, beta2, base + 1u + i, &host_rows[4 * i], &host_rows[4 * i + 1]);
        host_rows[4 * i + 2] = lr_coeffs_dc;
        host_rows[4 * i + 3] = sh_rest_lerp;
    }
    return BRUSH_OK;
}

// Shared by the fused backward+Adam forms: validates the optimizer arguments and fills the kernel-side struct.
static bool fill_adam_fuse(const BrushAdamConfig *cfg, float *means, float *log_scales, float *rotation,
                           float *raw_opacity, float *sh, uint32_t n, float *moment1, float *moment2,
                           float *next_quats_fed, float *grad_2d_accum, float *xy_grad_counts, uint32_t w, uint32_t h,
                           uint32_t sh_degree, AdamFuse *out) {
    if (!cfg || cfg->time == 0) return false;
    if (n > 0 && (!means || !log_scales || !rotation || !raw_opacity || !sh || !moment1 || !moment2)) return false;
    auto aligned = [](const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
    if (!aligned(rotation) || (next_quats_fed && !aligned(next_quats_fed))) return false;
    if ((grad_2d_accum == nullptr) != (xy_grad_counts == nullptr)) return false;
    AdamFuse af{};
    af.means = means, af.log_scales = log_scales, af.rotation = rotation, af.raw_opac = raw_opacity, af.sh = sh;
    af.m1 = moment1, af.m2 = moment2;
    af.lr[0] = cfg->lr_mean, af.lr[1] = cfg->lr_scale, af.lr[2] = cfg->lr_rotation, af.lr[3] = cfg->lr_opac;
    af.lr[4] = cfg->lr_coeffs_dc;
    af.sh_lerp = cfg->sh_rest_lerp, af.beta1 = cfg->beta1, af.beta2 = cfg->beta2, af.eps = cfg->epsilon;
    adam_bias_corrections(cfg->beta1, cfg->beta2, cfg->time, &af.rbc1, &af.rbc2);
    af.quat_vjp = cfg->rotation_grad_wrt_normalized;
    af.norm_rot_out = next_quats_fed;
    af.grad_2d_accum = grad_2d_accum, af.xy_grad_counts = xy_grad_counts;
    af.half_w = (float)w / 2.0f, af.half_h = (float)h / 2.0f;
    af.stat_scale = cfg->xy_stat_scale > 0.0f ? cfg->xy_stat_scale : 1.0f;
    af.vec_ok = (n % 4 == 0) && aligned(means) && aligned(log_scales) && aligned(sh) && aligned(moment1) &&
                aligned(moment2);
    if (cfg->lazy_sh) {
        // deferred Adam of the SH block: this step is optimizer time now + 1, the SH segments are the tails of the
        // moment arrays, and the chunked (16-byte) path must be the one that runs
        if (!make_lazy_sh(cfg->lazy_sh, sh_degree, &af.lazy) || !af.vec_ok || cfg->lazy_sh->now + 1u != cfg->time ||
            cfg->lazy_sh->sh_moment1 != moment1 + (size_t)11 * n || cfg->lazy_sh->sh_moment2 != moment2 + (size_t)11 * n)
            return false;
    }
    *out = af;
    return true;
}

extern "C" int brush_render_backward(const BrushUniforms *h_uniforms, const BrushAux *h_aux, const float *means,
                                     const float *log_scales, const float *quats, const float *raw_opacity,
                                     uint32_t n, const float *out_img, const float *v_out, float *v_means,
                                     float *v_xy, float *v_scales, float *v_quats, float *v_sh, float *v_opac,
                                     void *workspace, size_t workspace_bytes, brush_stream_t stream) {
    return render_backward_impl(h_uniforms, h_aux, means, log_scales, quats, raw_opacity, n, out_img, v_out, v_means,
                                v_xy, v_scales, v_quats, v_sh, v_opac, nullptr, workspace, workspace_bytes, stream);
}

extern "C" int brush_render_backward_adam(const BrushUniforms *h_uniforms, const BrushAux *h_aux,
                                          const BrushAdamConfig *cfg, float *means, float *log_scales,
                                          const float *quats_fed, float *rotation, float *raw_opacity, float *sh,
                                          uint32_t n, const float *out_img, const float *v_out, float *v_xy,
                                          float *moment1, float *moment2, float *next_quats_fed,
                                          float *grad_2d_accum, float *xy_grad_counts, void *workspace,
                                          size_t workspace_bytes, brush_stream_t stream) {
    if (!cfg || cfg->time == 0 || !h_uniforms || h_uniforms->sh_degree > 4) return BRUSH_ERR_INVALID_ARG;
    if ((reinterpret_cast<uintptr_t>(quats_fed) & 15) != 0) return BRUSH_ERR_INVALID_ARG;
    AdamFuse af{};
    if (!fill_adam_fuse(cfg, means, log_scales, rotation, raw_opacity, sh, n, moment1, moment2, next_quats_fed,
                        grad_2d_accum, xy_grad_counts, h_uniforms->img_size[0], h_uniforms->img_size[1],
                        h_uniforms->sh_degree, &af))
        return BRUSH_ERR_INVALID_ARG;
    return render_backward_impl(h_uniforms, h_aux, means, log_scales, quats_fed, raw_opacity, n, out_img, v_out,
                                nullptr, v_xy, nullptr, nullptr, nullptr, nullptr, &af, workspace, workspace_bytes,
                                stream);
}

// ---- view-sharded data parallelism (build extension) ------------------------------------------

extern "C" int brush_render_backward_records(const BrushUniforms *h_uniforms, const BrushAux *h_aux, const float *means,
                                             const float *log_scales, const float *quats, const float *raw_opacity,
                                             uint32_t n, const float *out_img, const float *v_out, float *records,
                                             uint32_t max_rows, void *workspace, size_t workspace_bytes,
                                             brush_stream_t stream) {
    if (!uniforms_ok(h_uniforms) || !aux_ok(h_aux, true) || !out_img || !v_out || !workspace)
        return BRUSH_ERR_INVALID_ARG;
    if (n > 0 && (!means || !log_scales || !quats || !raw_opacity || (max_rows > 0 && !records)))
        return BRUSH_ERR_INVALID_ARG;
    if ((reinterpret_cast<uintptr_t>(records) & 15) != 0) return BRUSH_ERR_INVALID_ARG;
    const BrushAux &aux = *h_aux;
    const BwdWs ws = carve_bwd(workspace, n, aux.max_intersects, aux_det(aux));
    if (workspace_bytes < ws.bytes) return BRUSH_ERR_WORKSPACE_SMALL;
    if ((aux.flags & BRUSH_AUX_ACCUM_ZEROED) && !aux_det(aux) && aux.bwd_accum != ws.v_compact) return BRUSH_ERR_INVALID_ARG;
    hipStream_t s = static_cast<hipStream_t>(stream);
    BrushUniforms u = *h_uniforms;
    u.total_splats = n;
    const ViewParams vp = make_view_params(u, n);
    mark_bwd(s, 0);
    DetSumsArgs det;
    bool filled = false;
    if (const int rc = composite_backward(u, aux, out_img, v_out, n, ws, &det, ZeroFill{}, &filled, s)) return rc;
    BRUSH_HIP_CHECK(launch_project_backward_records(vp, means, log_scales, quats, raw_opacity, aux.num_visible,
                                                    aux.global_from_compact_gid, ws.v_compact, records, max_rows, det,
                                                    s));
    mark_bwd(s, 3);
    return BRUSH_OK;
}

extern "C" int brush_view_index_size(uint32_t n, uint32_t num_views, size_t *bytes) {
    if (!bytes) return BRUSH_ERR_INVALID_ARG;
    *bytes = sizeof(uint32_t) * (size_t)(n ? n : 1) * (num_views ? num_views : 1);
    return BRUSH_OK;
}

static int reduce_views_impl(const float *records, uint32_t num_views, uint32_t rows_per_view, const uint32_t *view_rows,
                             const uint32_t *view_offsets, const float *campos, const float *means, uint32_t n,
                             uint32_t sh_degree, float *v_means,
                             float *v_scales, float *v_quats, float *v_sh, float *v_opac, const AdamFuse *adam,
                             void *view_index, size_t view_index_bytes, brush_stream_t stream) {
    if (sh_degree > 4 || num_views == 0) return BRUSH_ERR_INVALID_ARG;
    if (n == 0) return BRUSH_OK;
    if (!means || !view_index || !view_rows || !campos || (rows_per_view > 0 && !records)) return BRUSH_ERR_INVALID_ARG;
    if ((reinterpret_cast<uintptr_t>(records) & 15) != 0) return BRUSH_ERR_INVALID_ARG;
    if (!adam && (!v_means || !v_scales || !v_quats || !v_sh || !v_opac)) return BRUSH_ERR_INVALID_ARG;
    if (view_index_bytes < sizeof(uint32_t) * (size_t)n * num_views) return BRUSH_ERR_WORKSPACE_SMALL;
    if (!view_offsets && (uint64_t)num_views * rows_per_view > 0xFFFFFFFFull) return BRUSH_ERR_INVALID_ARG;
    BRUSH_HIP_CHECK(launch_reduce_view_records(records, num_views, rows_per_view, view_rows, view_offsets, campos, means, n,
                                               sh_degree,
                                               static_cast<uint32_t *>(view_index), v_means, v_scales, v_quats, v_sh,
                                               v_opac, adam, static_cast<hipStream_t>(stream)));
    return BRUSH_OK;
}

extern "C" int brush_reduce_view_records(const float *records, uint32_t num_views, uint32_t rows_per_view,
                                         const uint32_t *view_rows, const uint32_t *view_offsets, const float *campos,
                                         const float *means, uint32_t n, uint32_t sh_degree, float *v_means,
                                         float *v_scales, float *v_quats, float *v_sh, float *v_opac, void *view_index,
                                         size_t view_index_bytes, brush_stream_t stream) {
    return reduce_views_impl(records, num_views, rows_per_view, view_rows, view_offsets, campos, means, n, sh_degree, v_means,
                             v_scales, v_quats, v_sh, v_opac, nullptr, view_index, view_index_bytes, stream);
}

extern "C" int brush_reduce_view_records_adam(const float *records, uint32_t num_views, uint32_t rows_per_view,
                                              const uint32_t *view_rows, const uint32_t *view_offsets,
                                              const float *campos,
                                              const BrushAdamConfig *cfg, uint32_t width, uint32_t height,
                                              float *means, float *log_scales, float *rotation, float *raw_opacity,
                                              float *sh, uint32_t n, uint32_t sh_degree, float *moment1, float *moment2,
                                              float *next_quats_fed, float *grad_2d_accum, float *xy_grad_counts,
                                              void *view_index, size_t view_index_bytes, brush_stream_t stream) {
    AdamFuse af{};
    if (!fill_adam_fuse(cfg, means, log_scales, rotation, raw_opacity, sh, n, moment1, moment2, next_quats_fed,
                        grad_2d_accum, xy_grad_counts, width, height, sh_degree, &af))
        return BRUSH_ERR_INVALID_ARG;
    return reduce_views_impl(records, num_views, rows_per_view, view_rows, view_offsets, campos, means, n, sh_degree,
                             nullptr, nullptr,
                             nullptr, nullptr, nullptr, &af, view_index, view_index_bytes, stream);
}

extern "C" int brush_lazy_sh_flush(const BrushLazySh *h_lazy, float *sh, uint32_t n, uint32_t sh_degree,
                                   brush_stream_t stream) {
    LazySh lazy;
    if (!h_lazy || sh_degree > 4 || !make_lazy_sh(h_lazy, sh_degree, &lazy)) return BRUSH_ERR_INVALID_ARG;
    if (n == 0) return BRUSH_OK;
    if (!sh || (reinterpret_cast<uintptr_t>(sh) & 15) != 0) return BRUSH_ERR_INVALID_ARG;
    BRUSH_HIP_CHECK(launch_lazy_sh_flush(lazy, sh, n, 3u * (sh_degree + 1u) * (sh_degree + 1u),
                                         static_cast<hipStream_t>(stream)));
    return BRUSH_OK;
}

#ifdef BRUSH_TRACE
// ---- development build only (make trace): the in-kernel timeline ring of trace.hpp --------------------------
#include <vector>
namespace brush {
static std::vector<void (*)(TraceRec *, uint32_t *)> &trace_attachers() {
    static std::vector<void (*)(TraceRec *, uint32_t *)> v;
    return v;
}
void trace_register(void (*attach)(TraceRec *, uint32_t *)) { trace_attachers().push_back(attach); }
static TraceRec *g_trace_ring = nullptr;
static uint32_t *g_trace_cursor = nullptr;
}  // namespace brush
// Allocates the ring on first use, points every translation unit at it and resets the cursor.
extern "C" int brush_debug_trace_begin(void) {
    if (!g_trace_ring) {
        BRUSH_HIP_CHECK(hipMalloc(&g_trace_ring, sizeof(TraceRec) * kTraceCap));
        BRUSH_HIP_CHECK(hipMalloc(&g_trace_cursor, 256));
        for (auto f : trace_attachers()) f(g_trace_ring, g_trace_cursor);
    }
    BRUSH_HIP_CHECK(hipMemset(g_trace_ring, 0, sizeof(TraceRec) * kTraceCap));
    BRUSH_HIP_CHECK(hipDeviceSynchronize());
    return BRUSH_OK;
}
// Copies up to max_records 64-byte records to the host; returns how many were written (< 0: error).
extern "C" long brush_debug_trace_read(void *out, size_t max_records) {
    if (!g_trace_ring || !out) return -1;
    if (hipDeviceSynchronize() != hipSuccess) return -2;
    uint32_t n = kTraceCap;  // slots nobody wrote stay zero (t[0] == 0)
    if (n > max_records) n = (uint32_t)max_records;
    if (n && hipMemcpy(out, g_trace_ring, sizeof(TraceRec) * (size_t)n, hipMemcpyDeviceToHost) != hipSuccess) return -4;
    return (long)n;
}
#endif

// ---- opt-in stage timing ------------------------------------------------------------------

extern "C" int brush_profiler_create(BrushProfiler **out) {
    if (!out) return BRUSH_ERR_INVALID_ARG;
    BrushProfiler *p = new BrushProfiler();
    for (auto &e : p->fwd) BRUSH_HIP_CHECK(hipEventCreate(&e));
    for (auto &e : p->bwd) BRUSH_HIP_CHECK(hipEventCreate(&e));
    *out = p;
    return BRUSH_OK;
}

extern "C" void brush_profiler_destroy(BrushProfiler *p) {
    if (!p) return;
    if (g_prof == p) g_prof = nullptr;
    for (auto &e : p->fwd) (void)hipEventDestroy(e);
    for (auto &e : p->bwd) (void)hipEventDestroy(e);
    delete p;
}

extern "C" void brush_profiler_attach(BrushProfiler *p) { g_prof = p; }

extern "C" int brush_profiler_read(BrushProfiler *p, float *h_ms) {
    if (!p || !h_ms) return BRUSH_ERR_INVALID_ARG;
    for (int i = 0; i < BRUSH_NUM_STAGES; i++) h_ms[i] = 0.0f;
    // a stage that launched nothing recorded no event: it reads 0 and the next stage is measured from the last event
    if (p->fwd_recorded) {
        BRUSH_HIP_CHECK(hipEventSynchronize(p->fwd[BrushProfiler::kFwdStages]));
        int last = 0;
        for (int i = 0; i < BrushProfiler::kFwdStages; i++) {
            if (!((p->fwd_mask >> (i + 1)) & 1u)) continue;
            BRUSH_HIP_CHECK(hipEventElapsedTime(&h_ms[i], p->fwd[last], p->fwd[i + 1]));
            last = i + 1;
        }
    }
    if (p->bwd_recorded) {
        BRUSH_HIP_CHECK(hipEventSynchronize(p->bwd[BrushProfiler::kBwdStages]));
        int last = 0;
        for (int i = 0; i < BrushProfiler::kBwdStages; i++) {
            if (!((p->bwd_mask >> (i + 1)) & 1u)) continue;
            BRUSH_HIP_CHECK(hipEventElapsedTime(&h_ms[BrushProfiler::kFwdStages + i], p->bwd[last], p->bwd[i + 1]));
            last = i + 1;
        }
    }
    return BRUSH_OK;
}

extern "C" int brush_profiler_stop_after(BrushProfiler *p, int stage) {
    if (!p || stage < -1 || stage >= BRUSH_NUM_STAGES) return BRUSH_ERR_INVALID_ARG;
    p->stop_after = stage;
    return BRUSH_OK;
}

extern "C" const char *brush_stage_name(int stage) {
    static const char *names[BRUSH_NUM_STAGES] = {"project_cull", "depth_sort", "project_visible", "prefix_sum",
                                                  "map_intersects", "tile_sort", "tile_bins", "rasterize",
                                                  "bwd_zero", "rasterize_bwd", "project_bwd"};
    return (stage >= 0 && stage < BRUSH_NUM_STAGES) ? names[stage] : "?";
}
