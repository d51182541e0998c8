// internal.hpp — host-side launch functions, one group per translation unit.
#pragma once
#include "common.hpp"
#include "lazy_sh.hpp"

namespace brush {

struct ViewParams;

// project.hip
// Device work queue of (splat, tile-chunk) items for the balanced tile walks.
struct WalkWs {
    uint32_t *counter;      // [1]
    uint32_t *items;        // [capacity * 2]
    uint32_t *chunk_count;  // [capacity]
    uint32_t *chunk_mask;   // [capacity * 8] (four 64-bit hit masks per item)
    uint32_t *slot_of;      // [N]
    uint32_t *inline_mask;  // [N * 2] (64-bit hit mask per inline splat)
    uint32_t capacity;
};
size_t cull_block_count(uint32_t n);
hipError_t launch_project_cull(const ViewParams &vp, const BrushUniforms &u, const BrushAux &aux,
                               uint32_t num_tiles, const float *means, const float *log_scales,
                               const float *quats, const float *sh, const float *raw_opac, float *proj_global,
                               uint32_t *key_all, uint32_t *block_counts, uint32_t *keys,
                               uint32_t *gids, uint32_t *bin_edges /* [num_tiles][2], zeroed here */,
                               const WalkWs &walk, const LazySh &lazy /* BrushAux::lazy_sh, or off */, hipStream_t s);
hipError_t launch_project_visible(const ViewParams &vp, const float *proj_global, const uint32_t *num_visible,
                                  uint32_t *global_from_compact, uint32_t *compact_from_global, float *projected,
                                  uint32_t *tiles_hit, const WalkWs &walk, hipStream_t s);
hipError_t launch_map_intersects(const ViewParams &vp, const float *projected, const uint32_t *cum_tiles_hit,
                                 const uint32_t *num_visible, uint32_t cap, uint32_t *tile_ids, uint32_t *gids,
                                 const WalkWs &walk, hipStream_t s);
// perm / gid_unsorted / gid_sorted: deterministic mode only (nullptr otherwise), see k_tile_bin_edges.
hipError_t launch_tile_bin_edges(const uint32_t *sorted_tile_ids, const uint32_t *num_intersections,
                                 uint32_t cap, uint32_t *tile_bins, const uint32_t *perm,
                                 const uint32_t *gid_unsorted, uint32_t *gid_sorted, hipStream_t s);

// rasterize.hip
// bin_edges != nullptr: the tile sort's last pass left (~start, end) per tile there (sort_launch: edges) instead of a
// GetTileBinEdges launch; the kernel decodes them and writes tile_bins (every tile) for the aux / the backward.
hipError_t launch_rasterize(uint32_t w, uint32_t h, uint32_t tbx, uint32_t tby,
                            const uint32_t *compact_gid_from_isect, uint32_t *tile_bins, const uint32_t *bin_edges,
                            const float *projected, int raster_u32, uint32_t u32_pitch, void *out_img,
                            uint32_t *final_index, float *zero_rows /* nullable: [n][kCompactStride], the first
                            *num_visible rows are zeroed (BrushAux::bwd_accum) */,
                            const uint32_t *num_visible, uint32_t n, hipStream_t s);
// Zero-fill the compositing backward carries beside its arithmetic: the dense gradient arrays of the same backward
// (render.rs:539-547,573-575 zero-fills them with separate launches).  The kernel is bound by VALU issue and moves
// little memory, so its waves store the zeros in passing, one KiB (64 lanes x 16 B) per wave instruction, paced over
// the records they walk; the parameter VJP kernel behind it then writes the visible splats' rows only.
constexpr uint32_t kFillSegs = 6;
struct ZeroFill {
    float *base[kFillSegs];                // 16-byte aligned array starts
    uint32_t full[kFillSegs];              // whole 16-byte chunks of the array
    uint32_t tail[kFillSegs];              // floats behind them (0..3)
    uint32_t first_block[kFillSegs + 1];   // cumulative KiB blocks (the last block of an array may be partial)
    __host__ __device__ bool active() const { return first_block[kFillSegs] != 0u; }
};
// Fills `zf` for the given arrays (nullptr / 0 floats: skipped).  False — and an inactive `zf` — when an array is not
// 16-byte aligned or too large for 32-bit chunk counts: the caller then keeps the zeros in the VJP kernel.
bool make_zero_fill(ZeroFill *zf, float *const *arrays, const size_t *floats, uint32_t count);
hipError_t launch_rasterize_backward(uint32_t w, uint32_t h, uint32_t tbx, uint32_t tby,
                                     const uint32_t *compact_gid_from_isect, const uint32_t *tile_bins,
                                     const float *projected, const uint32_t *final_index,
                                     const float *out_img, const float *v_out, float *v_compact,
                                     const uint32_t *unsorted_pos /* deterministic mode, else nullptr */,
                                     float *rows /* deterministic mode: [max_intersects][12], else nullptr */,
                                     const ZeroFill &fill, hipStream_t s);

// project_bwd.hip
// Optimizer state for the fused backward + Adam form (brush_render_backward_adam).
struct AdamFuse {
    float *means, *log_scales, *rotation, *raw_opac, *sh;  // parameters, updated in place
    float *m1, *m2;                                        // [means 3N | log_scales 3N | quats 4N | raw_opac N | sh 3CN]
    float lr[5];                                           // means, log_scales, rotation, raw_opac, sh (dc)
    float sh_lerp, beta1, beta2, eps, rbc1, rbc2;          // rbc = 1 / (1 - beta^time), see adam_stepped()
    uint32_t quat_vjp, vec_ok;
    float *norm_rot_out;                                   // optional [N,4]: updated rotation / |rotation| (next forward's input)
    float *grad_2d_accum, *xy_grad_counts;                 // optional [N]: refinement statistics (train.rs:284-316)
    float half_w, half_h;
    float stat_scale;                                      // BrushAdamConfig::xy_stat_scale (0 -> 1)
    LazySh lazy;                                           // BrushAdamConfig::lazy_sh (off: every SH block is stepped)
};
hipError_t launch_lazy_sh_flush(const LazySh &lazy, float *sh, uint32_t n, uint32_t row_floats, hipStream_t s);
// Deterministic mode: where a splat's compact-order sums come from (see project_bwd.hip); partials == nullptr
// selects the default atomic accumulators in v_compact.
struct DetSumsArgs {
    const uint32_t *cum_tiles_hit = nullptr;
    const uint32_t *num_intersections = nullptr;
    const float *partials = nullptr;
    uint32_t cap = 0;
};
hipError_t launch_sum_isect_rows(const float *rows, const uint32_t *num_intersections, const uint32_t *cum_tiles_hit,
                                 uint32_t cap, float *v_compact, float *partials, hipStream_t s);
hipError_t launch_project_backward(const ViewParams &vp, const float *means, const float *log_scales,
                                   const float *quats, const float *raw_opac,
                                   const uint32_t *compact_from_global, const float *v_compact,
                                   float *v_means, float *v_xy,
                                   float *v_scales, float *v_quats, float *v_sh, float *v_opac,
                                   const AdamFuse *adam, const DetSumsArgs &det,
                                   bool prezeroed /* dense form only: the arrays are already zero (ZeroFill), the
                                   visible splats' rows alone are written */,
                                   hipStream_t s);
// View-sharded data parallelism (project_bwd.hip): per-view 64-byte gradient records, their index by global id and
// the deterministic per-splat sum over views (dense arrays, or straight into the Adam update when adam != nullptr).
hipError_t launch_project_backward_records(const ViewParams &vp, const float *means, const float *log_scales,
                                           const float *quats, const float *raw_opac, const uint32_t *num_visible,
                                           const uint32_t *global_from_compact, const float *v_compact,
                                           float *records, uint32_t max_rows, const DetSumsArgs &det, hipStream_t s);
hipError_t launch_reduce_view_records(const float *records, uint32_t num_views, uint32_t rows_per_view,
                                      const uint32_t *view_rows, const uint32_t *view_offsets /* nullable */,
                                      const float *campos, const float *means, uint32_t n,
                                      uint32_t sh_degree, uint32_t *index, float *v_means, float *v_scales,
                                      float *v_quats, float *v_sh, float *v_opac, const AdamFuse *adam, hipStream_t s);
hipError_t launch_zero_compact_grads(const uint32_t *num_visible, uint32_t n, float *v_compact, hipStream_t s);

}  // namespace brush
