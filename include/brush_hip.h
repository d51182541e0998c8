/*
 * brush_hip.h — C ABI of libbrush_hip.so: the MI355X (gfx950) replacement for the
 * splat-rasterizer hot path of wartron/brush.
 *
 * Drop-in boundary.  Each entry point names the reference interface it replaces
 * (paths relative to the reference checkout):
 *
 *   brush_render_forward    <- render_forward / Backend::render_splats
 *                              crates/brush-render/src/render.rs:55-323, src/lib.rs:66-86
 *   brush_render_backward   <- impl Backward<_,6> for RenderBackwards
 *                              crates/brush-render/src/render.rs:465-626
 *   brush_radix_argsort_u32 <- radix_argsort          crates/brush-sort/src/lib.rs:32-37
 *   brush_inclusive_scan_u32<- prefix_sum             crates/brush-prefix-sum/src/lib.rs:17
 *   *_workspace_size        <- create_tensor / client.empty scratch allocation
 *                              crates/brush-kernel/src/lib.rs:125-150
 *
 * Conventions
 *   - Every pointer is a DEVICE pointer unless its name starts with `h_`.
 *   - The caller owns all memory (inputs, outputs, aux, workspace).  The library never
 *     allocates, frees or synchronises on these paths; all work is enqueued on `stream`
 *     (a hipStream_t) and the call returns immediately.  Data-dependent sizes
 *     (num_visible, num_intersections) stay on the device.
 *   - Functions are re-entrant: no global mutable state, per-call workspace; every mode switch is an argument
 *     (BrushAux::flags), nothing on these paths reads the environment.
 *   - Return value: BRUSH_OK or a negative BrushStatus; nothing aborts.
 *   - All floating point is f32, all indices/counts u32 (i32-typed tensors in Burn).
 */
#ifndef BRUSH_HIP_H
#define BRUSH_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void *brush_stream_t; /* hipStream_t */

typedef enum BrushStatus {
    BRUSH_OK = 0,
    BRUSH_ERR_INVALID_ARG = -1,    /* bad shape / null pointer / sh_degree > 4 / bits > 32 */
    BRUSH_ERR_WORKSPACE_SMALL = -2,
    BRUSH_ERR_HIP = -3,            /* a HIP call failed; see brush_last_hip_error() */
    BRUSH_ERR_NO_DEVICE = -4
} BrushStatus;

/* RenderUniforms, 28 words — crates/brush-render/src/shaders/helpers.wgsl:7-30, filled like
 * render.rs:102-116.  viewmat is world->camera, column-major.  num_visible is an OUTPUT of the
 * forward pass inside aux.uniforms_buffer (word 25, render.rs:145-149); the host copy passed
 * in is read-only and its num_visible/total_splats/padding fields are ignored. */
typedef struct BrushUniforms {
    float viewmat[16];
    float focal[2];
    uint32_t img_size[2]; /* (w, h) */
    uint32_t tile_bounds[2];
    float pixel_center[2];
    uint32_t sh_degree;
    uint32_t num_visible;
    uint32_t total_splats;
    uint32_t padding;
} BrushUniforms;

#define BRUSH_TILE_WIDTH 16u
#define BRUSH_PROJECTED_FLOATS 9u /* ProjectedSplat, helpers.wgsl:33-43 */

/* Deferred Adam for the spherical-harmonics block (build extension of the training harness; optional everywhere it
 * appears).  81 % of the optimizer's bytes at SH degree 3 are the coefficients and their two moments (48 of 59 floats
 * per splat), yet the forward reads a splat's coefficients only when the splat passes the cull, and a splat that is
 * not visible has a zero gradient.  With this state attached, the optimizer steps of a splat's SH block that carry a
 * zero gradient are not applied when they happen; they stay pending and are replayed - the same float operations in the
 * same order as burn's Adam::step would have run them, step by step, so the values are bit-identical to the eager
 * optimizer - when the block is next needed: by the forward in registers (a pure read: the forward still modifies none
 * of its inputs), by the fused backward before it applies the step that does carry a gradient, or by
 * brush_lazy_sh_flush.  sh_time[g] is the optimizer time the stored SH block of splat g is current for.
 * `table` row i holds (1 / (1 - beta1^t), 1 / (1 - beta2^t), lr_coeffs_dc, sh_rest_lerp) of optimizer time
 * t = base + 1 + i, as brush_lazy_sh_fill_table computes them - the code that serves BrushAdamConfig::time in the fused
 * eager step, so a replayed step uses the very floats the eager step was given; every pending time must lie inside the
 * table: flush before `now` leaves it.  (Fused forms: the update is x - lr (m rbc1) rcp(sqrt(v rbc2) + eps) with
 * v_sqrt_f32 / v_rcp_f32, 1 ulp each; WGSL specifies its division to 2.5 ulp.) */
typedef struct BrushLazySh {
    const float *table;   /* device, [capacity][4] */
    uint32_t base;        /* optimizer time of row 0, minus 1 */
    uint32_t capacity;
    uint32_t now;         /* optimizer steps taken so far (the eager optimizer's `time` after its last step) */
    uint32_t *sh_time;    /* device, [N]; <= now */
    float *sh_moment1;    /* device, [N][C][3]: the SH segment of moment1 (moment1 + 11 N) */
    float *sh_moment2;
    float beta1, beta2, epsilon;
} BrushLazySh;

/* Device-pointer mirror of RenderAux (crates/brush-render/src/lib.rs:20-33).  All buffers are
 * caller-allocated with the shapes below and must stay alive and unmodified between
 * brush_render_forward and brush_render_backward (render.rs:436-446). */
typedef struct BrushAux {
    float *projected_splats;           /* [N,9] f32; rows < num_visible valid, compact order */
    uint32_t *uniforms_buffer;         /* [28]  written by forward (num_visible at word 25) */
    uint32_t *num_intersections;       /* [1]   = min(sum tiles hit, max_intersects) */
    uint32_t *num_visible;             /* [1] */
    uint32_t *final_index;             /* [h,w] last contributing isect id (0 if none) */
    uint32_t *cum_tiles_hit;           /* [N]   inclusive scan of tiles hit; tail == total */
    uint32_t *tile_bins;               /* [ty,tx,2] [start,end) into compact_gid_from_isect */
    uint32_t *compact_gid_from_isect;  /* [max_intersects] sorted by (tile, depth) */
    uint32_t *global_from_compact_gid; /* [N] depth order; entries >= num_visible are 0 */
    uint32_t *compact_from_global_gid; /* [N] inverse of global_from_compact_gid, 0xFFFFFFFF for
                                          non-visible splats (build extension: lets the backward
                                          write each dense gradient exactly once) */
    uint32_t *overflow;                /* [1] set to 1 when intersections were truncated at
                                          max_intersects (the reference truncates silently,
                                          map_gaussian_to_intersects.wgsl:40) */
    uint32_t max_intersects;           /* capacity; reference: min(N*tiles, 128*65535)
                                          (render.rs:204-206) */
    uint32_t *isect_unsorted_pos;      /* [max_intersects] or NULL.  Deterministic mode (flags &
                                          BRUSH_AUX_DETERMINISTIC) only: position each sorted intersection had in
                                          emission order (grouped by splat), written by the forward and used by
                                          the backward to sum a splat's per-tile gradient rows in a fixed order.
                                          Must be non-NULL in that mode, ignored otherwise. */
    uint32_t flags;                    /* BRUSH_AUX_* bits, chosen PER CALL; the forward and the backward of one
                                          render must be given the same BRUSH_AUX_DETERMINISTIC bit */
    float *bwd_accum;                  /* NULL, or the buffer the caller will pass as `workspace` to the backward of
                                          this render (brush_bwd_workspace_size* bytes).  Default mode only: the
                                          forward's last kernel then also zeroes the backward's per-splat accumulator
                                          rows (the first num_visible 64-byte rows of that buffer), and a backward
                                          called with BRUSH_AUX_ACCUM_ZEROED skips its zero-fill launch.  The
                                          reference zero-fills inside the backward (render.rs:505-507); at 100 k
                                          visible splats that launch is 6 us of a 345 us step for 6.6 MB of stores the
                                          VALU-bound compositing kernel carries for free. */
    const BrushLazySh *lazy_sh;        /* NULL, or the deferred-Adam state of the SH coefficients (host pointer, read
                                          during the call only): the forward then evaluates a visible splat's colour
                                          from its coefficients with the pending zero-gradient steps replayed in
                                          registers.  `sh_coeffs` itself is not written. */
} BrushAux;

/* BrushAux::flags */
#define BRUSH_AUX_DETERMINISTIC 1u /* bitwise reproducible gradients: the compositing backward stores one gradient
                                      row per intersection and sums a splat's rows in a fixed order instead of
                                      using hardware float atomics (whose arrival order is unspecified; the
                                      reference's CAS queue, rasterize_backwards.wgsl:276-301, has the same
                                      nondeterminism).  Needs isect_unsorted_pos and the larger backward workspace
                                      of brush_bwd_workspace_size_flags. */
#define BRUSH_AUX_ACCUM_ZEROED 2u  /* backward only, default mode: `workspace` == aux.bwd_accum of the forward of this
                                      render, nothing has written to it since, and no backward has consumed it yet
                                      (the FIRST backward after that forward): the accumulators are already zero.  A
                                      second backward of the same forward must clear the bit (it zero-fills itself). */

/* ---- introspection ------------------------------------------------------------------ */
const char *brush_version(void);
const char *brush_status_string(int status);
/* hipError_t of the last failing HIP call on this host thread (0 if none). */
int brush_last_hip_error(void);
/* Reference limit helper: min(N * tiles, 128*65535), at least 1 (render.rs:204-206). */
uint32_t brush_default_max_intersects(uint32_t n, uint32_t w, uint32_t h);

/* ---- radix argsort (brush-sort) ----------------------------------------------------- */
/* Stable LSD argsort of the first *d_n (device scalar, <= max_n) key/value pairs on the low
 * 4*ceil(bits/4) key bits (the reference runs ceil(bits/4) 4-bit passes,
 * brush-sort/src/lib.rs:58).  Inputs are not modified; outputs hold the sorted pairs in
 * [0, *d_n); elements beyond are unspecified (sort_scatter.wgsl:118-121). */
int brush_radix_argsort_workspace_size(uint32_t max_n, size_t *bytes);
int brush_radix_argsort_u32(const uint32_t *keys_in, const uint32_t *vals_in, uint32_t *keys_out,
                            uint32_t *vals_out, const uint32_t *d_n, uint32_t max_n,
                            uint32_t sorting_bits, void *workspace, size_t workspace_bytes,
                            brush_stream_t stream);

/* ---- prefix sum (brush-prefix-sum) ---------------------------------------------------- */
/* out[i] = in[0] + ... + in[i] (wrapping u32), n known on the host like the reference's
 * tensor shape (brush-prefix-sum/src/lib.rs:19). in == out is allowed. */
int brush_inclusive_scan_workspace_size(uint32_t n, size_t *bytes);
int brush_inclusive_scan_u32(const uint32_t *in, uint32_t *out, uint32_t n, void *workspace,
                             size_t workspace_bytes, brush_stream_t stream);

/* ---- render forward ------------------------------------------------------------------- */
int brush_fwd_workspace_size(uint32_t n, uint32_t w, uint32_t h, uint32_t sh_degree,
                             uint32_t max_intersects, size_t *bytes);
/* means[N,3] log_scales[N,3] quats[N,4] (w,x,y,z; already normalised) sh_coeffs[N,C,3]
 * raw_opacity[N].  out_img: raster_u32 == 0 -> float[h,w,4] (rgb, 1-T); != 0 -> uint32[h,w]
 * packed RGBA8 (rasterize.wgsl:106-109) and aux->final_index is not written. */
int brush_render_forward(const BrushUniforms *h_uniforms, const float *means,
                         const float *log_scales, const float *quats, const float *sh_coeffs,
                         const float *raw_opacity, uint32_t n, int raster_u32, void *out_img,
                         const BrushAux *h_aux, void *workspace, size_t workspace_bytes,
                         brush_stream_t stream);

/* Forward-only display path: packed RGBA8 (rasterize.wgsl:106-109) written
 * with rows `row_pitch_pixels` pixels apart, so the viewer's texture upload needs no padding
 * copy.  The reference pads the [h,w] u32 image into a zero [h, ceil(w/64)*64] tensor because
 * WebGPU wants bytes_per_row % 256 == 0 (crates/brush-ui/src/burn_texture.rs:17-26);
 * brush_rgba8_row_pitch(w) returns that pitch.  out_img: [h * row_pitch_pixels] u32; columns
 * >= w of each row are left untouched.  aux.final_index may be NULL (not written, as in the
 * reference's RASTER_U32 variant).  Otherwise identical to brush_render_forward(raster_u32=1). */
uint32_t brush_rgba8_row_pitch(uint32_t width);
int brush_render_forward_rgba8(const BrushUniforms *uniforms, const float *means, const float *log_scales,
                               const float *quats, const float *sh_coeffs, const float *raw_opacity,
                               uint32_t n, uint32_t *out_img, uint32_t row_pitch_pixels, const BrushAux *aux,
                               void *workspace, size_t workspace_bytes, brush_stream_t stream);

/* ---- render backward ------------------------------------------------------------------ */
/* A host-side DEFAULT for BrushAux::flags, nothing more: 1 when the environment holds BRUSH_DETERMINISTIC=1 (read
 * on every call, never cached, never consulted by the render entry points themselves).  A host that wants the
 * mode per call (a viewer thread beside a trainer task) sets or clears BRUSH_AUX_DETERMINISTIC itself. */
int brush_deterministic(void);
/* Backward workspace for a call with these BrushAux::flags and this intersection capacity (the deterministic mode
 * keeps 64 bytes per intersection). */
int brush_bwd_workspace_size_flags(uint32_t n, uint32_t w, uint32_t h, uint32_t sh_degree, uint32_t max_intersects,
                                   uint32_t flags, size_t *bytes);
/* Shorthands: flags = 0 (the default mode); brush_bwd_workspace_size also assumes
 * brush_default_max_intersects(n, w, h). */
int brush_bwd_workspace_size(uint32_t n, uint32_t w, uint32_t h, uint32_t sh_degree,
                             size_t *bytes);
int brush_bwd_workspace_size_ex(uint32_t n, uint32_t w, uint32_t h, uint32_t sh_degree, uint32_t max_intersects,
                                size_t *bytes);
/* Gradients in the parent order of render.rs:420-427,598-624:
 * v_means[N,3] v_xy[N,2] (global order, pixel units) v_scales[N,3] (log-space)
 * v_quats[N,4] v_sh[N,C,3] v_opac[N] — dense, every element written, 0 for non-visible
 * splats.  out_img / v_out are float[h,w,4].  The six gradient arrays are written from the first kernel of the call on
 * (their zeros ride on the compositing backward): they must not overlap any input of the call, the aux buffers or the
 * workspace — as separately allocated Burn tensors never do. */
int brush_render_backward(const BrushUniforms *h_uniforms, const BrushAux *h_aux,
                          const float *means, const float *log_scales, const float *quats,
                          const float *raw_opacity, uint32_t n, const float *out_img,
                          const float *v_out, float *v_means, float *v_xy, float *v_scales,
                          float *v_quats, float *v_sh, float *v_opac, void *workspace,
                          size_t workspace_bytes, brush_stream_t stream);

/* ---- view-sharded data parallelism (build extension; the reference is single-device, batch 1:
 *      crates/brush-train/src/train.rs:216-219) ------------------------------------------------ */
/* One process per GPU renders one view of the replicated splats; the step needs the SUM over views of the
 * parameter gradients.  A view's gradient is non-zero only for its visible splats and its SH row is rank one
 * (v_sh[g] = Y(dir_view(g)) (x) v_rgb[g], gather_grads.wgsl:186-222), so it is exchanged as one 64-byte record
 * per VISIBLE splat (16 f32, compact = depth order):
 *   [gid as u32 bits | v_means(3) | v_scales(3) | v_quats(4) | v_opac | v_rgb(3) | |v_xy * (w/2, h/2)|]
 * instead of 52+12C bytes per splat.  The caller all-gathers the records (RCCL; brush_amd/dist.py), then the
 * per-splat sum over views runs in view order 0..W-1 without atomics: bit-identical on every rank and from run
 * to run, so replicated parameters stay replicated.
 *
 * brush_render_backward_records: brush_render_backward without the dense outputs.  Writes rows
 * c < min(num_visible, max_rows) of `records` ([max_rows][16], 16-byte aligned); rows beyond max_rows are DROPPED:
 * size max_rows from the view's num_visible (known after the forward).  Same workspace as brush_render_backward. */
int brush_render_backward_records(const BrushUniforms *h_uniforms, const BrushAux *h_aux, const float *means,
                                  const float *log_scales, const float *quats, const float *raw_opacity, uint32_t n,
                                  const float *out_img, const float *v_out, float *records, uint32_t max_rows,
                                  void *workspace, size_t workspace_bytes, brush_stream_t stream);
/* Scratch of the reduction: num_views * n u32 (row of splat g in view v's records).  Results never depend on its
 * contents (entries are validated against the record they point to), but set it to 0xFF bytes once before the first
 * use: the reduction clears the entries it consumes, so from then on an entry nobody wrote this step is recognised
 * without a gather.  No reset between steps. */
int brush_view_index_size(uint32_t n, uint32_t num_views, size_t *bytes);
/* Sum over views -> dense gradients.  records: [num_views][rows_per_view][16]; view v owns its first
 * view_rows[v] rows (device array [num_views], values > rows_per_view are clamped).  view_offsets (device array
 * [num_views], or NULL): when given the views are PACKED — view v owns rows [view_offsets[v], + view_rows[v]) of a
 * buffer of rows_per_view rows in all — which is what an all-gather of exactly num_visible records per view leaves
 * (the padded form moves num_views x the largest view).  campos: [num_views][3] =
 * viewmat[3].xyz of each view (the term the reference uses as camera position, project_visible.wgsl:232-233);
 * means: [N,3].  Every element of v_means [N,3] v_scales [N,3] v_quats [N,4] v_sh [N,C,3] v_opac [N] is written
 * (0 for splats no view sees). */
int brush_reduce_view_records(const float *records, uint32_t num_views, uint32_t rows_per_view,
                              const uint32_t *view_rows, const uint32_t *view_offsets, const float *campos,
                              const float *means, uint32_t n, uint32_t sh_degree, float *v_means, float *v_scales,
                              float *v_quats, float *v_sh, float *v_opac, void *view_index, size_t view_index_bytes,
                              brush_stream_t stream);

/* ---- training iteration around the op (build extension) -------------------- */
/* The reference's SplatTrainer::step (crates/brush-train/src/train.rs:211-393) wraps the op in
 * Burn tensor ops: image loss, Adam, gradient statistics.  These entry points are the fused
 * HIP equivalents, so the metric's train iters/s is not bounded by caller-side launches. */

/* loss = mean|pred_cmp - gt| * (1 - ssim_weight) - SSIM(pred_rgb, gt_rgb) * ssim_weight when
 * ssim_weight > 0, else the plain L1 mean (train.rs:243-268); SSIM as ssim.rs:42-101 (Gaussian
 * window, sigma 1.5, zero padding div_ceil(window, 2) so the SSIM map is (h+2)x(w+2), variances clamped at 0).
 * pred: [h,w,4]; gt: [h,w,gt_channels], gt_channels 3 or 4 (4 compares alpha too, train.rs:248-252).
 * Writes loss[0] (device) and v_pred [h,w,4] = grad_scale * d loss / d pred.  ssim_window: odd sizes 3..15
 * (TrainConfig::ssim_window_size, train.rs:63, default 11); anything else returns BRUSH_ERR_INVALID_ARG. */
int brush_loss_workspace_size(uint32_t w, uint32_t h, size_t *bytes);
int brush_l1_ssim_loss(const float *pred, const float *gt, uint32_t w, uint32_t h, uint32_t gt_channels,
                       float ssim_weight, uint32_t ssim_window, float grad_scale, float *loss, float *v_pred,
                       void *workspace, size_t workspace_bytes, brush_stream_t stream);

/* Hyper-parameters of one optimizer step: the five learning rates of train.rs:275-282, the lerp
 * factor 1/lr_coeffs_sh_scale for SH coefficients >= 1 (train.rs:336-351), Adam betas/epsilon
 * (AdamConfig::new().with_epsilon(1e-15), train.rs:184) and the 1-based step count. */
typedef struct BrushAdamConfig {
    float lr_mean, lr_scale, lr_rotation, lr_opac, lr_coeffs_dc, sh_rest_lerp;
    float beta1, beta2, epsilon;
    uint32_t time;
    /* 1: `quats` holds the raw rotation parameter and v_quats is the gradient wrt rotation/|rotation|
     * (what Splats::render feeds the op, gaussian_splats.rs:174-175); the chain rule through the
     * normalisation is applied before the moment update.  0: v_quats is used as is. */
    uint32_t rotation_grad_wrt_normalized;
    /* Multiplies the screen-space statistic |v_xy * (w/2, h/2)| before it is added to grad_2d_accum (fused forms only).
     * With B views per step the upstream gradient carries the 1/B of the mean over the batch (train.rs:239-268), so the
     * statistic the densification threshold is compared with (train.rs:284-316, tuned for B = 1) would shrink B-fold:
     * pass B here to keep the reference's magnitude.  0 is read as 1. */
    float xy_stat_scale;
    /* Fused forms only (brush_render_backward_adam, brush_reduce_view_records_adam): NULL, or the deferred-Adam state
     * of the SH block.  The SH coefficients and moments of splats the view (no view of the batch) sees are then left
     * alone (their step stays pending); a seen splat's block first has its pending steps replayed, then takes this
     * step, and its sh_time becomes `time`.  Requires lazy_sh->now + 1 == time and 3 C floats per row a multiple of 4
     * (SH degree 1 or 3). */
    const BrushLazySh *lazy_sh;
} BrushAdamConfig;
/* One Adam step on all five parameter groups in one launch.  v_*: the gradient arrays of
 * brush_render_backward; moment1 / moment2: N*(11+3C) floats each, owned by the caller, laid out
 * [means 3N | log_scales 3N | quats 4N | raw_opac N | sh 3CN] and zero before the first step.
 * Parameters are updated in place.  Every optimizer entry point of this header computes the update as
 * x - lr (m rbc1) rcp(sqrt(v rbc2) + eps), rbc = 1 / (1 - beta^time) formed on the host, with v_sqrt_f32 / v_rcp_f32
 * (1 ulp each; WGSL, which the reference's optimizer runs in, allows 2.5 ulp on a division) and no FMA contraction:
 * from the same gradients the separate calls, the fused call and its deferred-SH form leave the same bits. */
int brush_adam_step(const BrushAdamConfig *cfg, uint32_t n, uint32_t sh_degree, float *means, float *log_scales,
                    float *quats, float *raw_opac, float *sh, const float *v_means, const float *v_scales,
                    const float *v_quats, const float *v_opac, const float *v_sh, float *moment1,
                    float *moment2, brush_stream_t stream);
/* brush_render_backward and brush_adam_step in one pass: the projection-backward kernel sends every
 * parameter-gradient element straight through the optimizer update instead of storing it, so the
 * dense gradients (52+12C bytes per splat) never travel to HBM and back.  Same arguments as the two
 * calls it replaces: `quats_fed` is the [N,4] array the forward was fed (rotation/|rotation| when
 * cfg->rotation_grad_wrt_normalized), `rotation` the raw parameter; means / log_scales / rotation /
 * raw_opacity / sh are updated in place, v_xy [N,2] is still written.  Optional outputs (NULL to
 * skip): next_quats_fed [N,4] = updated rotation / |rotation| (what the next forward is fed, saves
 * brush_normalize_quats; must not alias quats_fed); grad_2d_accum / xy_grad_counts [N] updated as
 * brush_refine_stats does.  Single-view training only: data-parallel training needs the gradients
 * (brush_render_backward). */
int brush_render_backward_adam(const BrushUniforms *uniforms, const BrushAux *aux, const BrushAdamConfig *cfg,
                               float *means, float *log_scales, const float *quats_fed, float *rotation,
                               float *raw_opacity, float *sh, uint32_t n, const float *out_img,
                               const float *v_out, float *v_xy, float *moment1, float *moment2,
                               float *next_quats_fed, float *grad_2d_accum, float *xy_grad_counts,
                               void *workspace, size_t workspace_bytes, brush_stream_t stream);
/* brush_reduce_view_records and brush_adam_step in one pass (data-parallel counterpart of
 * brush_render_backward_adam): the summed gradients go straight through the optimizer update, every rank applies
 * the same bits.  means / log_scales / rotation / raw_opacity / sh are updated in place (means is also the source
 * of the SH view directions: each splat is read before it is written).  width / height: image size of the
 * statistics (train.rs:300-302).  Optional (NULL to skip): next_quats_fed [N,4]; grad_2d_accum / xy_grad_counts
 * [N] += sum over views of the record's |v_xy * (w/2, h/2)| / number of views that saw the splat
 * (train.rs:284-316 for a batch of views). */
int brush_reduce_view_records_adam(const float *records, uint32_t num_views, uint32_t rows_per_view,
                                   const uint32_t *view_rows, const uint32_t *view_offsets, const float *campos,
                                   const BrushAdamConfig *cfg,
                                   uint32_t width, uint32_t height, float *means, float *log_scales, float *rotation,
                                   float *raw_opacity, float *sh, uint32_t n, uint32_t sh_degree, float *moment1,
                                   float *moment2, float *next_quats_fed, float *grad_2d_accum, float *xy_grad_counts,
                                   void *view_index, size_t view_index_bytes, brush_stream_t stream);
/* Applies every pending step of every splat's SH block (sh [N][C][3] and the two SH moment segments of `lazy`) and sets
 * sh_time[g] = lazy->now for all g: afterwards sh / moments are what the eager optimizer would hold.  Call it before
 * anything reads the coefficients without going through the op (export, refinement, a switch back to the eager step). */
int brush_lazy_sh_flush(const BrushLazySh *lazy, float *sh, uint32_t n, uint32_t sh_degree, brush_stream_t stream);
/* Host helper: rows [capacity][4] of BrushLazySh::table for optimizer times base+1 .. base+capacity, computed by the
 * code that turns BrushAdamConfig::time into the bias corrections of an eager step (so both use the same floats). */
int brush_lazy_sh_fill_table(float beta1, float beta2, float lr_coeffs_dc, float sh_rest_lerp, uint32_t base,
                             uint32_t capacity, float *host_rows);
/* normalized[i] = rotation[i] / |rotation[i]| (gaussian_splats.rs:174-175); [N,4], 16-byte aligned. */
int brush_normalize_quats(const float *rotation, float *normalized, uint32_t n, brush_stream_t stream);
/* train.rs:284-316: grad_2d_accum[g] += |v_xy[g] * (w/2, h/2)|; xy_grad_counts[g] += 1 for every
 * visible splat g of this view (both [N] f32). */
int brush_refine_stats(const BrushAux *h_aux, const float *v_xy, uint32_t n, uint32_t w, uint32_t h,
                       float *grad_2d_accum, float *xy_grad_counts, brush_stream_t stream);

/* ---- opt-in stage timing ---------------------------------------------------------------- */
/* Counterpart of the reference's tracing spans + sync-span layer (render.rs:69-267,474-577;
 * crates/sync-span/src/lib.rs:12-49): when a profiler is attached to the calling host thread,
 * brush_render_forward / brush_render_backward record a hipEvent on `stream` after every
 * stage.  Nothing is recorded (and no event exists) when no profiler is attached, so the
 * default path stays free of events and synchronisation. */
typedef struct BrushProfiler BrushProfiler;
enum {
    BRUSH_STAGE_PROJECT_CULL = 0, /* init + ProjectSplats + compaction          (fwd) */
    BRUSH_STAGE_DEPTH_SORT,       /* radix argsort of depth keys                (fwd) */
    BRUSH_STAGE_PROJECT_VISIBLE,  /* ProjectVisible                             (fwd) */
    BRUSH_STAGE_PREFIX_SUM,       /* cum_tiles_hit                              (fwd) */
    BRUSH_STAGE_MAP_INTERSECTS,   /* MapGaussiansToIntersect                    (fwd) */
    BRUSH_STAGE_TILE_SORT,        /* radix argsort of tile ids                  (fwd) */
    BRUSH_STAGE_TILE_BINS,        /* GetTileBinEdges                            (fwd) */
    BRUSH_STAGE_RASTERIZE,        /* Rasterize                                  (fwd) */
    BRUSH_STAGE_BWD_ZERO,         /* zero compact-order accumulators            (bwd) */
    BRUSH_STAGE_RASTERIZE_BWD,    /* RasterizeBackwards                         (bwd) */
    BRUSH_STAGE_PROJECT_BWD,      /* GatherGrads + ProjectBackwards (fused)     (bwd) */
    BRUSH_NUM_STAGES
};
int brush_profiler_create(BrushProfiler **out);
void brush_profiler_destroy(BrushProfiler *p);
/* Attach (or detach with NULL) a profiler to the calling host thread. */
void brush_profiler_attach(BrushProfiler *p);
/* Blocks until the recorded events have completed, then writes the milliseconds spent in each
 * stage of the LAST forward and LAST backward call recorded (0 for stages not recorded). */
int brush_profiler_read(BrushProfiler *p, float *h_ms /* [BRUSH_NUM_STAGES] */);
/* Measurement only: while `p` is attached, brush_render_forward* / brush_render_backward* return BRUSH_OK right after
 * the launches of `stage` have been enqueued (later stages are not launched and their outputs stay untouched) and
 * record no events; stage = -1 restores the whole pass.  Timing the captured prefixes 0..k of a step and taking
 * differences gives every stage's time IN SITU, without the ~3-5 us an event record adds to each stage of an eager
 * pass (bench.py: `stage_ms`; the event times stay available as `stage_ms_events`). */
int brush_profiler_stop_after(BrushProfiler *p, int stage);
const char *brush_stage_name(int stage);

#ifdef __cplusplus
}
#endif
#endif /* BRUSH_HIP_H */
