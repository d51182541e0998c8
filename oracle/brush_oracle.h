/*
 * brush_oracle.h — CPU restatement of wartron/brush's splat-rasterizer hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under brush_amd/ may include, link or call this.
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg use it, as the
 * checker / reported CPU baseline, never as the product path.
 *
 * Parity status: PINNED against the reference's own golden vectors
 *   crates/brush-render/test_cases/{tiny_case,basic_case}.safetensors at the tolerances of
 *   crates/brush-render/src/render.rs:815-830 (tests/test_oracle_golden.py), and against the
 *   sort / scan vectors of crates/brush-sort/src/lib.rs:164-265 and
 *   crates/brush-prefix-sum/src/lib.rs:110-175.
 *
 * Every function cites the reference file:line it follows (paths relative to /root/reference).
 * All arithmetic is f32 with no FMA contraction (compile with -ffp-contract=off); the
 * transcendental functions the per-splat stages need (exp, log) are the deterministic
 * polynomial forms of oracle/detmath.h so that the integer outputs (visible set, depth order,
 * tile lists) can be compared bit-for-bit with the GPU path.
 */
#ifndef BRUSH_ORACLE_H
#define BRUSH_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* crates/brush-render/src/shaders/helpers.wgsl:7-30 (28 words, viewmat column-major). */
typedef struct OracleUniforms {
    float viewmat[16];
    float focal[2];
    uint32_t img_size[2];
    uint32_t tile_bounds[2];
    float pixel_center[2];
    uint32_t sh_degree;
    uint32_t num_visible;
    uint32_t total_splats;
    uint32_t padding;
} OracleUniforms;

/* Host-memory mirror of RenderAux (crates/brush-render/src/lib.rs:20-33). */
typedef struct OracleAux {
    float *projected_splats;           /* [N,9]  first V rows valid (helpers.wgsl:33-43) */
    uint32_t *num_intersections;       /* [1] */
    uint32_t *num_visible;             /* [1] */
    uint32_t *final_index;             /* [h,w] */
    uint32_t *cum_tiles_hit;           /* [N] inclusive scan, tail == I */
    uint32_t *tile_bins;               /* [ty,tx,2] */
    uint32_t *compact_gid_from_isect;  /* [max_intersects] */
    uint32_t *global_from_compact_gid; /* [N] tail filled with 0 */
    uint32_t *tile_id_from_isect;      /* [max_intersects] sorted tile ids (debug/test aid) */
    uint8_t *flip_risk;                /* [h,w] optional (may be NULL): 1 where a composite
                                          threshold test was within 1e-5 relative of flipping */
    uint32_t max_intersects;
} OracleAux;

/* crates/brush-sort/src/lib.rs:32-147 — stable argsort on the low 4*ceil(bits/4) key bits. */
void oracle_radix_argsort(const uint32_t *keys_in, const uint32_t *vals_in, uint32_t n,
                          uint32_t sorting_bits, uint32_t *keys_out, uint32_t *vals_out);

/* crates/brush-prefix-sum/src/lib.rs:17-102 — inclusive u32 scan. */
void oracle_inclusive_scan(const uint32_t *in, uint32_t n, uint32_t *out);

/* crates/brush-render/src/render.rs:55-323. out_img: float[h*w*4] or (raster_u32) uint32[h*w]. */
int oracle_render_forward(const OracleUniforms *u, const float *means, const float *log_scales,
                          const float *quats, const float *sh_coeffs, const float *raw_opac,
                          uint32_t n, int raster_u32, void *out_img, OracleAux *aux);

/* crates/brush-render/src/render.rs:468-626. All v_* are dense [N,..], zero for non-visible. */
int oracle_render_backward(const OracleUniforms *u, const OracleAux *aux, const float *means,
                           const float *log_scales, const float *quats, const float *raw_opac,
                           uint32_t n, const float *out_img, const float *v_out, float *v_means,
                           float *v_xy, float *v_scales, float *v_quats, float *v_sh,
                           float *v_opac,
                           /* optional compact-order intermediates (may be NULL): */
                           float *v_xy_local, float *v_conics, float *v_colors);

/* As above with the summation mode explicit: f32_sums = 1 also sums in f32 (pixels in tile order, a splat's
 * tiles in ascending intersection order): one admissible execution of the reference's f32 arithmetic. */
int oracle_render_backward_ex(const OracleUniforms *u, const OracleAux *aux, const float *means,
                              const float *log_scales, const float *quats, const float *raw_opac,
                              uint32_t n, const float *out_img, const float *v_out, float *v_means,
                              float *v_xy, float *v_scales, float *v_quats, float *v_sh, float *v_opac,
                              float *v_xy_local, float *v_conics, float *v_colors, int f32_sums);

/* brush_oracle_f64.c: forward compositing in f64 under the f32 restatement's decisions (pixel-tolerance arbiter) */
int oracle_rasterize_forward_f64(const OracleUniforms *u, const OracleAux *aux, double *out /* [h,w,4] */,
                                 double *cond /* optional [h,w]: sensitivity to relative errors of sigma's terms */);

/* brush_oracle_f64.c: the same backward with every value in f64 and the walk's decisions taken as the f32
 * restatement takes them; all outputs f64.  Arbiter for tolerance questions, never a parity reference itself. */
int oracle_render_backward_f64(const OracleUniforms *u, const OracleAux *aux, const float *means,
                               const float *log_scales, const float *quats, const float *raw_opac, uint32_t n,
                               const float *out_img, const float *v_out, double *v_means, double *v_xy,
                               double *v_scales, double *v_quats, double *v_sh, double *v_opac,
                               /* optional (all NULL or none): per-element sum of the magnitudes of the terms */
                               double *mag_means, double *mag_xy, double *mag_scales, double *mag_quats,
                               double *mag_sh, double *mag_opac,
                               /* and what threshold decisions within f32 rounding of flipping can move */
                               double *flip_means, double *flip_xy, double *flip_scales, double *flip_quats,
                               double *flip_sh, double *flip_opac,
                               /* and the magnitude of the terms the projection VJP itself sums */
                               double *vjp_means, double *vjp_scales, double *vjp_quats,
                               /* optional [h,w]: relative uncertainty of each pixel's forward state, added
                                * (times the magnitude of the pixel's terms) to flip_* */
                               const double *pix_weight,
                               /* optional [h,w]: that other forward state's final_index (entries only one of the two
                                * walks reach count towards flip_* in full) */
                               const uint32_t *final_index_alt,
                               /* mag_* with every term weighted by the number of divisions its T went through */
                               double *dep_means, double *dep_xy, double *dep_scales, double *dep_quats,
                               double *dep_sh, double *dep_opac,
                               /* rounding magnitude of the SH basis / of the saturating sigmoid */
                               double *vjp_sh, double *vjp_opac,
                               /* optional [h,w,4]: the image of the forward state final_index_alt belongs to */
                               const float *out_img_alt);

/* Deterministic elementary functions (oracle/detmath.h), exported for tests. */
float oracle_det_expf(float x);
float oracle_det_logf(float x);

int oracle_num_threads(void);
/* Y_k(dir) of gather_grads.wgsl:186-222 for n means: Y_out[n][(sh_degree + 1)^2] */
int oracle_sh_basis_for_means(const OracleUniforms *u, const float *means, uint32_t n, float *Y_out);

#ifdef __cplusplus
}
#endif
#endif
