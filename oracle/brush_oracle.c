/*
 * brush_oracle.c — CPU restatement of wartron/brush's forward+backward splat rasterizer.
 *
 * TEST INFRASTRUCTURE ONLY (see brush_oracle.h).  Paths cited are relative to /root/reference.
 * f32 arithmetic throughout, summation orders fixed as written (compile with
 * -ffp-contract=off), OpenMP over splats / tiles.
 *
 * Matrix convention: all matrices below are plain row-major m[row][col].  WGSL is
 * column-major (m[col][row]); every transcription notes the swap where it matters.
 */
#include "brush_oracle.h"
#include "detmath.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define TILE_WIDTH 16u  /* helpers.wgsl:1 */
#define TILE_SIZE 256u  /* helpers.wgsl:3 */
#define COV_BLUR 0.3f   /* helpers.wgsl:166 */

int oracle_num_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

float oracle_det_expf(float x) { return det_expf(x); }
float oracle_det_logf(float x) { return det_logf(x); }

/* ---------------------------------------------------------------- small helpers */

static inline float clampf(float x, float lo, float hi) { return fminf(fmaxf(x, lo), hi); }

/* WGSL i32(f32): truncate toward zero, saturate; NaN -> 0 (SURVEY §2c "WGSL semantics"). */
static inline int32_t f2i_sat(float x) {
    if (x != x) return 0;
    if (x >= 2147483648.0f) return INT32_MAX;
    if (x <= -2147483648.0f) return INT32_MIN;
    return (int32_t)x;
}
/* WGSL u32(f32). */
static inline uint32_t f2u_sat(float x) {
    if (!(x > 0.0f)) return 0u;
    if (x >= 4294967296.0f) return UINT32_MAX;
    return (uint32_t)x;
}
static inline int32_t iclamp(int32_t v, int32_t lo, int32_t hi) {
    return v < lo ? lo : (v > hi ? hi : v);
}
static inline float signf(float x) { return x > 0.0f ? 1.0f : (x < 0.0f ? -1.0f : 0.0f); }

typedef struct { float m[3][3]; } mat3;

static inline mat3 mat3_mul(mat3 a, mat3 b) {
    mat3 c;
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++)
            c.m[i][j] = a.m[i][0] * b.m[0][j] + a.m[i][1] * b.m[1][j] + a.m[i][2] * b.m[2][j];
    return c;
}
static inline mat3 mat3_transpose(mat3 a) {
    mat3 c;
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) c.m[i][j] = a.m[j][i];
    return c;
}
static inline mat3 mat3_add(mat3 a, mat3 b) {
    mat3 c;
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) c.m[i][j] = a.m[i][j] + b.m[i][j];
    return c;
}

/* W = mat3x3f(viewmat[0].xyz, viewmat[1].xyz, viewmat[2].xyz): W[r][c] = viewmat[c*4+r]. */
static inline mat3 view_rot(const float *vm) {
    mat3 w;
    for (int r = 0; r < 3; r++)
        for (int c = 0; c < 3; c++) w.m[r][c] = vm[c * 4 + r];
    return w;
}

/* p_view = W * mean + viewmat[3].xyz (project_forward.wgsl:29-30). */
static inline void to_view(const float *vm, const float *mean, float *p) {
    mat3 w = view_rot(vm);
    for (int r = 0; r < 3; r++)
        p[r] = (w.m[r][0] * mean[0] + w.m[r][1] * mean[1] + w.m[r][2] * mean[2]) + vm[12 + r];
}

/* helpers.wgsl:74-109.  quat = (w,x,y,z) in .x .y .z .w. */
static inline mat3 quat_to_rotmat(const float *q) {
    float w = q[0], x = q[1], y = q[2], z = q[3];
    float x2 = x * x, y2 = y * y, z2 = z * z;
    float xy = x * y, xz = x * z, yz = y * z;
    float wx = w * x, wy = w * y, wz = w * z;
    mat3 r;
    /* WGSL columns -> r.m[row][col] */
    r.m[0][0] = 1.0f - 2.0f * (y2 + z2);
    r.m[1][0] = 2.0f * (xy + wz);
    r.m[2][0] = 2.0f * (xz - wy);
    r.m[0][1] = 2.0f * (xy - wz);
    r.m[1][1] = 1.0f - 2.0f * (x2 + z2);
    r.m[2][1] = 2.0f * (yz + wx);
    r.m[0][2] = 2.0f * (xz + wy);
    r.m[1][2] = 2.0f * (yz - wx);
    r.m[2][2] = 1.0f - 2.0f * (x2 + y2);
    return r;
}

/* helpers.wgsl:119-122 */
static inline void project_pix(const float *f, const float *p, const float *pp, float *xy) {
    xy[0] = (p[0] / p[2]) * f[0] + pp[0];
    xy[1] = (p[1] / p[2]) * f[1] + pp[1];
}

/* helpers.wgsl:124-158.  Returns (c00, c01, c11). */
static inline void calc_cov2d(const OracleUniforms *u, const float *p_view, const float *scale,
                              const float *quat, float *cov2d) {
    const float *focal = u->focal;
    float img[2] = {(float)u->img_size[0], (float)u->img_size[1]};
    float tan_fov[2] = {0.5f * img[0] / focal[0], 0.5f * img[1] / focal[1]};
    float lims_pos[2], lims_neg[2];
    for (int i = 0; i < 2; i++) {
        lims_pos[i] = (img[i] - u->pixel_center[i]) / focal[i] + 0.3f * tan_fov[i];
        lims_neg[i] = u->pixel_center[i] / focal[i] + 0.3f * tan_fov[i];
    }
    float rz = 1.0f / p_view[2];
    float rz2 = rz * rz;
    float t[2];
    for (int i = 0; i < 2; i++)
        t[i] = p_view[2] * clampf(p_view[i] * rz, -lims_neg[i], lims_pos[i]);

    mat3 M = quat_to_rotmat(quat);
    for (int r = 0; r < 3; r++)
        for (int c = 0; c < 3; c++) M.m[r][c] = M.m[r][c] * scale[c];

    /* J = mat3x2f((fx*rz,0),(0,fy*rz),-focal*t*rz2): 2 rows x 3 cols. */
    float j00 = focal[0] * rz, j11 = focal[1] * rz;
    float j02 = (-focal[0]) * t[0] * rz2, j12 = (-focal[1]) * t[1] * rz2;

    mat3 W = view_rot(u->viewmat);
    mat3 V = mat3_mul(M, mat3_transpose(M));
    float T[2][3], TV[2][3];
    for (int c = 0; c < 3; c++) {
        T[0][c] = (j00 * W.m[0][c] + 0.0f * W.m[1][c]) + j02 * W.m[2][c];
        T[1][c] = (0.0f * W.m[0][c] + j11 * W.m[1][c]) + j12 * W.m[2][c];
    }
    for (int r = 0; r < 2; r++)
        for (int c = 0; c < 3; c++)
            TV[r][c] = T[r][0] * V.m[0][c] + T[r][1] * V.m[1][c] + T[r][2] * V.m[2][c];
    /* cov = (T V) T^T; WGSL cov[0][1] is column 0 row 1. */
    float cov00 = TV[0][0] * T[0][0] + TV[0][1] * T[0][1] + TV[0][2] * T[0][2];
    float cov10 = TV[1][0] * T[0][0] + TV[1][1] * T[0][1] + TV[1][2] * T[0][2];
    float cov11 = TV[1][0] * T[1][0] + TV[1][1] * T[1][1] + TV[1][2] * T[1][2];
    cov2d[0] = cov00 + COV_BLUR;
    cov2d[1] = cov10;
    cov2d[2] = cov11 + COV_BLUR;
}

/* helpers.wgsl:160-164 */
static inline void cov_to_conic(const float *c, float *conic) {
    float det = c[0] * c[2] - c[1] * c[1];
    float inv_det = 1.0f / det;
    conic[0] = c[2] * inv_det;
    conic[1] = (-c[1]) * inv_det;
    conic[2] = c[0] * inv_det;
}

/* helpers.wgsl:192-201 (opacity argument ignored by the reference). */
static inline uint32_t radius_from_conic(const float *conic) {
    float det = 1.0f / (conic[0] * conic[2] - conic[1] * conic[1]);
    float cx = conic[2] * det, cz = conic[0] * det;
    float b = 0.5f * (cx + cz);
    float sq = sqrtf(fmaxf(0.1f, b * b - det));
    float v1 = b + sq, v2 = b - sq;
    float radius = 3.0f * sqrtf(fmaxf(0.0f, fmaxf(v1, v2)));
    return f2u_sat(ceilf(radius));
}

/* helpers.wgsl:55-71.  bbox = (min.x, min.y, max.x, max.y), max exclusive. */
static inline void get_tile_bbox(const float *xy, uint32_t radius, const uint32_t *bounds,
                                 uint32_t *bbox) {
    float tr = (float)radius / (float)TILE_WIDTH;
    for (int i = 0; i < 2; i++) {
        float tc = xy[i] / (float)TILE_WIDTH;
        bbox[i] = (uint32_t)iclamp(f2i_sat(tc - tr), 0, (int32_t)bounds[i]);
        bbox[2 + i] = (uint32_t)iclamp(f2i_sat((tc + tr) + 1.0f), 0, (int32_t)bounds[i]);
    }
}

/* v * Q with Q = mat2x2f(q0,q1,q1,q2) (row-vector times symmetric matrix). */
static inline void vq(const float *v, const float *q, float *o) {
    o[0] = v[0] * q[0] + v[1] * q[1];
    o[1] = v[0] * q[1] + v[1] * q[2];
}
static inline float dot2(const float *a, const float *b) { return a[0] * b[0] + a[1] * b[1]; }

/* helpers.wgsl:220-236 */
static inline int check_edge(const float *p1, const float *p2, const float *center,
                             const float *q) {
    float edge[2] = {p2[0] - p1[0], p2[1] - p1[1]};
    float f[2] = {p1[0] - center[0], p1[1] - center[1]};
    float eq[2], fq[2];
    vq(edge, q, eq);
    vq(f, q, fq);
    float a = dot2(eq, edge);
    float b = 2.0f * dot2(fq, edge);
    float c = dot2(fq, f) - 1.0f;
    float disc = b * b - 4.0f * a * c;
    if (disc < 0.0f) return 0;
    float sd = sqrtf(disc);
    float t1 = (-b - sd) / (2.0f * a);
    float t2 = (-b + sd) / (2.0f * a);
    return (t1 >= 0.0f && t1 <= 1.0f) || (t2 >= 0.0f && t2 <= 1.0f);
}

/* helpers.wgsl:238-262 */
static inline int ellipse_intersects_aabb(const float *box_pos, const float *ext,
                                          const float *center, const float *q) {
    float d[2] = {center[0] - box_pos[0], center[1] - box_pos[1]};
    if (fabsf(d[0]) <= ext[0] && fabsf(d[1]) <= ext[1]) return 1;
    float sg[2] = {signf(d[0]), signf(d[1])};
    float nc[2] = {box_pos[0] + sg[0] * ext[0], box_pos[1] + sg[1] * ext[1]};
    float cp[2] = {nc[0] - center[0], nc[1] - center[1]};
    float cq[2];
    vq(cp, q, cq);
    if (dot2(cq, cp) <= 1.0f) return 1;
    float e1[2] = {nc[0] - sg[0] * 2.0f * ext[0], nc[1] - 0.0f};
    float e2[2] = {nc[0] - 0.0f, nc[1] - sg[1] * 2.0f * ext[1]};
    return check_edge(nc, e1, center, q) || check_edge(nc, e2, center, q);
}

/* helpers.wgsl:264-279 */
static inline int can_be_visible(uint32_t tx, uint32_t ty, const float *xy, const float *conic,
                                 float opac) {
    float sigma = det_logf(opac * 255.0f);
    if (sigma <= 0.0f) return 0;
    float den = 2.0f * sigma;
    float q[3] = {conic[0] / den, conic[1] / den, conic[2] / den};
    float ext[2] = {(float)TILE_WIDTH / 2.0f, (float)TILE_WIDTH / 2.0f};
    float tc[2] = {(float)(tx * TILE_WIDTH) + ext[0], (float)(ty * TILE_WIDTH) + ext[1]};
    return ellipse_intersects_aabb(tc, ext, xy, q);
}

static inline float sigmoidf_det(float x) { return 1.0f / (1.0f + det_expf(-x)); }

/* ---------------------------------------------------------------- SH basis
 * project_visible.wgsl:51-147 / gather_grads.wgsl:17-112: Sloan's polynomial basis.
 * Fills Y[0..(deg+1)^2).
 */
static void sh_basis(uint32_t degree, const float *d, float *Y) {
    Y[0] = 0.2820947917738781f;
    if (degree == 0) return;
    float x = d[0], y = d[1], z = d[2];
    float fTmp0A = 0.48860251190292f;
    Y[1] = -fTmp0A * y;
    Y[2] = fTmp0A * z;
    Y[3] = -fTmp0A * x;
    if (degree == 1) return;
    float z2 = z * z;
    float fTmp0B = -1.092548430592079f * z;
    float fTmp1A = 0.5462742152960395f;
    float fC1 = x * x - y * y;
    float fS1 = 2.0f * x * y;
    float pSH6 = 0.9461746957575601f * z2 - 0.3153915652525201f;
    Y[4] = fTmp1A * fS1;
    Y[5] = fTmp0B * y;
    Y[6] = pSH6;
    Y[7] = fTmp0B * x;
    Y[8] = fTmp1A * fC1;
    if (degree == 2) return;
    float fTmp0C = -2.285228997322329f * z2 + 0.4570457994644658f;
    float fTmp1B = 1.445305721320277f * z;
    float fTmp2A = -0.5900435899266435f;
    float fC2 = x * fC1 - y * fS1;
    float fS2 = x * fS1 + y * fC1;
    float pSH12 = z * (1.865881662950577f * z2 - 1.119528997770346f);
    Y[9] = fTmp2A * fS2;
    Y[10] = fTmp1B * fS1;
    Y[11] = fTmp0C * y;
    Y[12] = pSH12;
    Y[13] = fTmp0C * x;
    Y[14] = fTmp1B * fC1;
    Y[15] = fTmp2A * fC2;
    if (degree == 3) return;
    float fTmp0D = z * (-4.683325804901025f * z2 + 2.007139630671868f);
    float fTmp1C = 3.31161143515146f * z2 - 0.47308734787878f;
    float fTmp2B = -1.770130769779931f * z;
    float fTmp3A = 0.6258357354491763f;
    float fC3 = x * fC2 - y * fS2;
    float fS3 = x * fS2 + y * fC2;
    Y[16] = fTmp3A * fS3;
    Y[17] = fTmp2B * fS2;
    Y[18] = fTmp1C * fS1;
    Y[19] = fTmp0D * y;
    Y[20] = 1.984313483298443f * z * pSH12 - 1.006230589874905f * pSH6;
    Y[21] = fTmp0D * x;
    Y[22] = fTmp1C * fC1;
    Y[23] = fTmp2B * fC2;
    Y[24] = fTmp3A * fC3;
}

/* gather_grads.wgsl:186-222: the basis values Y_k(dir) a splat's v_coeffs rows are built from (v_sh[k] = Y_k * v_rgb),
 * dir = normalize(mean - viewmat[3].xyz) as the shader takes it.  Y is [n][(degree + 1)^2]. */
int oracle_sh_basis_for_means(const OracleUniforms *u, const float *means, uint32_t n, float *Y_out) {
    const uint32_t ncoef = (u->sh_degree + 1) * (u->sh_degree + 1);
#pragma omp parallel for schedule(static)
    for (int64_t g = 0; g < (int64_t)n; g++) {
        const float *mean = means + (size_t)g * 3;
        float dir[3] = {mean[0] - u->viewmat[12], mean[1] - u->viewmat[13], mean[2] - u->viewmat[14]};
        float len = sqrtf(dir[0] * dir[0] + dir[1] * dir[1] + dir[2] * dir[2]);
        dir[0] = dir[0] / len; dir[1] = dir[1] / len; dir[2] = dir[2] / len;
        float Y[25];
        sh_basis(u->sh_degree, dir, Y);
        for (uint32_t k = 0; k < ncoef; k++) Y_out[(size_t)g * ncoef + k] = Y[k];
    }
    return 0;
}

/* project_visible.wgsl:51-147: colour = sum over bands, each band summed left to right and
 * then added to the running colour, exactly as the WGSL expression tree. */
static void sh_to_color(uint32_t degree, const float *dir, const float *sh /*[C][3]*/, float *rgb) {
    float Y[25];
    sh_basis(degree, dir, Y);
    static const int band_start[6] = {0, 1, 4, 9, 16, 25};
    for (int ch = 0; ch < 3; ch++) {
        float col = Y[0] * sh[ch];
        if (degree >= 1) {
            /* fTmp0A * (-y*c0 + z*c1 - x*c2) */
            float x = dir[0], y = dir[1], z = dir[2];
            float inner = ((-y) * sh[1 * 3 + ch] + z * sh[2 * 3 + ch]) - x * sh[3 * 3 + ch];
            col = col + 0.48860251190292f * inner;
        }
        for (uint32_t b = 2; b <= degree; b++) {
            int s = band_start[b], e = band_start[b + 1];
            float acc = Y[s] * sh[s * 3 + ch];
            for (int k = s + 1; k < e; k++) acc = acc + Y[k] * sh[k * 3 + ch];
            col = col + acc;
        }
        rgb[ch] = col;
    }
}

/* ---------------------------------------------------------------- sort / scan */

void oracle_radix_argsort(const uint32_t *keys_in, const uint32_t *vals_in, uint32_t n,
                          uint32_t sorting_bits, uint32_t *keys_out, uint32_t *vals_out) {
    /* brush-sort/src/lib.rs:58 — ceil(bits/4) passes of 4 bits, LSD, each pass stable
     * (sort_scatter.wgsl:55-121).  Restated as the same passes with a counting sort. */
    uint32_t passes = (sorting_bits + 3u) / 4u;
    uint32_t *ka = (uint32_t *)malloc(sizeof(uint32_t) * (n ? n : 1));
    uint32_t *va = (uint32_t *)malloc(sizeof(uint32_t) * (n ? n : 1));
    uint32_t *kb = (uint32_t *)malloc(sizeof(uint32_t) * (n ? n : 1));
    uint32_t *vb = (uint32_t *)malloc(sizeof(uint32_t) * (n ? n : 1));
    memcpy(ka, keys_in, sizeof(uint32_t) * n);
    memcpy(va, vals_in, sizeof(uint32_t) * n);
    for (uint32_t p = 0; p < passes; p++) {
        uint32_t shift = p * 4u;
        uint32_t count[16] = {0};
        for (uint32_t i = 0; i < n; i++) count[(ka[i] >> shift) & 0xFu]++;
        uint32_t off[16], run = 0;
        for (int b = 0; b < 16; b++) {
            off[b] = run;
            run += count[b];
        }
        for (uint32_t i = 0; i < n; i++) {
            uint32_t b = (ka[i] >> shift) & 0xFu;
            kb[off[b]] = ka[i];
            vb[off[b]] = va[i];
            off[b]++;
        }
        uint32_t *t = ka; ka = kb; kb = t;
        t = va; va = vb; vb = t;
    }
    memcpy(keys_out, ka, sizeof(uint32_t) * n);
    memcpy(vals_out, va, sizeof(uint32_t) * n);
    free(ka); free(va); free(kb); free(vb);
}

void oracle_inclusive_scan(const uint32_t *in, uint32_t n, uint32_t *out) {
    /* brush-prefix-sum/src/shaders/prefix_sum_helpers.wgsl:8-22 — inclusive, wrapping u32. */
    uint32_t run = 0;
    for (uint32_t i = 0; i < n; i++) {
        run += in[i];
        out[i] = run;
    }
}

/* ---------------------------------------------------------------- forward stages */

/* project_forward.wgsl:15-68.  Returns 1 and the depth if splat g survives the cull. */
static int project_forward_one(const OracleUniforms *u, const float *mean, const float *log_scale,
                               const float *quat, float *depth) {
    float p_view[3];
    to_view(u->viewmat, mean, p_view);
    if (p_view[2] <= 0.01f) return 0;
    float scale[3] = {det_expf(log_scale[0]), det_expf(log_scale[1]), det_expf(log_scale[2])};
    float cov2d[3], conic[3], xy[2];
    calc_cov2d(u, p_view, scale, quat, cov2d);
    float det = cov2d[0] * cov2d[2] - cov2d[1] * cov2d[1];
    if (det == 0.0f) return 0;
    cov_to_conic(cov2d, conic);
    project_pix(u->focal, p_view, u->pixel_center, xy);
    uint32_t radius = radius_from_conic(conic);
    uint32_t bb[4];
    get_tile_bbox(xy, radius, u->tile_bounds, bb);
    if ((bb[2] - bb[0]) == 0u || (bb[3] - bb[1]) == 0u) return 0;
    *depth = p_view[2];
    return 1;
}

/* project_visible.wgsl:163-258 */
static void project_visible_one(const OracleUniforms *u, const float *mean, const float *log_scale,
                                const float *quat, const float *sh, float raw_opac,
                                float *projected /*[9]*/, uint32_t *tiles_hit) {
    float scale[3] = {det_expf(log_scale[0]), det_expf(log_scale[1]), det_expf(log_scale[2])};
    float opac = sigmoidf_det(raw_opac);
    float p_view[3], cov2d[3], conic[3], xy[2];
    to_view(u->viewmat, mean, p_view);
    calc_cov2d(u, p_view, scale, quat, cov2d);
    cov_to_conic(cov2d, conic);
    project_pix(u->focal, p_view, u->pixel_center, xy);

    /* Quirk: camera_pos = viewmat[3].xyz, not the camera position. */
    float dir[3] = {mean[0] - u->viewmat[12], mean[1] - u->viewmat[13], mean[2] - u->viewmat[14]};
    float len = sqrtf(dir[0] * dir[0] + dir[1] * dir[1] + dir[2] * dir[2]);
    dir[0] = dir[0] / len; dir[1] = dir[1] / len; dir[2] = dir[2] / len;
    float rgb[3];
    sh_to_color(u->sh_degree, dir, sh, rgb);

    uint32_t radius = radius_from_conic(conic);
    uint32_t bb[4];
    get_tile_bbox(xy, radius, u->tile_bounds, bb);
    uint32_t area = 0;
    for (uint32_t ty = bb[1]; ty < bb[3]; ty++)
        for (uint32_t tx = bb[0]; tx < bb[2]; tx++)
            if (can_be_visible(tx, ty, xy, conic, opac)) area++;

    projected[0] = xy[0]; projected[1] = xy[1];
    projected[2] = conic[0]; projected[3] = conic[1]; projected[4] = conic[2];
    projected[5] = rgb[0] + 0.5f; projected[6] = rgb[1] + 0.5f; projected[7] = rgb[2] + 0.5f;
    projected[8] = opac;
    *tiles_hit = area;
}

/* map_gaussian_to_intersects.wgsl:10-48 */
static void map_one(const OracleUniforms *u, const float *projected, uint32_t compact_gid,
                    uint32_t isect_start, uint32_t cap, uint32_t *tile_ids, uint32_t *gids) {
    const float *xy = projected;
    const float *conic = projected + 2;
    float opac = projected[8];
    uint32_t radius = radius_from_conic(conic);
    uint32_t bb[4];
    get_tile_bbox(xy, radius, u->tile_bounds, bb);
    uint32_t isect = isect_start;
    for (uint32_t ty = bb[1]; ty < bb[3]; ty++)
        for (uint32_t tx = bb[0]; tx < bb[2]; tx++)
            if (can_be_visible(tx, ty, xy, conic, opac) && isect < cap) {
                tile_ids[isect] = tx + ty * u->tile_bounds[0];
                gids[isect] = compact_gid;
                isect++;
            }
}

/* rasterize.wgsl:20-115 — one tile. */
static void rasterize_tile(const OracleUniforms *u, uint32_t tile_x, uint32_t tile_y,
                           const uint32_t *gid_from_isect, const uint32_t *tile_bins,
                           const float *projected, int raster_u32, void *out_img,
                           uint32_t *final_index, uint8_t *flip_risk) {
    uint32_t w = u->img_size[0], h = u->img_size[1];
    uint32_t tile_id = tile_x + tile_y * u->tile_bounds[0];
    uint32_t r0 = tile_bins[tile_id * 2], r1 = tile_bins[tile_id * 2 + 1];
    for (uint32_t ly = 0; ly < TILE_WIDTH; ly++) {
        for (uint32_t lx = 0; lx < TILE_WIDTH; lx++) {
            uint32_t px = tile_x * TILE_WIDTH + lx, py = tile_y * TILE_WIDTH + ly;
            if (px >= w || py >= h) continue;
            float pcx = (float)px + 0.5f, pcy = (float)py + 0.5f;
            float T = 1.0f, rgb[3] = {0, 0, 0};
            uint32_t final_idx = 0;
            uint8_t risk = 0;
            for (uint32_t i = r0; i < r1; i++) {
                const float *p = projected + (size_t)gid_from_isect[i] * 9;
                float dx = p[0] - pcx, dy = p[1] - pcy;
                float sigma = 0.5f * (p[2] * dx * dx + p[4] * dy * dy) + p[3] * dx * dy;
                float vis = expf(-sigma);
                float alpha = fminf(0.999f, p[8] * vis);
                /* guard band for threshold flips under a different exp() implementation */
                if (fabsf(alpha - (1.0f / 255.0f)) <= 1e-5f * (1.0f / 255.0f) ||
                    fabsf(sigma) <= 1e-6f)
                    risk = 1;
                if (sigma >= 0.0f && alpha >= 1.0f / 255.0f) {
                    float next_T = T * (1.0f - alpha);
                    if (fabsf(next_T - 1e-4f) <= 1e-5f * 1e-4f) risk = 1;
                    if (next_T <= 1e-4f) break;
                    float fac = alpha * T;
                    rgb[0] += p[5] * fac; rgb[1] += p[6] * fac; rgb[2] += p[7] * fac;
                    T = next_T;
                    final_idx = i;
                }
            }
            size_t pix = (size_t)px + (size_t)py * w;
            float a = 1.0f - T;
            if (raster_u32) {
                /* rasterize.wgsl:106-109 */
                float c4[4] = {rgb[0], rgb[1], rgb[2], a};
                uint32_t packed = 0;
                for (int k = 0; k < 4; k++)
                    packed |= f2u_sat(clampf(c4[k] * 255.0f, 0.0f, 255.0f)) << (8 * k);
                ((uint32_t *)out_img)[pix] = packed;
            } else {
                float *o = (float *)out_img + pix * 4;
                o[0] = rgb[0]; o[1] = rgb[1]; o[2] = rgb[2]; o[3] = a;
                final_index[pix] = final_idx;
            }
            if (flip_risk) flip_risk[pix] = risk;
        }
    }
}

int oracle_render_forward(const OracleUniforms *u_in, const float *means, const float *log_scales,
                          const float *quats, const float *sh_coeffs, const float *raw_opac,
                          uint32_t n, int raster_u32, void *out_img, OracleAux *aux) {
    OracleUniforms uu = *u_in;
    uu.total_splats = n;
    const OracleUniforms *u = &uu;
    uint32_t ncoef = (u->sh_degree + 1) * (u->sh_degree + 1);
    uint32_t num_tiles = u->tile_bounds[0] * u->tile_bounds[1];

    /* --- ProjectSplats (render.rs:123-142).  The reference compacts through an atomic
     * counter (project_forward.wgsl:65) in nondeterministic order; the restatement compacts
     * in ascending global id, which with the stable depth sort fixes the tie order. */
    uint8_t *vis = (uint8_t *)calloc(n ? n : 1, 1);
    float *depth_all = (float *)malloc(sizeof(float) * (n ? n : 1));
#pragma omp parallel for schedule(static)
    for (int64_t g = 0; g < (int64_t)n; g++) {
        float d = 0.0f;
        vis[g] = (uint8_t)project_forward_one(u, means + g * 3, log_scales + g * 3, quats + g * 4, &d);
        depth_all[g] = d;
    }
    uint32_t V = 0;
    uint32_t *pre_gid = (uint32_t *)malloc(sizeof(uint32_t) * (n ? n : 1));
    uint32_t *pre_key = (uint32_t *)malloc(sizeof(uint32_t) * (n ? n : 1));
    for (uint32_t g = 0; g < n; g++)
        if (vis[g]) {
            pre_gid[V] = g;
            memcpy(&pre_key[V], &depth_all[g], 4);
            V++;
        }
    aux->num_visible[0] = V;

    /* --- DepthSort (render.rs:151-156): keys = depth bits, 32 bits. */
    uint32_t *sorted_key = (uint32_t *)malloc(sizeof(uint32_t) * (n ? n : 1));
    memset(aux->global_from_compact_gid, 0, sizeof(uint32_t) * n); /* tail = 0 */
    oracle_radix_argsort(pre_key, pre_gid, V, 32, sorted_key, aux->global_from_compact_gid);

    /* --- ProjectVisible (render.rs:161-184) */
    uint32_t *tiles_hit = (uint32_t *)calloc(n ? n : 1, sizeof(uint32_t));
#pragma omp parallel for schedule(dynamic, 256)
    for (int64_t c = 0; c < (int64_t)V; c++) {
        uint32_t g = aux->global_from_compact_gid[c];
        project_visible_one(u, means + (size_t)g * 3, log_scales + (size_t)g * 3, quats + (size_t)g * 4,
                            sh_coeffs + (size_t)g * ncoef * 3, raw_opac[g],
                            aux->projected_splats + (size_t)c * 9, &tiles_hit[c]);
    }

    /* --- PrefixSum over all N (render.rs:186-192) */
    oracle_inclusive_scan(tiles_hit, n, aux->cum_tiles_hit);
    uint32_t total = n ? aux->cum_tiles_hit[n - 1] : 0;
    uint32_t cap = aux->max_intersects;
    uint32_t I = total < cap ? total : cap; /* build: clamp instead of the reference's UB */
    aux->num_intersections[0] = I;

    /* --- MapGaussiansToIntersect (render.rs:211-223) */
    uint32_t *tile_unsorted = (uint32_t *)malloc(sizeof(uint32_t) * (cap ? cap : 1));
    uint32_t *gid_unsorted = (uint32_t *)malloc(sizeof(uint32_t) * (cap ? cap : 1));
#pragma omp parallel for schedule(dynamic, 256)
    for (int64_t c = 0; c < (int64_t)V; c++) {
        uint32_t start = c > 0 ? aux->cum_tiles_hit[c - 1] : 0u;
        map_one(u, aux->projected_splats + (size_t)c * 9, (uint32_t)c, start, cap, tile_unsorted,
                gid_unsorted);
    }

    /* --- Tile sort (render.rs:227-237): bits = 32 - clz(num_tiles) */
    uint32_t bits = 0;
    while (bits < 32 && (num_tiles >> bits) != 0) bits++;
    uint32_t *tile_sorted = aux->tile_id_from_isect;
    int own_tile_sorted = 0;
    if (!tile_sorted) {
        tile_sorted = (uint32_t *)malloc(sizeof(uint32_t) * (cap ? cap : 1));
        own_tile_sorted = 1;
    }
    oracle_radix_argsort(tile_unsorted, gid_unsorted, I, bits, tile_sorted,
                         aux->compact_gid_from_isect);

    /* --- GetTileBinEdges (get_tile_bin_edges.wgsl:15-42) */
    memset(aux->tile_bins, 0, sizeof(uint32_t) * 2 * num_tiles);
    for (uint32_t i = 0; i < I; i++) {
        uint32_t cur = tile_sorted[i];
        if (i == I - 1) aux->tile_bins[cur * 2 + 1] = I;
        if (i == 0) {
            aux->tile_bins[cur * 2 + 0] = 0;
        } else {
            uint32_t prev = tile_sorted[i - 1];
            if (prev != cur) {
                aux->tile_bins[prev * 2 + 1] = i;
                aux->tile_bins[cur * 2 + 0] = i;
            }
        }
    }

    /* --- Rasterize (render.rs:267-307) */
    int64_t nt = (int64_t)num_tiles;
#pragma omp parallel for schedule(dynamic, 4)
    for (int64_t t = 0; t < nt; t++) {
        uint32_t tx = (uint32_t)(t % u->tile_bounds[0]), ty = (uint32_t)(t / u->tile_bounds[0]);
        rasterize_tile(u, tx, ty, aux->compact_gid_from_isect, aux->tile_bins,
                       aux->projected_splats, raster_u32, out_img, aux->final_index,
                       aux->flip_risk);
    }

    free(vis); free(depth_all); free(pre_gid); free(pre_key); free(sorted_key); free(tiles_hit);
    free(tile_unsorted); free(gid_unsorted);
    if (own_tile_sorted) free(tile_sorted);
    return total > cap ? 1 : 0; /* 1 = intersections truncated */
}

/* ---------------------------------------------------------------- backward stages */

/* rasterize_backwards.wgsl:140-304 — one tile; sums over the tile's pixels are kept in f64
 * (the reference's subgroupAdd / atomic order is unspecified; this is the order-free value). */
static void rasterize_backward_tile(const OracleUniforms *u, uint32_t tile_id,
                                    const uint32_t *gid_from_isect, const uint32_t *tile_bins,
                                    const float *projected, const uint32_t *final_index,
                                    const float *out_img, const float *v_out, double *v_xy,
                                    double *v_conic, double *v_colors, float *rows32) {
    uint32_t w = u->img_size[0], h = u->img_size[1];
    uint32_t tbx = u->tile_bounds[0];
    uint32_t tile_x = tile_id % tbx, tile_y = tile_id / tbx;
    uint32_t r0 = tile_bins[tile_id * 2], r1 = tile_bins[tile_id * 2 + 1];
    if (r1 <= r0) return;

    float T[TILE_SIZE], T_final[TILE_SIZE], buf[TILE_SIZE][3], vo[TILE_SIZE][4];
    float pcx[TILE_SIZE], pcy[TILE_SIZE];
    uint32_t fin[TILE_SIZE];
    uint8_t inside[TILE_SIZE];
    for (uint32_t l = 0; l < TILE_SIZE; l++) {
        uint32_t px = tile_x * TILE_WIDTH + l % TILE_WIDTH, py = tile_y * TILE_WIDTH + l / TILE_WIDTH;
        inside[l] = px < w && py < h;
        pcx[l] = (float)px + 0.5f; pcy[l] = (float)py + 0.5f;
        buf[l][0] = buf[l][1] = buf[l][2] = 0.0f;
        fin[l] = 0; T[l] = T_final[l] = 1.0f;
        vo[l][0] = vo[l][1] = vo[l][2] = vo[l][3] = 0.0f;
        if (inside[l]) {
            size_t pix = (size_t)px + (size_t)py * w;
            T_final[l] = 1.0f - out_img[pix * 4 + 3];
            T[l] = T_final[l];
            fin[l] = final_index[pix];
            for (int k = 0; k < 4; k++) vo[l][k] = v_out[pix * 4 + k];
        }
    }
    for (uint32_t i = r1; i-- > r0;) {
        uint32_t cg = gid_from_isect[i];
        const float *p = projected + (size_t)cg * 9;
        double s_xy[2] = {0, 0}, s_conic[3] = {0, 0, 0}, s_col[4] = {0, 0, 0, 0};
        int any = 0;
        for (uint32_t l = 0; l < TILE_SIZE; l++) {
            if (!(inside[l] && i <= fin[l])) continue;
            float dx = p[0] - pcx[l], dy = p[1] - pcy[l];
            float sigma = 0.5f * (p[2] * dx * dx + p[4] * dy * dy) + p[3] * dx * dy;
            float vis = expf(-sigma);
            float alpha = fminf(0.99f, p[8] * vis); /* 0.99 here, 0.999 forward (quirk 2) */
            if (!(sigma >= 0.0f && alpha >= 1.0f / 255.0f)) continue;
            any = 1;
            float ra = 1.0f / (1.0f - alpha);
            T[l] *= ra;
            float fac = alpha * T[l];
            float v_alpha = ((p[5] * T[l] - buf[l][0] * ra) * vo[l][0] +
                             (p[6] * T[l] - buf[l][1] * ra) * vo[l][1]) +
                            (p[7] * T[l] - buf[l][2] * ra) * vo[l][2];
            v_alpha += T_final[l] * ra * vo[l][3];
            buf[l][0] += p[5] * fac; buf[l][1] += p[6] * fac; buf[l][2] += p[7] * fac;
            float v_sigma = -p[8] * vis * v_alpha;
            s_xy[0] += v_sigma * (p[2] * dx + p[3] * dy);
            s_xy[1] += v_sigma * (p[3] * dx + p[4] * dy);
            s_conic[0] += 0.5f * v_sigma * dx * dx;
            s_conic[1] += v_sigma * dx * dy;
            s_conic[2] += 0.5f * v_sigma * dy * dy;
            s_col[0] += fac * vo[l][0]; s_col[1] += fac * vo[l][1]; s_col[2] += fac * vo[l][2];
            s_col[3] += vis * v_alpha;
            if (rows32) { /* (double)(float)(a + b) with a, b floats IS the f32 sum a + b */
                s_xy[0] = (float)s_xy[0]; s_xy[1] = (float)s_xy[1];
                s_conic[0] = (float)s_conic[0]; s_conic[1] = (float)s_conic[1]; s_conic[2] = (float)s_conic[2];
                s_col[0] = (float)s_col[0]; s_col[1] = (float)s_col[1]; s_col[2] = (float)s_col[2];
                s_col[3] = (float)s_col[3];
            }
        }
        if (rows32) {
            /* f32-sum mode: every addition above is re-done in f32 below; the row is summed per splat in
             * ascending intersection order by the caller. */
            float *r = rows32 + (size_t)i * 9;
            r[0] = (float)s_xy[0]; r[1] = (float)s_xy[1];
            r[2] = (float)s_conic[0]; r[3] = (float)s_conic[1]; r[4] = (float)s_conic[2];
            r[5] = (float)s_col[0]; r[6] = (float)s_col[1]; r[7] = (float)s_col[2]; r[8] = (float)s_col[3];
            continue;
        }
        if (!any) continue;
        for (int k = 0; k < 2; k++) {
#pragma omp atomic
            v_xy[(size_t)cg * 2 + k] += s_xy[k];
        }
        for (int k = 0; k < 3; k++) {
#pragma omp atomic
            v_conic[(size_t)cg * 3 + k] += s_conic[k];
        }
        for (int k = 0; k < 4; k++) {
#pragma omp atomic
            v_colors[(size_t)cg * 4 + k] += s_col[k];
        }
    }
}

/* project_backwards.wgsl:19-23 */
static inline void project_pix_vjp(const float *f, const float *p, const float *v_xy, float *o) {
    float rw = 1.0f / (p[2] + 1e-6f);
    float vp0 = f[0] * v_xy[0], vp1 = f[1] * v_xy[1];
    o[0] = vp0 * rw;
    o[1] = vp1 * rw;
    o[2] = -(vp0 * p[0] + vp1 * p[1]) * rw * rw;
}

/* project_backwards.wgsl:25-57.  G(a,b) = WGSL v_R[a][b] = column a, row b. */
static inline void quat_to_rotmat_vjp(const float *q, const mat3 *vR, float *o) {
#define G(a, b) (vR->m[b][a])
    float w = q[0], x = q[1], y = q[2], z = q[3];
    o[0] = 2.0f * ((x * (G(1, 2) - G(2, 1)) + y * (G(2, 0) - G(0, 2))) + z * (G(0, 1) - G(1, 0)));
    o[1] = 2.0f * (((-2.0f * x * (G(1, 1) + G(2, 2)) + y * (G(0, 1) + G(1, 0))) +
                    z * (G(0, 2) + G(2, 0))) + w * (G(1, 2) - G(2, 1)));
    o[2] = 2.0f * (((x * (G(0, 1) + G(1, 0)) - 2.0f * y * (G(0, 0) + G(2, 2))) +
                    z * (G(1, 2) + G(2, 1))) + w * (G(2, 0) - G(0, 2)));
    o[3] = 2.0f * (((x * (G(0, 2) + G(2, 0)) + y * (G(1, 2) + G(2, 1))) -
                    2.0f * z * (G(0, 0) + G(1, 1))) + w * (G(0, 1) - G(1, 0)));
#undef G
}

/* project_backwards.wgsl:59-72 */
static inline void cov2d_to_conic_vjp(const float *conic, const float *v_conic, float *o) {
    float X[2][2] = {{conic[0], conic[1]}, {conic[1], conic[2]}};
    float Gm[2][2] = {{v_conic[0], v_conic[1] / 2.0f}, {v_conic[1] / 2.0f, v_conic[2]}};
    float XG[2][2], S[2][2];
    for (int i = 0; i < 2; i++)
        for (int j = 0; j < 2; j++) XG[i][j] = X[i][0] * Gm[0][j] + X[i][1] * Gm[1][j];
    for (int i = 0; i < 2; i++)
        for (int j = 0; j < 2; j++) S[i][j] = XG[i][0] * X[0][j] + XG[i][1] * X[1][j];
    o[0] = -S[0][0];
    o[1] = -(S[0][1] + S[1][0]);
    o[2] = -S[1][1];
}

/* project_backwards.wgsl:75-227 for one visible splat. */
static void project_backward_one(const OracleUniforms *u, const float *mean, const float *log_scale,
                                 const float *quat, const float *v_xy, const float *v_conic,
                                 float *v_mean, float *v_scale_out, float *v_quat) {
    const float *focal = u->focal;
    float scale[3] = {det_expf(log_scale[0]), det_expf(log_scale[1]), det_expf(log_scale[2])};
    mat3 W = view_rot(u->viewmat);
    float p_view[3];
    to_view(u->viewmat, mean, p_view);
    float vp[3];
    project_pix_vjp(focal, p_view, v_xy, vp);
    float vm[3];
    for (int i = 0; i < 3; i++) /* transpose(W) * vp */
        vm[i] = W.m[0][i] * vp[0] + W.m[1][i] * vp[1] + W.m[2][i] * vp[2];

    float cov2d[3], conic[3], v_cov2d[3];
    calc_cov2d(u, p_view, scale, quat, cov2d);
    cov_to_conic(cov2d, conic);
    cov2d_to_conic_vjp(conic, v_conic, v_cov2d);

    float rz = 1.0f / p_view[2];
    float rz2 = rz * rz;
    /* Quirk 3: J from the UNCLAMPED p_view (project_backwards.wgsl:134-138). */
    mat3 J = {{{focal[0] * rz, 0.0f, (-focal[0]) * p_view[0] * rz2},
               {0.0f, focal[1] * rz, (-focal[1]) * p_view[1] * rz2},
               {0.0f, 0.0f, 0.0f}}};
    mat3 R = quat_to_rotmat(quat);
    mat3 S = {{{scale[0], 0, 0}, {0, scale[1], 0}, {0, 0, scale[2]}}};
    mat3 M = mat3_mul(R, S);
    mat3 V = mat3_mul(M, mat3_transpose(M));
    mat3 v_cov = {{{v_cov2d[0], 0.5f * v_cov2d[1], 0.0f},
                   {0.5f * v_cov2d[1], v_cov2d[2], 0.0f},
                   {0.0f, 0.0f, 0.0f}}};
    mat3 T = mat3_mul(J, W);
    mat3 Tt = mat3_transpose(T);
    mat3 Vt = mat3_transpose(V);
    mat3 v_V = mat3_mul(mat3_mul(Tt, v_cov), T);
    mat3 v_T = mat3_add(mat3_mul(mat3_mul(v_cov, T), Vt),
                        mat3_mul(mat3_mul(mat3_transpose(v_cov), T), V));

    float c0 = v_V.m[0][0];
    float c1 = v_V.m[1][0] + v_V.m[0][1];
    float c2 = v_V.m[2][0] + v_V.m[0][2];
    float c3 = v_V.m[1][1];
    float c4 = v_V.m[2][1] + v_V.m[1][2];
    float c5 = v_V.m[2][2];

    mat3 v_J = mat3_mul(v_T, mat3_transpose(W));
    float rz3 = rz2 * rz;
    /* WGSL v_J[2][0] = row 0 col 2, v_J[2][1] = row 1 col 2 */
    float vJ02 = v_J.m[0][2], vJ12 = v_J.m[1][2], vJ00 = v_J.m[0][0], vJ11 = v_J.m[1][1];
    float v_t[3];
    v_t[0] = (-focal[0]) * rz2 * vJ02;
    v_t[1] = (-focal[1]) * rz2 * vJ12;
    v_t[2] = (((-focal[0]) * rz2 * vJ00 + 2.0f * focal[0] * p_view[0] * rz3 * vJ02) -
              focal[1] * rz2 * vJ11) + 2.0f * focal[1] * p_view[1] * rz3 * vJ12;
    for (int i = 0; i < 3; i++) /* dot(v_t, W[i]) with W[i] = column i */
        vm[i] = vm[i] + ((v_t[0] * W.m[0][i] + v_t[1] * W.m[1][i]) + v_t[2] * W.m[2][i]);

    mat3 vVs = {{{c0, 0.5f * c1, 0.5f * c2}, {0.5f * c1, c3, 0.5f * c4}, {0.5f * c2, 0.5f * c4, c5}}};
    mat3 two_vVs;
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) two_vVs.m[i][j] = 2.0f * vVs.m[i][j];
    mat3 v_M = mat3_mul(two_vVs, M);
    for (int j = 0; j < 3; j++) { /* dot(R[j], v_M[j]) over columns */
        float vs = (R.m[0][j] * v_M.m[0][j] + R.m[1][j] * v_M.m[1][j]) + R.m[2][j] * v_M.m[2][j];
        v_scale_out[j] = vs * scale[j]; /* log-space: v_scale * scale (:219) */
    }
    mat3 v_R = mat3_mul(v_M, S);
    quat_to_rotmat_vjp(quat, &v_R, v_quat);
    v_mean[0] = vm[0]; v_mean[1] = vm[1]; v_mean[2] = vm[2];
}

/* f32_sums = 0: per-pixel f32 arithmetic, tile and cross-tile sums in f64 (the order-free value the
 * reference's unordered subgroupAdd / atomics approximate).  f32_sums = 1: every sum in f32 as well,
 * pixels in tile order, then a splat's tiles in ascending intersection order: one admissible
 * execution of the reference's own f32 arithmetic, used as the yardstick of the f64 arbiter
 * (brush_oracle_f64.c). */
int oracle_render_backward_ex(const OracleUniforms *u_in, const OracleAux *aux, const float *means,
                              const float *log_scales, const float *quats, const float *raw_opac,
                              uint32_t n, const float *out_img, const float *v_out, float *v_means,
                              float *v_xy, float *v_scales, float *v_quats, float *v_sh,
                              float *v_opac, float *o_v_xy_local, float *o_v_conics,
                              float *o_v_colors, int f32_sums);

int oracle_render_backward(const OracleUniforms *u_in, const OracleAux *aux, const float *means,
                           const float *log_scales, const float *quats, const float *raw_opac,
                           uint32_t n, const float *out_img, const float *v_out, float *v_means,
                           float *v_xy, float *v_scales, float *v_quats, float *v_sh,
                           float *v_opac, float *o_v_xy_local, float *o_v_conics,
                           float *o_v_colors) {
    return oracle_render_backward_ex(u_in, aux, means, log_scales, quats, raw_opac, n, out_img, v_out, v_means,
                                     v_xy, v_scales, v_quats, v_sh, v_opac, o_v_xy_local, o_v_conics,
                                     o_v_colors, 0);
}

int oracle_render_backward_ex(const OracleUniforms *u_in, const OracleAux *aux, const float *means,
                              const float *log_scales, const float *quats, const float *raw_opac,
                              uint32_t n, const float *out_img, const float *v_out, float *v_means,
                              float *v_xy, float *v_scales, float *v_quats, float *v_sh,
                              float *v_opac, float *o_v_xy_local, float *o_v_conics,
                              float *o_v_colors, int f32_sums) {
    OracleUniforms uu = *u_in;
    uu.total_splats = n;
    const OracleUniforms *u = &uu;
    uint32_t V = aux->num_visible[0];
    uint32_t ncoef = (u->sh_degree + 1) * (u->sh_degree + 1);
    uint32_t num_tiles = u->tile_bounds[0] * u->tile_bounds[1];

    size_t vn = V ? V : 1;
    double *d_xy = (double *)calloc(vn * 2, sizeof(double));
    double *d_conic = (double *)calloc(vn * 3, sizeof(double));
    double *d_col = (double *)calloc(vn * 4, sizeof(double));

    /* RasterizeBackwards (render.rs:505-532) */
    uint32_t I = aux->num_intersections[0];
    float *rows32 = f32_sums ? (float *)calloc((size_t)(I ? I : 1) * 9, sizeof(float)) : NULL;
#pragma omp parallel for schedule(dynamic, 4)
    for (int64_t t = 0; t < (int64_t)num_tiles; t++)
        rasterize_backward_tile(u, (uint32_t)t, aux->compact_gid_from_isect, aux->tile_bins,
                                aux->projected_splats, aux->final_index, out_img, v_out, d_xy,
                                d_conic, d_col, rows32);
    if (rows32) {
        for (size_t i = 0; i < I; i++) { /* f32 adds, ascending intersection id */
            size_t c = aux->compact_gid_from_isect[i];
            const float *r = rows32 + i * 9;
            d_xy[c * 2] = (float)d_xy[c * 2] + r[0]; d_xy[c * 2 + 1] = (float)d_xy[c * 2 + 1] + r[1];
            for (int k = 0; k < 3; k++) d_conic[c * 3 + k] = (float)d_conic[c * 3 + k] + r[2 + k];
            for (int k = 0; k < 4; k++) d_col[c * 4 + k] = (float)d_col[c * 4 + k] + r[5 + k];
        }
        for (size_t c = 0; c < (size_t)V; c++) { /* the stored value is the f32 sum */
            d_xy[c * 2] = (float)d_xy[c * 2]; d_xy[c * 2 + 1] = (float)d_xy[c * 2 + 1];
            for (int k = 0; k < 3; k++) d_conic[c * 3 + k] = (float)d_conic[c * 3 + k];
            for (int k = 0; k < 4; k++) d_col[c * 4 + k] = (float)d_col[c * 4 + k];
        }
        free(rows32);
    }

    /* dense outputs are zero-initialised (render.rs:539-547,573-575) */
    memset(v_means, 0, sizeof(float) * 3 * n);
    memset(v_xy, 0, sizeof(float) * 2 * n);
    memset(v_scales, 0, sizeof(float) * 3 * n);
    memset(v_quats, 0, sizeof(float) * 4 * n);
    memset(v_sh, 0, sizeof(float) * 3 * (size_t)ncoef * n);
    memset(v_opac, 0, sizeof(float) * n);

#pragma omp parallel for schedule(static)
    for (int64_t c = 0; c < (int64_t)V; c++) {
        uint32_t g = aux->global_from_compact_gid[c];
        float vxy[2] = {(float)d_xy[c * 2], (float)d_xy[c * 2 + 1]};
        float vconic[3] = {(float)d_conic[c * 3], (float)d_conic[c * 3 + 1], (float)d_conic[c * 3 + 2]};
        float vcol[4] = {(float)d_col[c * 4], (float)d_col[c * 4 + 1], (float)d_col[c * 4 + 2],
                         (float)d_col[c * 4 + 3]};
        if (o_v_xy_local) { o_v_xy_local[c * 2] = vxy[0]; o_v_xy_local[c * 2 + 1] = vxy[1]; }
        if (o_v_conics) for (int k = 0; k < 3; k++) o_v_conics[c * 3 + k] = vconic[k];
        if (o_v_colors) for (int k = 0; k < 4; k++) o_v_colors[c * 4 + k] = vcol[k];

        /* GatherGrads (gather_grads.wgsl:165-232) */
        const float *mean = means + (size_t)g * 3;
        float dir[3] = {mean[0] - u->viewmat[12], mean[1] - u->viewmat[13], mean[2] - u->viewmat[14]};
        float len = sqrtf(dir[0] * dir[0] + dir[1] * dir[1] + dir[2] * dir[2]);
        dir[0] = dir[0] / len; dir[1] = dir[1] / len; dir[2] = dir[2] / len;
        float Y[25];
        sh_basis(u->sh_degree, dir, Y);
        float *vs = v_sh + (size_t)g * ncoef * 3;
        for (uint32_t k = 0; k < ncoef; k++)
            for (int ch = 0; ch < 3; ch++) vs[k * 3 + ch] = Y[k] * vcol[ch];
        float s = sigmoidf_det(raw_opac[g]);
        v_opac[g] = vcol[3] * (s * (1.0f - s));
        v_xy[(size_t)g * 2] = vxy[0];
        v_xy[(size_t)g * 2 + 1] = vxy[1];

        /* ProjectBackwards (project_backwards.wgsl:75-227) */
        project_backward_one(u, mean, log_scales + (size_t)g * 3, quats + (size_t)g * 4, vxy, vconic,
                             v_means + (size_t)g * 3, v_scales + (size_t)g * 3, v_quats + (size_t)g * 4);
    }
    free(d_xy); free(d_conic); free(d_col);
    return 0;
}
