/*
 * brush_oracle_f64.c — double-precision ARBITER of the backward pass (test infrastructure only).
 *
 * Same algorithm, same inputs and the same forward state as oracle_render_backward() in
 * brush_oracle.c (the f32 restatement of rasterize_backwards.wgsl / gather_grads.wgsl /
 * project_backwards.wgsl), but every VALUE is computed and summed in f64.  The DECISIONS of the
 * walk (`isect <= final_index`, `sigma >= 0`, `alpha >= 1/255`, the 0.99 clamp) are taken exactly
 * as the f32 restatement takes them (f32 arithmetic, same expression trees), so both runs sum the
 * same set of contributions.  It answers "which of two f32 results is closer to the exact value
 * of the reference's formulas": tests bound |gpu - f64| by a small multiple of |oracle_f32 - f64|.
 * Not pinned by reference fixtures on its own; it is only ever compared with the pinned f32 oracle.
 */
#include "brush_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

#define TILE_WIDTH 16u
#define TILE_SIZE 256u
#define COV_BLUR 0.3

typedef struct { double m[3][3]; } dmat3;

static inline dmat3 dmul(dmat3 a, dmat3 b) {
    dmat3 c;
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++)
            c.m[i][j] = a.m[i][0] * b.m[0][j] + a.m[i][1] * b.m[1][j] + a.m[i][2] * b.m[2][j];
    return c;
}
static inline dmat3 dtr(dmat3 a) {
    dmat3 c;
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) c.m[i][j] = a.m[j][i];
    return c;
}
static inline dmat3 dadd(dmat3 a, dmat3 b) {
    dmat3 c;
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) c.m[i][j] = a.m[i][j] + b.m[i][j];
    return c;
}
static inline dmat3 dview_rot(const float *vm) {
    dmat3 w;
    for (int r = 0; r < 3; r++)
        for (int c = 0; c < 3; c++) w.m[r][c] = (double)vm[c * 4 + r];
    return w;
}
/* helpers.wgsl:74-109, (w,x,y,z) */
static inline dmat3 dquat_to_rotmat(const double *q) {
    double w = q[0], x = q[1], y = q[2], z = q[3];
    dmat3 r;
    r.m[0][0] = 1.0 - 2.0 * (y * y + z * z);
    r.m[1][0] = 2.0 * (x * y + w * z);
    r.m[2][0] = 2.0 * (x * z - w * y);
    r.m[0][1] = 2.0 * (x * y - w * z);
    r.m[1][1] = 1.0 - 2.0 * (x * x + z * z);
    r.m[2][1] = 2.0 * (y * z + w * x);
    r.m[0][2] = 2.0 * (x * z + w * y);
    r.m[1][2] = 2.0 * (y * z - w * x);
    r.m[2][2] = 1.0 - 2.0 * (x * x + y * y);
    return r;
}
static inline double dclamp(double x, double lo, double hi) { return fmin(fmax(x, lo), hi); }

/* helpers.wgsl:124-158 */
static void dcalc_cov2d(const OracleUniforms *u, const double *p_view, const double *scale, const double *quat,
                        double *cov2d) {
    double focal[2] = {u->focal[0], u->focal[1]};
    double img[2] = {(double)u->img_size[0], (double)u->img_size[1]};
    double pc[2] = {u->pixel_center[0], u->pixel_center[1]};
    double t[2];
    double rz = 1.0 / p_view[2], rz2 = rz * rz;
    for (int i = 0; i < 2; i++) {
        double tan_fov = 0.5 * img[i] / focal[i];
        double lp = (img[i] - pc[i]) / focal[i] + 0.3 * tan_fov, ln = pc[i] / focal[i] + 0.3 * tan_fov;
        t[i] = p_view[2] * dclamp(p_view[i] * rz, -ln, lp);
    }
    dmat3 M = dquat_to_rotmat(quat);
    for (int r = 0; r < 3; r++)
        for (int c = 0; c < 3; c++) M.m[r][c] *= scale[c];
    double j00 = focal[0] * rz, j11 = focal[1] * rz;
    double j02 = -focal[0] * t[0] * rz2, j12 = -focal[1] * t[1] * rz2;
    dmat3 W = dview_rot(u->viewmat);
    dmat3 V = dmul(M, dtr(M));
    double T[2][3], TV[2][3];
    for (int c = 0; c < 3; c++) {
        T[0][c] = j00 * W.m[0][c] + j02 * W.m[2][c];
        T[1][c] = j11 * W.m[1][c] + j12 * W.m[2][c];
    }
    for (int r = 0; r < 2; r++)
        for (int c = 0; c < 3; c++) TV[r][c] = T[r][0] * V.m[0][c] + T[r][1] * V.m[1][c] + T[r][2] * V.m[2][c];
    cov2d[0] = TV[0][0] * T[0][0] + TV[0][1] * T[0][1] + TV[0][2] * T[0][2] + COV_BLUR;
    cov2d[1] = TV[1][0] * T[0][0] + TV[1][1] * T[0][1] + TV[1][2] * T[0][2];
    cov2d[2] = TV[1][0] * T[1][0] + TV[1][1] * T[1][1] + TV[1][2] * T[1][2] + COV_BLUR;
}

/* project_visible.wgsl:51-147 (Sloan's basis, same constants as brush_oracle.c:sh_basis) */
static void dsh_basis(uint32_t degree, const double *d, double *Y) {
    double x = d[0], y = d[1], z = d[2];
    Y[0] = 0.2820947917738781;
    if (degree == 0) return;
    double fTmp0A = 0.48860251190292;
    Y[2] = fTmp0A * z; Y[3] = -fTmp0A * x; Y[1] = -fTmp0A * y;
    if (degree == 1) return;
    double z2 = z * z;
    double fTmp0B = -1.092548430592079 * z;
    double fTmp1A = 0.5462742152960395;
    double fC1 = x * x - y * y, fS1 = 2.0 * x * y;
    Y[6] = 0.9461746957575601 * z2 - 0.3153915652525201;
    Y[7] = fTmp0B * x; Y[5] = fTmp0B * y; Y[8] = fTmp1A * fC1; Y[4] = fTmp1A * fS1;
    if (degree == 2) return;
    double fTmp0C = -2.285228997322329 * z2 + 0.4570457994644658;
    double fTmp1B = 1.445305721320277 * z;
    double fTmp2A = -0.5900435899266435;
    double fC2 = x * fC1 - y * fS1, fS2 = x * fS1 + y * fC1;
    Y[12] = z * (1.865881662950577 * z2 - 1.119528997770346);
    Y[13] = fTmp0C * x; Y[11] = fTmp0C * y; Y[14] = fTmp1B * fC1; Y[10] = fTmp1B * fS1;
    Y[15] = fTmp2A * fC2; Y[9] = fTmp2A * fS2;
    if (degree == 3) return;
    double fTmp0D = z * (-4.683325804901025 * z2 + 2.007139630671868);
    double fTmp1C = 3.31161143515146 * z2 - 0.47308734787878;
    double fTmp2B = -1.770130769779931 * z;
    double fTmp3A = 0.6258357354491763;
    double fC3 = x * fC2 - y * fS2, fS3 = x * fS2 + y * fC2;
    Y[20] = 1.984313483298443 * z * Y[12] - 1.006230589874905 * Y[6];
    Y[21] = fTmp0D * x; Y[19] = fTmp0D * y; Y[22] = fTmp1C * fC1; Y[18] = fTmp1C * fS1;
    Y[23] = fTmp2B * fC2; Y[17] = fTmp2B * fS2; Y[24] = fTmp3A * fC3; Y[16] = fTmp3A * fS3;
}

/* Magnitude companion of dsh_basis: every monomial with |x|, |y|, |z| and every subtraction turned into an addition,
 * i.e. the sum of the magnitudes of the terms an f32 evaluation of Y_k adds up.  Where a basis polynomial passes
 * through zero (Y6 = 0.946 z^2 - 0.315 at z^2 = 1/3, ...) its f32 rounding error is eps * this, not eps * |Y_k|. */
static void dsh_basis_mag(uint32_t degree, const double *d, double *Y) {
    double x = fabs(d[0]), y = fabs(d[1]), z = fabs(d[2]);
    Y[0] = 0.2820947917738781;
    if (degree == 0) return;
    double fTmp0A = 0.48860251190292;
    Y[2] = fTmp0A * z; Y[3] = fTmp0A * x; Y[1] = fTmp0A * y;
    if (degree == 1) return;
    double z2 = z * z;
    double fTmp0B = 1.092548430592079 * z;
    double fTmp1A = 0.5462742152960395;
    double fC1 = x * x + y * y, fS1 = 2.0 * x * y;
    Y[6] = 0.9461746957575601 * z2 + 0.3153915652525201;
    Y[7] = fTmp0B * x; Y[5] = fTmp0B * y; Y[8] = fTmp1A * fC1; Y[4] = fTmp1A * fS1;
    if (degree == 2) return;
    double fTmp0C = 2.285228997322329 * z2 + 0.4570457994644658;
    double fTmp1B = 1.445305721320277 * z;
    double fTmp2A = 0.5900435899266435;
    double fC2 = x * fC1 + y * fS1, fS2 = x * fS1 + y * fC1;
    Y[12] = z * (1.865881662950577 * z2 + 1.119528997770346);
    Y[13] = fTmp0C * x; Y[11] = fTmp0C * y; Y[14] = fTmp1B * fC1; Y[10] = fTmp1B * fS1;
    Y[15] = fTmp2A * fC2; Y[9] = fTmp2A * fS2;
    if (degree == 3) return;
    double fTmp0D = z * (4.683325804901025 * z2 + 2.007139630671868);
    double fTmp1C = 3.31161143515146 * z2 + 0.47308734787878;
    double fTmp2B = 1.770130769779931 * z;
    double fTmp3A = 0.6258357354491763;
    double fC3 = x * fC2 + y * fS2, fS3 = x * fS2 + y * fC2;
    Y[20] = 1.984313483298443 * z * Y[12] + 1.006230589874905 * Y[6];
    Y[21] = fTmp0D * x; Y[19] = fTmp0D * y; Y[22] = fTmp1C * fC1; Y[18] = fTmp1C * fS1;
    Y[23] = fTmp2B * fC2; Y[17] = fTmp2B * fS2; Y[24] = fTmp3A * fC3; Y[16] = fTmp3A * fS3;
}

/* rasterize_backwards.wgsl:140-304, one tile; values f64, decisions as the f32 restatement. */
static void d_rasterize_backward_tile(const OracleUniforms *u, uint32_t tile_id, const uint32_t *gid_from_isect,
                                      const uint32_t *tile_bins, const float *projected,
                                      const uint32_t *final_index, const float *out_img, const float *v_out,
                                      double *rows /* [I][9] */, double *rows_abs /* [I][9] sums of |terms| */,
                                      double *rows_flip /* [I][9] what flipped threshold decisions can move */,
                                      double *rows_dep /* [I][9] sums of |terms| x (divisions T went through) */,
                                      const double *pix_weight /* optional [h,w], see oracle_render_backward_f64 */,
                                      const uint32_t *final_index_alt /* optional [h,w], likewise */,
                                      const float *out_img_alt /* optional [h,w,4], likewise */) {
    uint32_t w = u->img_size[0], h = u->img_size[1], tbx = u->tile_bounds[0];
    uint32_t tile_x = tile_id % tbx, tile_y = tile_id / tbx;
    uint32_t r0 = tile_bins[tile_id * 2], r1 = tile_bins[tile_id * 2 + 1];
    if (r1 <= r0) return;
    double T[TILE_SIZE], T_final[TILE_SIZE], buf[TILE_SIZE][3], vo[TILE_SIZE][4];
    float pcx[TILE_SIZE], pcy[TILE_SIZE];
    uint32_t fin[TILE_SIZE], fin_lo[TILE_SIZE], fin_hi[TILE_SIZE];
    uint8_t inside[TILE_SIZE];
    /* relative change of T at pixel l if the threshold decisions met so far on it went the other way */
    double flip_w[TILE_SIZE], pix_w[TILE_SIZE];
    /* Stop mismatch (final_index_alt): the entries between the two stops are walked by ONE of the two backwards.  They
     * leave T of the entries walked later unchanged (each run divides back exactly the factors its own forward
     * multiplied in), but that run's colour accumulator carries their contributions and its T_final their factors:
     * dbuf = sum |rgb_k| alpha_k T_k per channel, dtfin = relative change of T_final.  Round 3 priced these entries as
     * a RELATIVE change alpha/(1 - alpha) of every later term, i.e. in units of the later entry's OWN colour: short
     * wherever the skipped entries are brighter than the entry under test (1.53x on the 556-entry lists of c5). */
    double dbuf[TILE_SIZE][3], dtfin[TILE_SIZE];
    /* ... and each run's T at the LAST COMMON entry is its own forward's (the two forwards reach that entry with
     * transmittances that differ by their accumulated rounding, ~1e-5 relative after hundreds of factors near the
     * 1e-4 stop): recovered here as T_final x prod 1/(1 - alpha) over the entries in between and priced like pix_weight
     * (3 |dT| / min T on every later term of the pixel) once the walk reaches the common part. */
    double tprod[TILE_SIZE], t_alt[TILE_SIZE];
    uint8_t mismatch_open[TILE_SIZE];
    double ntaken[TILE_SIZE]; /* roundings (in eps) the recovered T of pixel l has gone through so far */
    for (uint32_t l = 0; l < TILE_SIZE; l++) {
        uint32_t px = tile_x * TILE_WIDTH + l % TILE_WIDTH, py = tile_y * TILE_WIDTH + l / TILE_WIDTH;
        inside[l] = px < w && py < h;
        pcx[l] = (float)px + 0.5f; pcy[l] = (float)py + 0.5f;
        flip_w[l] = 0.0;
        pix_w[l] = 0.0;
        ntaken[l] = 0.0;
        dbuf[l][0] = dbuf[l][1] = dbuf[l][2] = dtfin[l] = 0.0;
        tprod[l] = 1.0; t_alt[l] = 0.0; mismatch_open[l] = 0;
        buf[l][0] = buf[l][1] = buf[l][2] = 0.0;
        fin[l] = fin_lo[l] = fin_hi[l] = 0; T[l] = T_final[l] = 1.0;
        vo[l][0] = vo[l][1] = vo[l][2] = vo[l][3] = 0.0;
        if (inside[l]) {
            size_t pix = (size_t)px + (size_t)py * w;
            T_final[l] = 1.0 - (double)out_img[pix * 4 + 3];
            T[l] = T_final[l];
            fin[l] = fin_lo[l] = fin_hi[l] = final_index[pix];
            if (final_index_alt) {
                uint32_t alt = final_index_alt[pix];
                fin_lo[l] = alt < fin[l] ? alt : fin[l];
                fin_hi[l] = alt > fin[l] ? alt : fin[l];
            }
            if (pix_weight) pix_w[l] = pix_weight[pix];
            if (final_index_alt && out_img_alt && fin_lo[l] != fin_hi[l]) {
                mismatch_open[l] = 1;
                t_alt[l] = 1.0 - (double)out_img_alt[pix * 4 + 3];
            }
            for (int k = 0; k < 4; k++) vo[l][k] = (double)v_out[pix * 4 + k];
        }
    }
    for (uint32_t i = r1; i-- > r0;) {
        const float *p = projected + (size_t)gid_from_isect[i] * 9;
        double s[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}, sa[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
        double sf[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}, sd[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
        for (uint32_t l = 0; l < TILE_SIZE; l++) {
            if (!(inside[l] && i <= fin_hi[l])) continue;
            /* decision: exactly brush_oracle.c (f32) */
            float fdx = p[0] - pcx[l], fdy = p[1] - pcy[l];
            float fsigma = 0.5f * (p[2] * fdx * fdx + p[4] * fdy * fdy) + p[3] * fdx * fdy;
            float falpha = fminf(0.99f, p[8] * expf(-fsigma));
            int take = fsigma >= 0.0f && falpha >= 1.0f / 255.0f && i <= fin[l];
            /* A decision another f32 evaluation (different exp, contraction, association) may take the other
             * way.  sigma is a sum of terms of total magnitude fterms, so two f32 evaluations of it differ by up to
             * a few eps * fterms (far more than eps * sigma when the conic is strongly correlated and the terms
             * cancel); alpha = opac exp(-sigma) inherits that absolute error as a relative one, plus ~1e-6 for the
             * exponential itself (v_exp_f32 on an f32 argument of magnitude <= 8). */
            float fterms = 0.5f * (fabsf(p[2] * fdx * fdx) + fabsf(p[4] * fdy * fdy)) + fabsf(p[3] * fdx * fdy);
            float dsig = 4.8e-7f * fterms; /* 8 eps32 * fterms */
            int risky = (fabsf(falpha * 255.0f - 1.0f) < 1e-6f + dsig && fsigma >= -dsig) ||
                        (fabsf(fsigma) <= dsig && p[8] >= 1.0f / 255.0f);
            /* an entry only ONE of the two forward states walks (the saturation stop of rasterize.wgsl:88-91 fell on a
             * different entry): present in one backward, absent from the other */
            const int between = i > fin_lo[l] && fsigma >= 0.0f && falpha >= 1.0f / 255.0f;
            if (between) risky = 1;
            if (!take && !risky) continue;
            /* values: f64 */
            double dx = (double)p[0] - (double)pcx[l], dy = (double)p[1] - (double)pcy[l];
            double a = p[2], b = p[3], c = p[4], opac = p[8];
            double sigma = 0.5 * (a * dx * dx + c * dy * dy) + b * dx * dy;
            double vis = exp(-sigma);
            double alpha = fmin(0.99, opac * vis);
            double ra = 1.0 / (1.0 - alpha);
            double Tn = T[l] * ra;
            double fac = alpha * Tn;
            if (mismatch_open[l] && take && !between) {
                /* first entry both runs walk: T after it in this run vs in the other one */
                double t_here = T[l], t_other = (fin[l] == fin_hi[l]) ? t_alt[l] : t_alt[l] * tprod[l];
                double lo_t = fmin(t_here, t_other);
                if (lo_t > 0.0) pix_w[l] += 3.0 * fabs(t_here - t_other) / lo_t;
                mismatch_open[l] = 0;
            }
            double v_alpha = (p[5] * Tn - buf[l][0] * ra) * vo[l][0] + (p[6] * Tn - buf[l][1] * ra) * vo[l][1] +
                             (p[7] * Tn - buf[l][2] * ra) * vo[l][2] + T_final[l] * ra * vo[l][3];
            double v_sigma = -opac * vis * v_alpha;
            /* magnitudes of the same terms: v_alpha is itself a sum of signed pieces, so its magnitude is
             * taken piecewise (what an error in T or alpha multiplies) */
            double m_alpha = (fabs(p[5] * Tn) + fabs(buf[l][0] * ra)) * fabs(vo[l][0]) +
                             (fabs(p[6] * Tn) + fabs(buf[l][1] * ra)) * fabs(vo[l][1]) +
                             (fabs(p[7] * Tn) + fabs(buf[l][2] * ra)) * fabs(vo[l][2]) + fabs(T_final[l] * ra * vo[l][3]);
            double m_sigma = opac * vis * m_alpha;
            double ma[9] = {m_sigma * (fabs(a * dx) + fabs(b * dy)), m_sigma * (fabs(b * dx) + fabs(c * dy)),
                            0.5 * m_sigma * dx * dx, m_sigma * fabs(dx * dy), 0.5 * m_sigma * dy * dy,
                            fabs(fac * vo[l][0]), fabs(fac * vo[l][1]), fabs(fac * vo[l][2]), vis * m_alpha};
            if (take) { /* a risky record the f32 decision skips leaves the pixel's state untouched */
                T[l] = Tn;
                buf[l][0] += p[5] * fac; buf[l][1] += p[6] * fac; buf[l][2] += p[7] * fac;
                s[0] += v_sigma * (a * dx + b * dy);
                s[1] += v_sigma * (b * dx + c * dy);
                s[2] += 0.5 * v_sigma * dx * dx;
                s[3] += v_sigma * dx * dy;
                s[4] += 0.5 * v_sigma * dy * dy;
                s[5] += fac * vo[l][0]; s[6] += fac * vo[l][1]; s[7] += fac * vo[l][2];
                s[8] += vis * v_alpha;
                for (int k = 0; k < 9; k++) sa[k] += ma[k];
                /* T (and with it every term of this entry) is T_final divided by the f32 (1 - alpha) of every entry walked
                 * so far (rasterize_backwards.wgsl:244-246).  Each division costs a rounding, and (1 - alpha) itself is
                 * the difference of 1 and a rounded alpha: its relative error is eps alpha / (1 - alpha), up to 99 eps
                 * at the 0.99 clamp.  ntaken counts those roundings in units of eps. */
                ntaken[l] += 1.0 + alpha * ra;
                for (int k = 0; k < 9; k++) sd[k] += ma[k] * ntaken[l];
            }
            /* flips: this record's own terms if ITS decision is risky (taken or not: the other evaluation does the
             * opposite); every term of a pixel whose T already carries possible flips of relative size flip_w */
            double wgt = (risky ? 1.0 : 0.0) + (take ? flip_w[l] + pix_w[l] : 0.0);
            if (wgt > 0.0)
                for (int k = 0; k < 9; k++) sf[k] += wgt * ma[k];
            if (take && !between && (dtfin[l] > 0.0 || dbuf[l][0] + dbuf[l][1] + dbuf[l][2] > 0.0)) {
                /* an entry both runs walk, behind a stop mismatch: v_alpha of the other run differs by the skipped
                 * entries' colour in buf and by their factors in T_final (rasterize_backwards.wgsl:229-246) */
                double d_alpha = (dbuf[l][0] * fabs(vo[l][0]) + dbuf[l][1] * fabs(vo[l][1]) + dbuf[l][2] * fabs(vo[l][2])) * ra +
                                 fabs(T_final[l] * ra * vo[l][3]) * dtfin[l];
                double d_sigma = opac * vis * d_alpha;
                sf[0] += d_sigma * (fabs(a * dx) + fabs(b * dy));
                sf[1] += d_sigma * (fabs(b * dx) + fabs(c * dy));
                sf[2] += 0.5 * d_sigma * dx * dx;
                sf[3] += d_sigma * fabs(dx * dy);
                sf[4] += 0.5 * d_sigma * dy * dy;
                sf[8] += vis * d_alpha;
            }
            if (between) {
                /* T before this entry in the run that walks it: T[l] here if that run is this one (take), at most
                 * T_final (of the run that stopped earlier) otherwise */
                double fk = alpha * (take ? Tn : T_final[l] * ra) * 1.05;
                dbuf[l][0] += fabs(p[5]) * fk; dbuf[l][1] += fabs(p[6]) * fk; dbuf[l][2] += fabs(p[7]) * fk;
                dtfin[l] += alpha * ra * 1.05;
                tprod[l] *= ra;
            } else if (risky) {
                flip_w[l] += alpha * ra * 1.05; /* T of the later-walked entries changes by 1/(1 - alpha) */
            }
        }
        for (int k = 0; k < 9; k++) rows[(size_t)i * 9 + k] = s[k];
        for (int k = 0; k < 9; k++) rows_abs[(size_t)i * 9 + k] = sa[k];
        for (int k = 0; k < 9; k++) rows_flip[(size_t)i * 9 + k] = sf[k];
        for (int k = 0; k < 9; k++) rows_dep[(size_t)i * 9 + k] = sd[k];
    }
}

/* project_backwards.wgsl:75-227 in f64 */
static void d_project_backward_one(const OracleUniforms *u, const float *fmean, const float *flog_scale,
                                   const float *fquat, const double *v_xy, const double *v_conic, double *v_mean,
                                   double *v_scale_out, double *v_quat) {
    double focal[2] = {u->focal[0], u->focal[1]};
    double mean[3] = {fmean[0], fmean[1], fmean[2]}, quat[4] = {fquat[0], fquat[1], fquat[2], fquat[3]};
    double scale[3] = {exp((double)flog_scale[0]), exp((double)flog_scale[1]), exp((double)flog_scale[2])};
    dmat3 W = dview_rot(u->viewmat);
    double p_view[3];
    for (int r = 0; r < 3; r++)
        p_view[r] = W.m[r][0] * mean[0] + W.m[r][1] * mean[1] + W.m[r][2] * mean[2] + (double)u->viewmat[12 + r];
    double rw = 1.0 / (p_view[2] + 1e-6);
    double vp0 = focal[0] * v_xy[0], vp1 = focal[1] * v_xy[1];
    double vp[3] = {vp0 * rw, vp1 * rw, -(vp0 * p_view[0] + vp1 * p_view[1]) * rw * rw};
    double vm[3];
    for (int i = 0; i < 3; i++) vm[i] = W.m[0][i] * vp[0] + W.m[1][i] * vp[1] + W.m[2][i] * vp[2];

    double cov2d[3];
    dcalc_cov2d(u, p_view, scale, quat, cov2d);
    double det = cov2d[0] * cov2d[2] - cov2d[1] * cov2d[1];
    double conic[3] = {cov2d[2] / det, -cov2d[1] / det, cov2d[0] / det};
    /* cov2d_to_conic_vjp (project_backwards.wgsl:59-72) */
    double X[2][2] = {{conic[0], conic[1]}, {conic[1], conic[2]}};
    double Gm[2][2] = {{v_conic[0], v_conic[1] / 2.0}, {v_conic[1] / 2.0, v_conic[2]}};
    double XG[2][2], S2[2][2];
    for (int i = 0; i < 2; i++)
        for (int j = 0; j < 2; j++) XG[i][j] = X[i][0] * Gm[0][j] + X[i][1] * Gm[1][j];
    for (int i = 0; i < 2; i++)
        for (int j = 0; j < 2; j++) S2[i][j] = XG[i][0] * X[0][j] + XG[i][1] * X[1][j];
    double v_cov2d[3] = {-S2[0][0], -(S2[0][1] + S2[1][0]), -S2[1][1]};

    double rz = 1.0 / p_view[2], rz2 = rz * rz, rz3 = rz2 * rz;
    /* quirk 3: J from the UNCLAMPED p_view (project_backwards.wgsl:134-138) */
    dmat3 J = {{{focal[0] * rz, 0.0, -focal[0] * p_view[0] * rz2}, {0.0, focal[1] * rz, -focal[1] * p_view[1] * rz2},
                {0.0, 0.0, 0.0}}};
    dmat3 R = dquat_to_rotmat(quat);
    dmat3 Sm = {{{scale[0], 0, 0}, {0, scale[1], 0}, {0, 0, scale[2]}}};
    dmat3 M = dmul(R, Sm);
    dmat3 V = dmul(M, dtr(M));
    dmat3 v_cov = {{{v_cov2d[0], 0.5 * v_cov2d[1], 0.0}, {0.5 * v_cov2d[1], v_cov2d[2], 0.0}, {0.0, 0.0, 0.0}}};
    dmat3 T = dmul(J, W);
    dmat3 v_V = dmul(dmul(dtr(T), v_cov), T);
    dmat3 v_T = dadd(dmul(dmul(v_cov, T), dtr(V)), dmul(dmul(dtr(v_cov), T), V));
    double c0 = v_V.m[0][0], c1 = v_V.m[1][0] + v_V.m[0][1], c2 = v_V.m[2][0] + v_V.m[0][2];
    double c3 = v_V.m[1][1], c4 = v_V.m[2][1] + v_V.m[1][2], c5 = v_V.m[2][2];
    dmat3 v_J = dmul(v_T, dtr(W));
    double vJ02 = v_J.m[0][2], vJ12 = v_J.m[1][2], vJ00 = v_J.m[0][0], vJ11 = v_J.m[1][1];
    double v_t[3];
    v_t[0] = -focal[0] * rz2 * vJ02;
    v_t[1] = -focal[1] * rz2 * vJ12;
    v_t[2] = -focal[0] * rz2 * vJ00 + 2.0 * focal[0] * p_view[0] * rz3 * vJ02 - focal[1] * rz2 * vJ11 +
             2.0 * focal[1] * p_view[1] * rz3 * vJ12;
    for (int i = 0; i < 3; i++) vm[i] += v_t[0] * W.m[0][i] + v_t[1] * W.m[1][i] + v_t[2] * W.m[2][i];
    dmat3 two_vVs = {{{2.0 * c0, c1, c2}, {c1, 2.0 * c3, c4}, {c2, c4, 2.0 * c5}}};
    dmat3 v_M = dmul(two_vVs, M);
    for (int j = 0; j < 3; j++) {
        double vs = R.m[0][j] * v_M.m[0][j] + R.m[1][j] * v_M.m[1][j] + R.m[2][j] * v_M.m[2][j];
        v_scale_out[j] = vs * scale[j];
    }
    dmat3 vR = dmul(v_M, Sm);
#define G(a, b) (vR.m[b][a])
    double w = quat[0], x = quat[1], y = quat[2], z = quat[3];
    v_quat[0] = 2.0 * (x * (G(1, 2) - G(2, 1)) + y * (G(2, 0) - G(0, 2)) + z * (G(0, 1) - G(1, 0)));
    v_quat[1] = 2.0 * (-2.0 * x * (G(1, 1) + G(2, 2)) + y * (G(0, 1) + G(1, 0)) + z * (G(0, 2) + G(2, 0)) +
                       w * (G(1, 2) - G(2, 1)));
    v_quat[2] = 2.0 * (x * (G(0, 1) + G(1, 0)) - 2.0 * y * (G(0, 0) + G(2, 2)) + z * (G(1, 2) + G(2, 1)) +
                       w * (G(2, 0) - G(0, 2)));
    v_quat[3] = 2.0 * (x * (G(0, 2) + G(2, 0)) + y * (G(1, 2) + G(2, 1)) - 2.0 * z * (G(0, 0) + G(1, 1)) +
                       w * (G(0, 1) - G(1, 0)));
#undef G
    v_mean[0] = vm[0]; v_mean[1] = vm[1]; v_mean[2] = vm[2];
}

/* Magnitude companion of d_project_backward_one: the same chain of products and sums with every operand
 * replaced by its absolute value and every subtraction by an addition, i.e. for each output the sum of the
 * magnitudes of the terms an f32 evaluation of the VJP adds up.  The projection VJP cancels terms of size
 * scale^2 against each other (v_V = T^t v_cov T, the column dot products of v_scale), so its own rounding
 * noise is eps * this, which can exceed the propagated input error by orders of magnitude. */
static dmat3 dabs3(dmat3 a) {
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) a.m[i][j] = fabs(a.m[i][j]);
    return a;
}
static void d_project_backward_mag(const OracleUniforms *u, const float *fmean, const float *flog_scale,
                                   const float *fquat, const double *v_xy, const double *v_conic, double *m_mean,
                                   double *m_scale, double *m_quat) {
    double focal[2] = {fabs(u->focal[0]), fabs(u->focal[1])};
    double mean[3] = {fmean[0], fmean[1], fmean[2]}, quat[4] = {fquat[0], fquat[1], fquat[2], fquat[3]};
    double scale[3] = {exp((double)flog_scale[0]), exp((double)flog_scale[1]), exp((double)flog_scale[2])};
    dmat3 Ws = dview_rot(u->viewmat), W = dabs3(Ws);
    double p_view[3];
    for (int r = 0; r < 3; r++)
        p_view[r] = Ws.m[r][0] * mean[0] + Ws.m[r][1] * mean[1] + Ws.m[r][2] * mean[2] + (double)u->viewmat[12 + r];
    double ap[3] = {fabs(p_view[0]), fabs(p_view[1]), fabs(p_view[2])};
    double rw = 1.0 / fabs(p_view[2] + 1e-6);
    double vp0 = focal[0] * fabs(v_xy[0]), vp1 = focal[1] * fabs(v_xy[1]);
    double vp[3] = {vp0 * rw, vp1 * rw, (vp0 * ap[0] + vp1 * ap[1]) * rw * rw};
    double vm[3];
    for (int i = 0; i < 3; i++) vm[i] = W.m[0][i] * vp[0] + W.m[1][i] * vp[1] + W.m[2][i] * vp[2];
    double cov2d[3];
    dcalc_cov2d(u, p_view, scale, quat, cov2d);
    double det = cov2d[0] * cov2d[2] - cov2d[1] * cov2d[1];
    double conic[3] = {fabs(cov2d[2] / det), fabs(cov2d[1] / det), fabs(cov2d[0] / det)};
    double X[2][2] = {{conic[0], conic[1]}, {conic[1], conic[2]}};
    double Gm[2][2] = {{fabs(v_conic[0]), fabs(v_conic[1]) / 2.0}, {fabs(v_conic[1]) / 2.0, fabs(v_conic[2])}};
    double XG[2][2], S2[2][2];
    for (int i = 0; i < 2; i++)
        for (int j = 0; j < 2; j++) XG[i][j] = X[i][0] * Gm[0][j] + X[i][1] * Gm[1][j];
    for (int i = 0; i < 2; i++)
        for (int j = 0; j < 2; j++) S2[i][j] = XG[i][0] * X[0][j] + XG[i][1] * X[1][j];
    double v_cov2d[3] = {S2[0][0], S2[0][1] + S2[1][0], S2[1][1]};
    double rz = 1.0 / ap[2], rz2 = rz * rz, rz3 = rz2 * rz;
    dmat3 J = {{{focal[0] * rz, 0.0, focal[0] * ap[0] * rz2}, {0.0, focal[1] * rz, focal[1] * ap[1] * rz2}, {0.0, 0.0, 0.0}}};
    dmat3 R = dabs3(dquat_to_rotmat(quat));
    dmat3 Sm = {{{scale[0], 0, 0}, {0, scale[1], 0}, {0, 0, scale[2]}}};
    dmat3 M = dmul(R, Sm);
    dmat3 V = dmul(M, dtr(M));
    dmat3 v_cov = {{{v_cov2d[0], 0.5 * v_cov2d[1], 0.0}, {0.5 * v_cov2d[1], v_cov2d[2], 0.0}, {0.0, 0.0, 0.0}}};
    dmat3 T = dmul(J, W);
    dmat3 v_V = dmul(dmul(dtr(T), v_cov), T);
    dmat3 v_T = dadd(dmul(dmul(v_cov, T), dtr(V)), dmul(dmul(dtr(v_cov), T), V));
    double c0 = v_V.m[0][0], c1 = v_V.m[1][0] + v_V.m[0][1], c2 = v_V.m[2][0] + v_V.m[0][2];
    double c3 = v_V.m[1][1], c4 = v_V.m[2][1] + v_V.m[1][2], c5 = v_V.m[2][2];
    dmat3 v_J = dmul(v_T, dtr(W));
    double vJ02 = v_J.m[0][2], vJ12 = v_J.m[1][2], vJ00 = v_J.m[0][0], vJ11 = v_J.m[1][1];
    double v_t[3];
    v_t[0] = focal[0] * rz2 * vJ02;
    v_t[1] = focal[1] * rz2 * vJ12;
    v_t[2] = focal[0] * rz2 * vJ00 + 2.0 * focal[0] * ap[0] * rz3 * vJ02 + focal[1] * rz2 * vJ11 +
             2.0 * focal[1] * ap[1] * rz3 * vJ12;
    for (int i = 0; i < 3; i++) m_mean[i] = vm[i] + v_t[0] * W.m[0][i] + v_t[1] * W.m[1][i] + v_t[2] * W.m[2][i];
    dmat3 two_vVs = {{{2.0 * c0, c1, c2}, {c1, 2.0 * c3, c4}, {c2, c4, 2.0 * c5}}};
    dmat3 v_M = dmul(two_vVs, M);
    for (int j = 0; j < 3; j++)
        m_scale[j] = (R.m[0][j] * v_M.m[0][j] + R.m[1][j] * v_M.m[1][j] + R.m[2][j] * v_M.m[2][j]) * scale[j];
    dmat3 vR = dmul(v_M, Sm);
#define G(a, b) (vR.m[b][a])
    double w = fabs(quat[0]), x = fabs(quat[1]), y = fabs(quat[2]), z = fabs(quat[3]);
    m_quat[0] = 2.0 * (x * (G(1, 2) + G(2, 1)) + y * (G(2, 0) + G(0, 2)) + z * (G(0, 1) + G(1, 0)));
    m_quat[1] = 2.0 * (2.0 * x * (G(1, 1) + G(2, 2)) + y * (G(0, 1) + G(1, 0)) + z * (G(0, 2) + G(2, 0)) +
                       w * (G(1, 2) + G(2, 1)));
    m_quat[2] = 2.0 * (x * (G(0, 1) + G(1, 0)) + 2.0 * y * (G(0, 0) + G(2, 2)) + z * (G(1, 2) + G(2, 1)) +
                       w * (G(2, 0) + G(0, 2)));
    m_quat[3] = 2.0 * (x * (G(0, 2) + G(2, 0)) + y * (G(1, 2) + G(2, 1)) + 2.0 * z * (G(0, 0) + G(1, 1)) +
                       w * (G(0, 1) + G(1, 0)));
#undef G
}

/* All outputs f64, dense [N,..], zero for non-visible splats (render.rs:468-626).  mag_* (optional, all or
 * none): for every output element the sum of the MAGNITUDES of the per-pixel terms it is made of, carried
 * through the absolute value of the (linear) gather / projection VJP: an f32 evaluation whose per-term
 * relative error is eps differs from the exact value by at most eps * mag, whatever its summation order.
 * flip_* : how far the element can move when the threshold decisions that sit within f32 rounding of
 * flipping (alpha ~ 1/255, sigma ~ 0) go the other way in another f32 evaluation: the flipped entry's own
 * terms plus alpha/(1-alpha) of every term the pixel contributes afterwards (T is recovered by division).
 * vjp_*  : magnitude of the terms the projection VJP itself adds up for the element (d_project_backward_mag); for
 *   v_sh the terms of the SH basis polynomial, for v_opac the saturating sigmoid's (1 - s).
 * dep_*  : like mag_*, every term weighted by the number of f32 divisions its recovered T has gone through.
 * pix_weight (optional, [h,w]): a relative uncertainty of the pixel's forward state (every term of a pixel is
 *   proportional to T_final = 1 - out.a, rasterize_backwards.wgsl:163,173); weight x |terms| of the pixel is added to
 *   flip_*.  Used when the gradients under test come from a DIFFERENT forward state than the one passed here.
 * final_index_alt (optional, [h,w]): that other forward state's final_index.  Where the two differ (the saturation
 *   stop `T (1 - alpha) <= 1e-4`, rasterize.wgsl:88-91, fell on a different entry) the entries between them exist in
 *   one backward only: their own terms go to flip_* in full; the entries both runs walk get the skipped entries'
 *   colour in the accumulator and their factors in T_final (dbuf / dtfin), and — with out_img_alt, the other state's
 *   image — the difference of the two runs' recovered T at the last common entry. */
int oracle_render_backward_f64(const OracleUniforms *u_in, const OracleAux *aux, const float *means,
                               const float *log_scales, const float *quats, const float *raw_opac, uint32_t n,
                               const float *out_img, const float *v_out, double *v_means, double *v_xy,
                               double *v_scales, double *v_quats, double *v_sh, double *v_opac, double *mag_means,
                               double *mag_xy, double *mag_scales, double *mag_quats, double *mag_sh,
                               double *mag_opac, double *flip_means, double *flip_xy, double *flip_scales,
                               double *flip_quats, double *flip_sh, double *flip_opac, double *vjp_means,
                               double *vjp_scales, double *vjp_quats, const double *pix_weight,
                               const uint32_t *final_index_alt, double *dep_means, double *dep_xy, double *dep_scales,
                               double *dep_quats, double *dep_sh, double *dep_opac, double *vjp_sh, double *vjp_opac,
                               const float *out_img_alt) {
    OracleUniforms uu = *u_in;
    uu.total_splats = n;
    const OracleUniforms *u = &uu;
    uint32_t V = aux->num_visible[0], I = aux->num_intersections[0];
    uint32_t ncoef = (u->sh_degree + 1) * (u->sh_degree + 1);
    uint32_t num_tiles = u->tile_bounds[0] * u->tile_bounds[1];
    double *rows = (double *)calloc((size_t)(I ? I : 1) * 9, sizeof(double));
    double *rows_abs = (double *)calloc((size_t)(I ? I : 1) * 9, sizeof(double));
    double *acc = (double *)calloc((size_t)(V ? V : 1) * 9, sizeof(double));
    double *acc_abs = (double *)calloc((size_t)(V ? V : 1) * 9, sizeof(double));
    double *rows_flip = (double *)calloc((size_t)(I ? I : 1) * 9, sizeof(double));
    double *acc_flip = (double *)calloc((size_t)(V ? V : 1) * 9, sizeof(double));
    double *rows_dep = (double *)calloc((size_t)(I ? I : 1) * 9, sizeof(double));
    double *acc_dep = (double *)calloc((size_t)(V ? V : 1) * 9, sizeof(double));
#pragma omp parallel for schedule(dynamic, 4)
    for (int64_t t = 0; t < (int64_t)num_tiles; t++)
        d_rasterize_backward_tile(u, (uint32_t)t, aux->compact_gid_from_isect, aux->tile_bins, aux->projected_splats,
                                  aux->final_index, out_img, v_out, rows, rows_abs, rows_flip, rows_dep, pix_weight,
                                  final_index_alt, out_img_alt);
    for (size_t i = 0; i < I; i++) { /* fixed order: ascending intersection id */
        double *a = acc + (size_t)aux->compact_gid_from_isect[i] * 9;
        double *m = acc_abs + (size_t)aux->compact_gid_from_isect[i] * 9;
        for (int k = 0; k < 9; k++) a[k] += rows[i * 9 + k];
        for (int k = 0; k < 9; k++) m[k] += rows_abs[i * 9 + k];
        double *f = acc_flip + (size_t)aux->compact_gid_from_isect[i] * 9;
        for (int k = 0; k < 9; k++) f[k] += rows_flip[i * 9 + k];
        double *dp = acc_dep + (size_t)aux->compact_gid_from_isect[i] * 9;
        for (int k = 0; k < 9; k++) dp[k] += rows_dep[i * 9 + k];
    }
    if (mag_means) {
        memset(mag_means, 0, sizeof(double) * 3 * n);
        memset(mag_xy, 0, sizeof(double) * 2 * n);
        memset(mag_scales, 0, sizeof(double) * 3 * n);
        memset(mag_quats, 0, sizeof(double) * 4 * n);
        memset(mag_sh, 0, sizeof(double) * 3 * (size_t)ncoef * n);
        memset(mag_opac, 0, sizeof(double) * n);
        memset(flip_means, 0, sizeof(double) * 3 * n);
        memset(flip_xy, 0, sizeof(double) * 2 * n);
        memset(flip_scales, 0, sizeof(double) * 3 * n);
        memset(flip_quats, 0, sizeof(double) * 4 * n);
        memset(flip_sh, 0, sizeof(double) * 3 * (size_t)ncoef * n);
        memset(flip_opac, 0, sizeof(double) * n);
        memset(vjp_means, 0, sizeof(double) * 3 * n);
        memset(vjp_scales, 0, sizeof(double) * 3 * n);
        memset(vjp_quats, 0, sizeof(double) * 4 * n);
        memset(dep_means, 0, sizeof(double) * 3 * n);
        memset(dep_xy, 0, sizeof(double) * 2 * n);
        memset(dep_scales, 0, sizeof(double) * 3 * n);
        memset(dep_quats, 0, sizeof(double) * 4 * n);
        memset(dep_sh, 0, sizeof(double) * 3 * (size_t)ncoef * n);
        memset(dep_opac, 0, sizeof(double) * n);
        memset(vjp_sh, 0, sizeof(double) * 3 * (size_t)ncoef * n);
        memset(vjp_opac, 0, sizeof(double) * n);
    }
    memset(v_means, 0, sizeof(double) * 3 * n);
    memset(v_xy, 0, sizeof(double) * 2 * n);
    memset(v_scales, 0, sizeof(double) * 3 * n);
    memset(v_quats, 0, sizeof(double) * 4 * n);
    memset(v_sh, 0, sizeof(double) * 3 * (size_t)ncoef * n);
    memset(v_opac, 0, sizeof(double) * n);
#pragma omp parallel for schedule(static)
    for (int64_t c = 0; c < (int64_t)V; c++) {
        uint32_t g = aux->global_from_compact_gid[c];
        const double *a = acc + (size_t)c * 9;
        const float *mean = means + (size_t)g * 3;
        /* gather_grads.wgsl:165-232 */
        double dir[3] = {(double)mean[0] - u->viewmat[12], (double)mean[1] - u->viewmat[13],
                         (double)mean[2] - u->viewmat[14]};
        double len = sqrt(dir[0] * dir[0] + dir[1] * dir[1] + dir[2] * dir[2]);
        dir[0] /= len; dir[1] /= len; dir[2] /= len;
        double Y[25];
        dsh_basis(u->sh_degree, dir, Y);
        double *vs = v_sh + (size_t)g * ncoef * 3;
        for (uint32_t k = 0; k < ncoef; k++)
            for (int ch = 0; ch < 3; ch++) vs[k * 3 + ch] = Y[k] * a[5 + ch];
        double s = 1.0 / (1.0 + exp(-(double)raw_opac[g]));
        v_opac[g] = a[8] * (s * (1.0 - s));
        v_xy[(size_t)g * 2] = a[0];
        v_xy[(size_t)g * 2 + 1] = a[1];
        d_project_backward_one(u, mean, log_scales + (size_t)g * 3, quats + (size_t)g * 4, a, a + 2,
                               v_means + (size_t)g * 3, v_scales + (size_t)g * 3, v_quats + (size_t)g * 4);
        if (mag_means) {
            const double *m = acc_abs + (size_t)c * 9, *fl = acc_flip + (size_t)c * 9, *dp = acc_dep + (size_t)c * 9;
            double *ms = mag_sh + (size_t)g * ncoef * 3, *fs = flip_sh + (size_t)g * ncoef * 3;
            double *ds = dep_sh + (size_t)g * ncoef * 3, *vs2 = vjp_sh + (size_t)g * ncoef * 3;
            double Ym[25];
            dsh_basis_mag(u->sh_degree, dir, Ym);
            for (uint32_t k = 0; k < ncoef; k++)
                for (int ch = 0; ch < 3; ch++) {
                    ms[k * 3 + ch] = fabs(Y[k]) * m[5 + ch];
                    fs[k * 3 + ch] = fabs(Y[k]) * fl[5 + ch];
                    ds[k * 3 + ch] = fabs(Y[k]) * dp[5 + ch];
                    /* rounding noise of the SH basis itself (gather_grads.wgsl:17-112 evaluated in f32): the terms of
                     * Y_k by magnitude, once more per degree for the normalised direction they are built from */
                    vs2[k * 3 + ch] = (1.0 + (double)u->sh_degree) * Ym[k] * fabs(a[5 + ch]);
                }
            mag_opac[g] = m[8] * (s * (1.0 - s));
            flip_opac[g] = fl[8] * (s * (1.0 - s));
            dep_opac[g] = dp[8] * (s * (1.0 - s));
            /* v_opac = v * s (1 - s) (gather_grads.wgsl:224-227): the f32 (1 - s) carries an absolute error eps s, i.e. a
             * relative one of eps s / (1 - s) that no summation magnitude sees when the sigmoid saturates */
            vjp_opac[g] = fabs(a[8]) * s;
            mag_xy[(size_t)g * 2] = m[0];
            mag_xy[(size_t)g * 2 + 1] = m[1];
            flip_xy[(size_t)g * 2] = fl[0];
            flip_xy[(size_t)g * 2 + 1] = fl[1];
            dep_xy[(size_t)g * 2] = dp[0];
            dep_xy[(size_t)g * 2 + 1] = dp[1];
            d_project_backward_mag(u, mean, log_scales + (size_t)g * 3, quats + (size_t)g * 4, a, a + 2,
                                   vjp_means + (size_t)g * 3, vjp_scales + (size_t)g * 3, vjp_quats + (size_t)g * 4);
            /* the projection VJP is linear in (v_xy, v_conic): its columns, one unit input at a time */
            for (int k = 0; k < 5; k++) {
                double in[5] = {0, 0, 0, 0, 0}, om[3], os[3], oq[4];
                in[k] = 1.0;
                d_project_backward_one(u, mean, log_scales + (size_t)g * 3, quats + (size_t)g * 4, in, in + 2, om, os, oq);
                for (int j = 0; j < 3; j++) mag_means[(size_t)g * 3 + j] += fabs(om[j]) * m[k];
                for (int j = 0; j < 3; j++) mag_scales[(size_t)g * 3 + j] += fabs(os[j]) * m[k];
                for (int j = 0; j < 4; j++) mag_quats[(size_t)g * 4 + j] += fabs(oq[j]) * m[k];
                for (int j = 0; j < 3; j++) flip_means[(size_t)g * 3 + j] += fabs(om[j]) * fl[k];
                for (int j = 0; j < 3; j++) flip_scales[(size_t)g * 3 + j] += fabs(os[j]) * fl[k];
                for (int j = 0; j < 4; j++) flip_quats[(size_t)g * 4 + j] += fabs(oq[j]) * fl[k];
                for (int j = 0; j < 3; j++) dep_means[(size_t)g * 3 + j] += fabs(om[j]) * dp[k];
                for (int j = 0; j < 3; j++) dep_scales[(size_t)g * 3 + j] += fabs(os[j]) * dp[k];
                for (int j = 0; j < 4; j++) dep_quats[(size_t)g * 4 + j] += fabs(oq[j]) * dp[k];
            }
        }
    }
    free(rows); free(acc); free(rows_abs); free(acc_abs); free(rows_flip); free(acc_flip); free(rows_dep); free(acc_dep);
    return 0;
}

/* Forward compositing in f64 (rasterize.wgsl:20-115) from the f32 ProjectedSplat records and the f32 tile lists, the
 * walk's decisions (sigma >= 0, alpha >= 1/255, the saturation stop) taken as the f32 restatement takes them.  out:
 * [h,w,4] f64.  Arbiter for the pixel tolerance: where the quadratic form cancels heavily (a splat tens of thousands of
 * pixels wide and far off-screen), an f32 evaluation of sigma with or without fused multiply-adds differs by 1e-7 of the
 * cancelling terms, i.e. by far more than 1e-7 of sigma, and the reference leaves the contraction to its shader compiler.
 * cond (optional, [h,w]): first-order bound of what a RELATIVE perturbation of every sigma's terms can move in the pixel,
 *   2 max(1, max_i |c_i|) sum_i mag_i alpha_i T_i,  mag_i = 0.5 (|a| dx^2 + |c| dy^2) + |b dx dy|
 * (d out / d sigma_i = -alpha_i T_i (c_i - colour behind i), |c_i - behind| <= 2 max |c|; alpha channel: <= alpha_i T_i). */
int oracle_rasterize_forward_f64(const OracleUniforms *u, const OracleAux *aux, double *out, double *cond) {
    const uint32_t w = u->img_size[0], h = u->img_size[1];
    const uint32_t tbx = u->tile_bounds[0], tby = u->tile_bounds[1];
#pragma omp parallel for schedule(dynamic, 1)
    for (int64_t tile = 0; tile < (int64_t)tbx * tby; tile++) {
        const uint32_t tile_x = (uint32_t)(tile % tbx), tile_y = (uint32_t)(tile / tbx);
        const uint32_t r0 = aux->tile_bins[tile * 2], r1 = aux->tile_bins[tile * 2 + 1];
        for (uint32_t ly = 0; ly < 16; ly++)
            for (uint32_t lx = 0; lx < 16; lx++) {
                const uint32_t px = tile_x * 16 + lx, py = tile_y * 16 + ly;
                if (px >= w || py >= h) continue;
                const float pcx = (float)px + 0.5f, pcy = (float)py + 0.5f;
                float T = 1.0f;
                double Td = 1.0, rgb[3] = {0, 0, 0}, sens = 0.0, cmax = 1.0;
                for (uint32_t i = r0; i < r1; i++) {
                    const float *p = aux->projected_splats + (size_t)aux->compact_gid_from_isect[i] * 9;
                    const float dx = p[0] - pcx, dy = p[1] - pcy;
                    const float sigma = 0.5f * (p[2] * dx * dx + p[4] * dy * dy) + p[3] * dx * dy;
                    const float alpha = fminf(0.999f, p[8] * expf(-sigma));
                    if (sigma >= 0.0f && alpha >= 1.0f / 255.0f) {
                        const float next_T = T * (1.0f - alpha);
                        if (next_T <= 1e-4f) break;
                        const double ddx = (double)p[0] - (double)pcx, ddy = (double)p[1] - (double)pcy;
                        const double sd = 0.5 * ((double)p[2] * ddx * ddx + (double)p[4] * ddy * ddy) + (double)p[3] * ddx * ddy;
                        const double ad = fmin(0.999, (double)p[8] * exp(-sd));
                        const double fac = ad * Td;
                        sens += fac * (0.5 * (fabs((double)p[2]) * ddx * ddx + fabs((double)p[4]) * ddy * ddy) + fabs((double)p[3] * ddx * ddy));
                        cmax = fmax(cmax, fmax(fabs((double)p[5]), fmax(fabs((double)p[6]), fabs((double)p[7]))));
                        rgb[0] += (double)p[5] * fac, rgb[1] += (double)p[6] * fac, rgb[2] += (double)p[7] * fac;
                        Td *= 1.0 - ad;
                        T = next_T;
                    }
                }
                double *o = out + ((size_t)px + (size_t)py * w) * 4;
                o[0] = rgb[0], o[1] = rgb[1], o[2] = rgb[2], o[3] = 1.0 - Td;
                if (cond) cond[(size_t)px + (size_t)py * w] = 2.0 * cmax * sens;
            }
    }
    return 0;
}
