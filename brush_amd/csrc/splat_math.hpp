// splat_math.hpp — per-splat projection math shared by the forward and backward kernels.
//
// Follows crates/brush-render/src/shaders/helpers.wgsl (file:line cited per function).
// Matrices are row-major m[row][col]; WGSL is column-major, swaps are noted where they matter.
// Every expression keeps the written association; translation units that include this file
// are compiled with -ffp-contract=off so that each a*b+c is two roundings (bit-for-bit
// comparable with a scalar CPU evaluation of the same formulas).
#pragma once
#include "common.hpp"
#include "detmath.hpp"

namespace brush {

constexpr float kCovBlur = 0.3f;  // helpers.wgsl:166

struct Mat3 {
    float m[3][3];
};

__device__ __forceinline__ Mat3 mul(const Mat3 &a, const Mat3 &b) {
    Mat3 c;
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 3; j++)
            c.m[i][j] = a.m[i][0] * b.m[0][j] + a.m[i][1] * b.m[1][j] + a.m[i][2] * b.m[2][j];
    return c;
}
__device__ __forceinline__ Mat3 transpose(const Mat3 &a) {
    Mat3 c;
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 3; j++) c.m[i][j] = a.m[j][i];
    return c;
}
__device__ __forceinline__ Mat3 add(const Mat3 &a, const Mat3 &b) {
    Mat3 c;
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 3; j++) c.m[i][j] = a.m[i][j] + b.m[i][j];
    return c;
}

// Uniform values every per-splat kernel needs, passed by value as a kernel argument.
struct ViewParams {
    float vm[16];  // world->camera, column-major (helpers.wgsl:9)
    float focal[2];
    float pixel_center[2];
    uint32_t img_size[2];
    uint32_t tile_bounds[2];
    uint32_t sh_degree;
    uint32_t total_splats;
    float cull_k;  // upper bound of (z * |J W|_F)^2, see make_view_params
};

inline ViewParams make_view_params(const BrushUniforms &u, uint32_t n) {
    ViewParams v;
    for (int i = 0; i < 16; i++) v.vm[i] = u.viewmat[i];
    for (int i = 0; i < 2; i++) {
        v.focal[i] = u.focal[i];
        v.pixel_center[i] = u.pixel_center[i];
        v.img_size[i] = u.img_size[i];
        v.tile_bounds[i] = u.tile_bounds[i];
    }
    v.sh_degree = u.sh_degree;
    v.total_splats = n;
    // Conservative cull constant.  cov = (J W) V (J W)^T with |V|_2 = s_max^2, so
    // lambda_max(cov) <= s_max^2 |J|_F^2 |W|_F^2, and with t clamped to z*lims (helpers.wgsl:127-134)
    // |J|_F^2 <= (fx^2 (1 + lx^2) + fy^2 (1 + ly^2)) / z^2.  1 % head-room for rounding.
    double wf2 = 0.0;
    for (int c = 0; c < 3; c++)
        for (int r = 0; r < 3; r++) wf2 += (double)u.viewmat[c * 4 + r] * u.viewmat[c * 4 + r];
    double k = 0.0;
    for (int i = 0; i < 2; i++) {
        const double f = u.focal[i], img = u.img_size[i], pc = u.pixel_center[i];
        const double tan_fov = 0.5 * img / f;
        const double lp = (img - pc) / f + 0.3 * tan_fov, ln = pc / f + 0.3 * tan_fov;
        const double l = lp > ln ? lp : ln;
        k += f * f * (1.0 + l * l);
    }
    v.cull_k = (float)(k * wf2 * 1.01);
    return v;
}

// W = mat3x3f(viewmat[0].xyz, viewmat[1].xyz, viewmat[2].xyz)
__device__ __forceinline__ Mat3 view_rot(const ViewParams &vp) {
    Mat3 w;
#pragma unroll
    for (int r = 0; r < 3; r++)
#pragma unroll
        for (int c = 0; c < 3; c++) w.m[r][c] = vp.vm[c * 4 + r];
    return w;
}

// p_view = W * mean + viewmat[3].xyz   (project_forward.wgsl:29-30)
__device__ __forceinline__ void to_view(const ViewParams &vp, const float mean[3], float p[3]) {
#pragma unroll
    for (int r = 0; r < 3; r++)
        p[r] = (vp.vm[0 * 4 + r] * mean[0] + vp.vm[1 * 4 + r] * mean[1] + vp.vm[2 * 4 + r] * mean[2]) +
               vp.vm[12 + r];
}

// helpers.wgsl:74-109, quat = (w,x,y,z)
__device__ __forceinline__ Mat3 quat_to_rotmat(const float q[4]) {
    const float w = q[0], x = q[1], y = q[2], z = q[3];
    const float x2 = x * x, y2 = y * y, z2 = z * z;
    const float xy = x * y, xz = x * z, yz = y * z;
    const float wx = w * x, wy = w * y, wz = w * z;
    Mat3 r;
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

// helpers.wgsl:119-122
__device__ __forceinline__ void project_pix(const ViewParams &vp, const float p[3], float xy[2]) {
    xy[0] = (p[0] / p[2]) * vp.focal[0] + vp.pixel_center[0];
    xy[1] = (p[1] / p[2]) * vp.focal[1] + vp.pixel_center[1];
}

__device__ __forceinline__ float clampf(float x, float lo, float hi) {
    return fminf(fmaxf(x, lo), hi);
}

// helpers.wgsl:124-158 -> (c00, c01, c11)
__device__ __forceinline__ void calc_cov2d(const ViewParams &vp, const float p_view[3],
                                           const float scale[3], const float quat[4],
                                           float cov2d[3]) {
    const float img[2] = {(float)vp.img_size[0], (float)vp.img_size[1]};
    float t[2];
    const float rz = 1.0f / p_view[2];
    const float rz2 = rz * rz;
#pragma unroll
    for (int i = 0; i < 2; i++) {
        const float tan_fov = 0.5f * img[i] / vp.focal[i];
        const float lim_pos = (img[i] - vp.pixel_center[i]) / vp.focal[i] + 0.3f * tan_fov;
        const float lim_neg = vp.pixel_center[i] / vp.focal[i] + 0.3f * tan_fov;
        t[i] = p_view[2] * clampf(p_view[i] * rz, -lim_neg, lim_pos);
    }
    Mat3 M = quat_to_rotmat(quat);
#pragma unroll
    for (int r = 0; r < 3; r++)
#pragma unroll
        for (int c = 0; c < 3; c++) M.m[r][c] = M.m[r][c] * scale[c];

    const float j00 = vp.focal[0] * rz, j11 = vp.focal[1] * rz;
    const float j02 = (-vp.focal[0]) * t[0] * rz2, j12 = (-vp.focal[1]) * t[1] * rz2;
    const Mat3 W = view_rot(vp);
    const Mat3 V = mul(M, transpose(M));
    float T[2][3], TV[2][3];
#pragma unroll
    for (int c = 0; c < 3; c++) {
        T[0][c] = (j00 * W.m[0][c] + 0.0f * W.m[1][c]) + j02 * W.m[2][c];
        T[1][c] = (0.0f * W.m[0][c] + j11 * W.m[1][c]) + j12 * W.m[2][c];
    }
#pragma unroll
    for (int r = 0; r < 2; r++)
#pragma unroll
        for (int c = 0; c < 3; c++)
            TV[r][c] = T[r][0] * V.m[0][c] + T[r][1] * V.m[1][c] + T[r][2] * V.m[2][c];
    const float cov00 = TV[0][0] * T[0][0] + TV[0][1] * T[0][1] + TV[0][2] * T[0][2];
    const float cov10 = TV[1][0] * T[0][0] + TV[1][1] * T[0][1] + TV[1][2] * T[0][2];
    const float cov11 = TV[1][0] * T[1][0] + TV[1][1] * T[1][1] + TV[1][2] * T[1][2];
    cov2d[0] = cov00 + kCovBlur;
    cov2d[1] = cov10;  // WGSL cov[0][1] = column 0, row 1
    cov2d[2] = cov11 + kCovBlur;
}

// helpers.wgsl:160-164
__device__ __forceinline__ void cov_to_conic(const float c[3], float conic[3]) {
    const float det = c[0] * c[2] - c[1] * c[1];
    const float inv_det = 1.0f / det;
    conic[0] = c[2] * inv_det;
    conic[1] = (-c[1]) * inv_det;
    conic[2] = c[0] * inv_det;
}

// WGSL u32(f32) / i32(f32): truncate, saturate, NaN -> 0.
__device__ __forceinline__ uint32_t f2u_sat(float x) {
    if (!(x > 0.0f)) return 0u;
    if (x >= 4294967296.0f) return 0xFFFFFFFFu;
    return (uint32_t)x;
}
__device__ __forceinline__ int32_t f2i_sat(float x) {
    if (x != x) return 0;
    if (x >= 2147483648.0f) return 2147483647;
    if (x <= -2147483648.0f) return (int32_t)0x80000000;
    return (int32_t)x;
}
__device__ __forceinline__ int32_t iclamp(int32_t v, int32_t lo, int32_t hi) {
    return v < lo ? lo : (v > hi ? hi : v);
}

// helpers.wgsl:192-201 (opacity ignored by the reference)
__device__ __forceinline__ uint32_t radius_from_conic(const float conic[3]) {
    const float det = 1.0f / (conic[0] * conic[2] - conic[1] * conic[1]);
    const float cx = conic[2] * det, cz = conic[0] * det;
    const float b = 0.5f * (cx + cz);
    const float sq = sqrtf(fmaxf(0.1f, b * b - det));
    const float v1 = b + sq, v2 = b - sq;
    const float radius = 3.0f * sqrtf(fmaxf(0.0f, fmaxf(v1, v2)));
    return f2u_sat(ceilf(radius));
}

// helpers.wgsl:55-71 -> (min.x, min.y, max.x, max.y), max exclusive
__device__ __forceinline__ void get_tile_bbox(const float xy[2], uint32_t radius,
                                              const uint32_t bounds[2], uint32_t bb[4]) {
    const float tr = (float)radius / (float)kTileWidth;
#pragma unroll
    for (int i = 0; i < 2; i++) {
        const float tc = xy[i] / (float)kTileWidth;
        bb[i] = (uint32_t)iclamp(f2i_sat(tc - tr), 0, (int32_t)bounds[i]);
        bb[2 + i] = (uint32_t)iclamp(f2i_sat((tc + tr) + 1.0f), 0, (int32_t)bounds[i]);
    }
}

__device__ __forceinline__ float signf(float x) { return x > 0.0f ? 1.0f : (x < 0.0f ? -1.0f : 0.0f); }
__device__ __forceinline__ float dot2(const float a[2], const float b[2]) { return a[0] * b[0] + a[1] * b[1]; }
// v * Q, Q = mat2x2f(q0,q1,q1,q2)
__device__ __forceinline__ void vq(const float v[2], const float q[3], float o[2]) {
    o[0] = v[0] * q[0] + v[1] * q[1];
    o[1] = v[0] * q[1] + v[1] * q[2];
}

// helpers.wgsl:220-236
__device__ __forceinline__ bool check_edge(const float p1[2], const float p2[2],
                                           const float center[2], const float q[3]) {
    const float edge[2] = {p2[0] - p1[0], p2[1] - p1[1]};
    const float f[2] = {p1[0] - center[0], p1[1] - center[1]};
    float eq[2], fq[2];
    vq(edge, q, eq);
    vq(f, q, fq);
    const float a = dot2(eq, edge);
    const float b = 2.0f * dot2(fq, edge);
    const float c = dot2(fq, f) - 1.0f;
    const float disc = b * b - 4.0f * a * c;
    if (disc < 0.0f) return false;
    const float sd = sqrtf(disc);
    const float t1 = (-b - sd) / (2.0f * a);
    const float t2 = (-b + sd) / (2.0f * a);
    return (t1 >= 0.0f && t1 <= 1.0f) || (t2 >= 0.0f && t2 <= 1.0f);
}

// helpers.wgsl:238-262
__device__ __forceinline__ bool ellipse_intersects_aabb(const float box_pos[2], const float ext[2],
                                                        const float center[2], const float q[3]) {
    const float d[2] = {center[0] - box_pos[0], center[1] - box_pos[1]};
    if (fabsf(d[0]) <= ext[0] && fabsf(d[1]) <= ext[1]) return true;
    const float sg[2] = {signf(d[0]), signf(d[1])};
    const float nc[2] = {box_pos[0] + sg[0] * ext[0], box_pos[1] + sg[1] * ext[1]};
    const float cp[2] = {nc[0] - center[0], nc[1] - center[1]};
    float cq[2];
    vq(cp, q, cq);
    if (dot2(cq, cp) <= 1.0f) return true;
    const float e1[2] = {nc[0] - sg[0] * 2.0f * ext[0], nc[1] - 0.0f};
    const float e2[2] = {nc[0] - 0.0f, nc[1] - sg[1] * 2.0f * ext[1]};
    return check_edge(nc, e1, center, q) || check_edge(nc, e2, center, q);
}

// helpers.wgsl:264-279, split so callers evaluate log() and the conic scaling once per splat.
struct TileTest {
    float q[3];
    bool any;
};
__device__ __forceinline__ TileTest make_tile_test(const float conic[3], float opac) {
    TileTest t;
    const float sigma = det_logf(opac * 255.0f);
    t.any = sigma > 0.0f;
    const float den = 2.0f * sigma;
    t.q[0] = conic[0] / den;
    t.q[1] = conic[1] / den;
    t.q[2] = conic[2] / den;
    return t;
}
__device__ __forceinline__ bool can_be_visible(const TileTest &t, uint32_t tx, uint32_t ty,
                                               const float xy[2]) {
    if (!t.any) return false;
    const float ext[2] = {(float)kTileWidth / 2.0f, (float)kTileWidth / 2.0f};
    const float tc[2] = {(float)(tx * kTileWidth) + ext[0], (float)(ty * kTileWidth) + ext[1]};
    return ellipse_intersects_aabb(tc, ext, xy, t.q);
}

// ---- the same test in two phases (tile walks) ----------------------------------------------------------------------
// ellipse_intersects_aabb is ~25 instructions for its first two clauses (centre inside the box; nearest corner inside
// the ellipse) and ~250 for the two check_edge() calls behind them (two square roots, four divisions).  Over the bbox
// tiles of the bench scenes 50-57 % of the tests end in the first two clauses and another 23-32 % are tiles the ellipse
// cannot reach at all, so the walks classify every candidate with the HEAD below and queue the ~18 % left for the TAIL,
// which then runs on full waves.  Decisions are bit-identical to can_be_visible():
//   * head: the function's own first two clauses, same expressions, same order -> kTileHit;
//   * kTileMiss only when the tile box lies outside the axis-aligned bounding box of the ellipse
//     d^T Q d <= 1 (half extents sqrt(q2 / det Q), sqrt(q0 / det Q)) by a margin of 0.1 % + 0.02 px: every point of
//     both edge segments is then outside the ellipse by far more than the f32 rounding of check_edge's roots, so
//     both calls return false.  Non-finite or non-positive-definite Q never takes this exit (rx = ry = +inf);
//   * tail: exactly the two check_edge() calls.
enum : uint32_t { kTileMiss = 0u, kTileHit = 1u, kTileEdge = 2u };
struct TileReach {
    float rx, ry;  // |centre - tile centre| beyond which the tile cannot be reached
};
__device__ __forceinline__ TileReach make_tile_reach(const TileTest &t) {
    TileReach r;
    const float dq = t.q[0] * t.q[2] - t.q[1] * t.q[1];
    const float hx = sqrtf(t.q[2] / dq), hy = sqrtf(t.q[0] / dq);
    const float half = (float)kTileWidth / 2.0f;
    // det Q = q0 q2 - q1^2 cancels for elongated, tilted ellipses: its f32 error is ~eps q0 q2, so the extents are only
    // trusted (to 1e-4, against a margin of 1e-3) while det Q >= q0 q2 / 1024; beyond that every tile takes the tail
    // (found by tests/aux/tile_reach_check.c: 432 wrong exits in 2e8 random cases without this condition, 0 with it)
    const bool ok = t.any && dq > 0.0f && t.q[0] > 0.0f && t.q[2] > 0.0f && dq * 1024.0f >= t.q[0] * t.q[2] &&
                    hx < 3.0e37f && hy < 3.0e37f;  // NaN -> false
    r.rx = ok ? hx * 1.001f + (half + 0.02f) : __builtin_inff();
    r.ry = ok ? hy * 1.001f + (half + 0.02f) : __builtin_inff();
    return r;
}
// The rectangle of tiles a walk has to visit: the reference's bbox (get_tile_bbox: radius 3 sigma of the LARGER axis,
// opacity-blind) cut down to the tiles the opacity-aware ellipse of the exact test can reach at all
// (the kTileMiss exit above, applied to whole rows and columns at once; one tile of slack on either side for the
// rounding of the bounds).  Tiles outside it are misses of can_be_visible(), so the hits and their row-major ORDER are
// unchanged; what changes is how many candidates are enumerated (measured over the bench scenes: 23-32 % fewer) and how
// many 64-tile chunks a large splat is cut into.  Every kernel that walks or replays a splat's tiles derives the
// rectangle with this one function from the same ProjectedSplat bits, so they all agree on it.
__device__ __forceinline__ void walk_rect(const float xy[2], const float conic[3], const TileTest &t,
                                          const TileReach &r, const uint32_t bounds[2], uint32_t bb[4]) {
    get_tile_bbox(xy, radius_from_conic(conic), bounds, bb);
    if (!t.any) {  // log(255 opac) <= 0: no tile can be visible (helpers.wgsl:264-279)
        bb[2] = bb[0], bb[3] = bb[1];
        return;
    }
    const float rr[2] = {r.rx, r.ry};
    const float half = (float)kTileWidth / 2.0f;
#pragma unroll
    for (int i = 0; i < 2; i++) {
        if (!(rr[i] < 3.0e37f)) continue;  // reach unknown: the whole bbox
        // tile t can be reached only if |xy - (16 t + 8)| <= r  <=>  t in [(xy - r - 8) / 16, (xy + r - 8) / 16]
        const int32_t lo = f2i_sat(floorf((xy[i] - rr[i] - half) / (float)kTileWidth)) - 1;
        const int32_t hi = f2i_sat(floorf((xy[i] + rr[i] - half) / (float)kTileWidth)) + 2;  // exclusive
        const uint32_t nlo = (uint32_t)iclamp(lo, (int32_t)bb[i], (int32_t)bb[2 + i]);
        const uint32_t nhi = (uint32_t)iclamp(hi, (int32_t)bb[i], (int32_t)bb[2 + i]);
        bb[i] = nlo;
        bb[2 + i] = nhi > nlo ? nhi : nlo;
    }
}

__device__ __forceinline__ uint32_t tile_test_head(const TileTest &t, const TileReach &r, uint32_t tx, uint32_t ty,
                                                   const float center[2]) {
    if (!t.any) return kTileMiss;
    const float ext[2] = {(float)kTileWidth / 2.0f, (float)kTileWidth / 2.0f};
    const float box_pos[2] = {(float)(tx * kTileWidth) + ext[0], (float)(ty * kTileWidth) + ext[1]};
    const float d[2] = {center[0] - box_pos[0], center[1] - box_pos[1]};
    if (fabsf(d[0]) <= ext[0] && fabsf(d[1]) <= ext[1]) return kTileHit;
    const float sg[2] = {signf(d[0]), signf(d[1])};
    const float nc[2] = {box_pos[0] + sg[0] * ext[0], box_pos[1] + sg[1] * ext[1]};
    const float cp[2] = {nc[0] - center[0], nc[1] - center[1]};
    float cq[2];
    vq(cp, t.q, cq);
    if (dot2(cq, cp) <= 1.0f) return kTileHit;
    if (fabsf(d[0]) > r.rx || fabsf(d[1]) > r.ry) return kTileMiss;
    return kTileEdge;
}
// The tail of ellipse_intersects_aabb for a tile whose head returned kTileEdge (its first two clauses are false).
__device__ __forceinline__ bool tile_test_tail(const float q[3], uint32_t tx, uint32_t ty, const float center[2]) {
    const float ext[2] = {(float)kTileWidth / 2.0f, (float)kTileWidth / 2.0f};
    const float box_pos[2] = {(float)(tx * kTileWidth) + ext[0], (float)(ty * kTileWidth) + ext[1]};
    const float d[2] = {center[0] - box_pos[0], center[1] - box_pos[1]};
    const float sg[2] = {signf(d[0]), signf(d[1])};
    const float nc[2] = {box_pos[0] + sg[0] * ext[0], box_pos[1] + sg[1] * ext[1]};
    const float e1[2] = {nc[0] - sg[0] * 2.0f * ext[0], nc[1] - 0.0f};
    const float e2[2] = {nc[0] - 0.0f, nc[1] - sg[1] * 2.0f * ext[1]};
    return check_edge(nc, e1, center, q) || check_edge(nc, e2, center, q);
}

// Sloan SH basis, project_visible.wgsl:51-147 / gather_grads.wgsl:17-112.
template <int MAXC>
__device__ __forceinline__ void sh_basis(uint32_t degree, const float d[3], float Y[MAXC]) {
    Y[0] = 0.2820947917738781f;
    if (degree == 0) return;
    const float x = d[0], y = d[1], z = d[2];
    const float fTmp0A = 0.48860251190292f;
    Y[1] = -fTmp0A * y;
    Y[2] = fTmp0A * z;
    Y[3] = -fTmp0A * x;
    if (degree == 1) return;
    const float z2 = z * z;
    const float fTmp0B = -1.092548430592079f * z;
    const float fTmp1A = 0.5462742152960395f;
    const float fC1 = x * x - y * y;
    const float fS1 = 2.0f * x * y;
    const float pSH6 = 0.9461746957575601f * z2 - 0.3153915652525201f;
    Y[4] = fTmp1A * fS1;
    Y[5] = fTmp0B * y;
    Y[6] = pSH6;
    Y[7] = fTmp0B * x;
    Y[8] = fTmp1A * fC1;
    if (degree == 2) return;
    const float fTmp0C = -2.285228997322329f * z2 + 0.4570457994644658f;
    const float fTmp1B = 1.445305721320277f * z;
    const float fTmp2A = -0.5900435899266435f;
    const float fC2 = x * fC1 - y * fS1;
    const float fS2 = x * fS1 + y * fC1;
    const float pSH12 = z * (1.865881662950577f * z2 - 1.119528997770346f);
    Y[9] = fTmp2A * fS2;
    Y[10] = fTmp1B * fS1;
    Y[11] = fTmp0C * y;
    Y[12] = pSH12;
    Y[13] = fTmp0C * x;
    Y[14] = fTmp1B * fC1;
    Y[15] = fTmp2A * fC2;
    if (degree == 3) return;
    const float fTmp0D = z * (-4.683325804901025f * z2 + 2.007139630671868f);
    const float fTmp1C = 3.31161143515146f * z2 - 0.47308734787878f;
    const float fTmp2B = -1.770130769779931f * z;
    const float fTmp3A = 0.6258357354491763f;
    const float fC3 = x * fC2 - y * fS2;
    const float fS3 = x * fS2 + y * fC2;
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

// Quirk: the "camera position" is viewmat[3].xyz (project_visible.wgsl:232-233).
__device__ __forceinline__ void view_dir(const ViewParams &vp, const float mean[3], float dir[3]) {
    dir[0] = mean[0] - vp.vm[12];
    dir[1] = mean[1] - vp.vm[13];
    dir[2] = mean[2] - vp.vm[14];
    const float len = sqrtf(dir[0] * dir[0] + dir[1] * dir[1] + dir[2] * dir[2]);
    dir[0] = dir[0] / len;
    dir[1] = dir[1] / len;
    dir[2] = dir[2] / len;
}

}  // namespace brush
