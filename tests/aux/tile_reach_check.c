/* CPU check of the two-phase tile test (brush_amd/csrc/splat_math.hpp: tile_test_head / make_tile_reach):
 * the conservative kTileMiss exit must never fire where the exact test (oracle: ellipse_intersects_aabb) says "hit".
 * Includes the oracle's translation unit to reach its static functions; test infrastructure only.
 *   gcc -O2 -ffp-contract=off -fopenmp -I oracle tests/aux/tile_reach_check.c -lm -o /tmp/tile_reach_check */
#include "../../oracle/brush_oracle.c"
#include <stdio.h>

static uint64_t rng_state = 0x9E3779B97F4A7C15ull;
static inline uint64_t next_u64(uint64_t *s) { uint64_t z = (*s += 0x9E3779B97F4A7C15ull); z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull; z = (z ^ (z >> 27)) * 0x94D049BB133111EBull; return z ^ (z >> 31); }
static inline float uni(uint64_t *s) { return (float)((next_u64(s) >> 40) * (1.0 / 16777216.0)); }

int main(int argc, char **argv) {
    long cases = argc > 1 ? atol(argv[1]) : 20000000L;
    long bad = 0, miss_exit = 0, hit_exit = 0, edge = 0, exact_hits = 0;
#pragma omp parallel for reduction(+ : bad, miss_exit, hit_exit, edge, exact_hits)
    for (long i = 0; i < cases; i++) {
        uint64_t s = rng_state + (uint64_t)i * 0xD1B54A32D192ED03ull;
        /* a random ellipse: axes 0.3 .. 3000 px, any orientation, opacity-scaled like make_tile_test */
        float l1 = expf(-1.2f + 9.2f * uni(&s)), l2 = expf(-1.2f + 9.2f * uni(&s)), th = 6.2831853f * uni(&s);
        float c = cosf(th), sn = sinf(th);
        float a = c * c / (l1 * l1) + sn * sn / (l2 * l2), b = c * sn * (1.0f / (l1 * l1) - 1.0f / (l2 * l2)),
              d = sn * sn / (l1 * l1) + c * c / (l2 * l2);
        float q[3] = {a, b, d};
        float xy[2] = {-200.0f + 4400.0f * uni(&s), -200.0f + 2600.0f * uni(&s)};
        /* a tile somewhere around the boundary of the ellipse: centre + direction * (0.3 .. 1.8) * extent */
        float ang = 6.2831853f * uni(&s), rr = (0.3f + 1.5f * uni(&s));
        float ex = sqrtf(d / (a * d - b * b)), ey = sqrtf(a / (a * d - b * b));
        float px = xy[0] + cosf(ang) * rr * (ex + 12.0f), py = xy[1] + sinf(ang) * rr * (ey + 12.0f);
        if (px < 0 || py < 0 || px > 65535.0f * 16 || py > 65535.0f * 16) continue;
        uint32_t tx = (uint32_t)(px / 16.0f), ty = (uint32_t)(py / 16.0f);
        float ext[2] = {8.0f, 8.0f}, tc[2] = {(float)(tx * 16) + 8.0f, (float)(ty * 16) + 8.0f};
        int exact = ellipse_intersects_aabb(tc, ext, xy, q);
        exact_hits += exact;
        /* head, as in splat_math.hpp */
        float dq = q[0] * q[2] - q[1] * q[1];
        float hx = sqrtf(q[2] / dq), hy = sqrtf(q[0] / dq);
        int ok = dq > 0.0f && q[0] > 0.0f && q[2] > 0.0f && dq * 1024.0f >= q[0] * q[2] && hx < 3.0e37f && hy < 3.0e37f;
        float rx = ok ? hx * 1.001f + 8.02f : INFINITY, ry = ok ? hy * 1.001f + 8.02f : INFINITY;
        float dd[2] = {xy[0] - tc[0], xy[1] - tc[1]};
        int cls;
        if (fabsf(dd[0]) <= 8.0f && fabsf(dd[1]) <= 8.0f) cls = 1;
        else {
            float sg[2] = {signf(dd[0]), signf(dd[1])};
            float nc[2] = {tc[0] + sg[0] * 8.0f, tc[1] + sg[1] * 8.0f};
            float cp[2] = {nc[0] - xy[0], nc[1] - xy[1]}, cq[2];
            vq(cp, q, cq);
            if (dot2(cq, cp) <= 1.0f) cls = 1;
            else if (fabsf(dd[0]) > rx || fabsf(dd[1]) > ry) cls = 0;
            else cls = 2;
        }
        /* walk_rect (splat_math.hpp): a tile the exact test accepts must lie inside the rectangle the walks enumerate */
        if (exact && ok) {
            float cxy[2] = {xy[0], xy[1]}, rr[2] = {rx, ry};
            uint32_t tt[2] = {tx, ty};
            for (int k = 0; k < 2; k++) {
                long lo = (long)floorf((cxy[k] - rr[k] - 8.0f) / 16.0f) - 1;
                long hi = (long)floorf((cxy[k] + rr[k] - 8.0f) / 16.0f) + 2;
                if ((long)tt[k] < lo || (long)tt[k] >= hi) bad++;
            }
        }
        if (cls == 0) { miss_exit++; if (exact) bad++; }
        if (cls == 1) { hit_exit++; if (!exact) bad++; }
        if (cls == 2) edge++;
    }
    printf("cases %ld exact_hits %ld hit_exit %ld miss_exit %ld edge %ld BAD %ld\n", cases, exact_hits, hit_exit, miss_exit, edge, bad);
    return bad != 0;
}
