"""Footprint statistics of the compositing kernels' work (CPU, numpy; uses the oracle as the source of the
forward state, so this is an analysis tool, not product code).

For every (tile, splat) intersection record of a synthetic scene: how many of the tile's 256 pixels pass
`sigma >= 0 and alpha >= 1/255` (rasterize.wgsl:80-87), and how many 64-lane issue slots different
sub-tile granularities would need.
"""
import argparse
import math
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from oracle import oracle as O
from brush_amd.synthetic import synthetic_cloud


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--splats", type=int, default=1 << 20)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--mean-mult", type=float, default=1.0)
    ap.add_argument("--sample", type=int, default=200000)
    a = ap.parse_args()
    w, h = a.width, a.height
    c = synthetic_cloud(a.splats, 0, mean_mult=a.mean_mult)
    fov = math.pi * 0.5
    focal = O.fov_to_focal(fov, w)
    u = O.make_uniforms([0, 0, -8.0], [0, 0, 0, 1.0], O.focal_to_fov(focal, w), O.focal_to_fov(focal, h), [0.5, 0.5], [w, h], 0)
    out, aux = O.render_forward(u, c["means"], c["log_scales"], c["quats"], c["sh"], c["raw_opac"], max_intersects=40_000_000)
    V, I = int(aux["num_visible"][0]), int(aux["num_intersections"][0])
    print("V", V, "I", I, "I/V", I / V)
    tid = aux["tile_id_from_isect"][:I]  # sorted
    gid = aux["compact_gid_from_isect"][:I]
    proj = aux["projected_splats"]
    tbx = int(u["tile_bounds"][0])
    rng = np.random.default_rng(0)
    idx = rng.choice(I, size=min(a.sample, I), replace=False)
    t = tid[idx].astype(np.int64); g = gid[idx]
    p = proj[g]
    ox = (t % tbx) * 16; oy = (t // tbx) * 16
    px = ox[:, None, None] + np.arange(16)[None, None, :] + 0.5
    py = oy[:, None, None] + np.arange(16)[None, :, None] + 0.5
    dx = p[:, 0, None, None] - px; dy = p[:, 1, None, None] - py
    sigma = 0.5 * (p[:, 2, None, None] * dx * dx + p[:, 4, None, None] * dy * dy) + p[:, 3, None, None] * dx * dy
    alpha = np.minimum(0.999, p[:, 8, None, None] * np.exp(-sigma))
    m = (sigma >= 0) & (alpha >= 1.0 / 255.0) & (px < w) & (py < h)  # [S,16(y),16(x)]
    S = len(idx)
    npx = m.reshape(S, -1).sum(1)
    print(f"pixels passing per record: mean {npx.mean():.1f} median {np.median(npx):.0f}  zero-hit records {np.mean(npx == 0) * 100:.1f}%  frac of 256: {npx.mean() / 256:.3f}")
    def slots(bh, bw):
        mm = m.reshape(S, 16 // bh, bh, 16 // bw, bw).any(axis=(2, 4))
        nb = mm.reshape(S, -1).sum(1)
        lanes = bh * bw
        return nb, nb * lanes / 64.0
    for bh, bw in [(16, 16), (8, 8), (4, 16), (16, 4), (4, 4), (2, 8), (8, 2), (2, 2), (1, 16), (1, 4), (1, 1)]:
        nb, s64 = slots(bh, bw)
        print(f"block {bh}x{bw}: blocks hit mean {nb.mean():.2f}; 64-lane slot-equivalents mean {s64.mean():.2f} (vs 4.0 now); lane efficiency {npx.mean() / max(1e-9, (s64.mean() * 64)):.3f}")
    # rows touched, row-span widths
    rows = m.any(axis=2).sum(1)
    cols = m.any(axis=1).sum(1)
    print(f"rows touched mean {rows.mean():.2f} cols touched mean {cols.mean():.2f}")
    hist = np.bincount(np.minimum(npx, 256) // 16, minlength=17)
    print("hist of passing pixels /16:", (hist / S).round(3).tolist())


if __name__ == "__main__":
    main()
