#!/usr/bin/env python3
"""Evaluation-efficiency counters of the compositing kernels (GPU; reads the HIP forward's own aux state).

For one forward pass it reports, per launch:
  issued   — pixel evaluations the forward compositing kernel executes: every record of a tile's list up to the
             tile's early exit, times 256 pixels (bounds: whole lists / lists cut at the tile's last contributing
             record);
  passing  — evaluations with `sigma >= 0 and alpha >= 1/255` (rasterize.wgsl:80-87), i.e. the ones that can
             change a pixel;
  and how many 64-lane issue slots sub-tile skipping at 8x8 / 4x4 granularity would need (a slot = one pixel
  per lane of a wave64).
"""
from __future__ import annotations

import argparse
import json
import math
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--splats", type=int, default=1 << 20)
    ap.add_argument("--width", type=int, default=1920)
    ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--sh-degree", type=int, default=3)
    ap.add_argument("--mean-mult", type=float, default=1.0)
    ap.add_argument("--out", default=None)
    a = ap.parse_args()
    import brush_amd
    from brush_amd import render as R
    from brush_amd.synthetic import synthetic_cloud

    dev = torch.device("cuda:0")
    w, h = a.width, a.height
    c = {k: torch.as_tensor(v, device=dev) for k, v in synthetic_cloud(a.splats, a.sh_degree, mean_mult=a.mean_mult).items()}
    focal = brush_amd.fov_to_focal(math.pi * 0.5, w)
    cam = brush_amd.Camera([0.0, 0.0, -8.0], [0.0, 0.0, 0.0, 1.0], brush_amd.focal_to_fov(focal, w),
                           brush_amd.focal_to_fov(focal, h), (0.5, 0.5))
    out, aux, u = R._forward_impl(cam, (w, h), c["means"], c["log_scales"], c["quats"], c["sh"], c["raw_opac"], False,
                                  40_000_000)
    V, I = aux.read_num_visible(), aux.read_num_intersections()
    tbx = int(u.tile_bounds[0])
    bins = aux.tile_bins.long().reshape(-1, 2)
    T = bins.shape[0]
    # tile id of every record
    tile_of = torch.repeat_interleave(torch.arange(T, device=dev), (bins[:, 1] - bins[:, 0]).clamp_min(0))
    assert tile_of.numel() == I
    gid = aux.compact_gid_from_isect[:I].long()
    proj = aux.projected_splats
    fin = aux.final_index.long()
    # per-tile largest final index -> records the kernels certainly walk
    fin_pad = torch.zeros((((h + 15) // 16) * 16, ((w + 15) // 16) * 16), dtype=torch.long, device=dev)
    fin_pad[:h, :w] = fin
    tile_max_fin = fin_pad.reshape(fin_pad.shape[0] // 16, 16, fin_pad.shape[1] // 16, 16).amax(dim=(1, 3)).reshape(-1)
    lists = (bins[:, 1] - bins[:, 0]).clamp_min(0)
    walked_lo = torch.where(lists > 0, (tile_max_fin + 1 - bins[:, 0]).clamp(min=0), torch.zeros_like(lists))
    walked_lo = torch.minimum(walked_lo, lists)
    tot_pass = 0
    slots8 = 0
    slots4 = 0
    zero = 0
    chunk = 1 << 16
    ar = torch.arange(16, device=dev, dtype=torch.float32) + 0.5
    for s in range(0, I, chunk):
        t = tile_of[s:s + chunk]
        p = proj[gid[s:s + chunk]]
        ox = ((t % tbx) * 16).float()
        oy = ((t // tbx) * 16).float()
        px = ox[:, None, None] + ar[None, None, :]
        py = oy[:, None, None] + ar[None, :, None]
        dx = p[:, 0, None, None] - px
        dy = p[:, 1, None, None] - py
        sigma = 0.5 * (p[:, 2, None, None] * dx * dx + p[:, 4, None, None] * dy * dy) + p[:, 3, None, None] * dx * dy
        alpha = torch.clamp(p[:, 8, None, None] * torch.exp(-sigma), max=0.999)
        m = (sigma >= 0) & (alpha >= 1.0 / 255.0) & (px < w) & (py < h)
        n = m.shape[0]
        cnt = m.reshape(n, -1).sum(1)
        tot_pass += int(cnt.sum())
        zero += int((cnt == 0).sum())
        slots8 += int(m.reshape(n, 2, 8, 2, 8).any(dim=4).any(dim=2).sum())
        slots4 += int(m.reshape(n, 4, 4, 4, 4).any(dim=4).any(dim=2).sum())
    res = {
        "workload": f"{a.splats} splats @{w}x{h}, SH {a.sh_degree}, mean_mult {a.mean_mult}",
        "num_visible": V, "num_intersections": I, "tiles": T,
        "issued_pixel_evals_upper": int(lists.sum()) * 256, "issued_pixel_evals_lower": int(walked_lo.sum()) * 256,
        "passing_pixel_evals": tot_pass,
        "passing_fraction_of_upper": round(tot_pass / max(1, int(lists.sum()) * 256), 4),
        "records_with_no_passing_pixel": zero,
        "slots_per_record_now": 4.0,
        "slots_per_record_8x8_skipping": round(slots8 / max(1, I), 3),
        "slots_per_record_4x4_skipping": round(slots4 / 4 / max(1, I), 3),
    }
    print(json.dumps(res))
    if a.out:
        os.makedirs(os.path.dirname(os.path.abspath(a.out)), exist_ok=True)
        json.dump(res, open(a.out, "w"), indent=1)


if __name__ == "__main__":
    main()
