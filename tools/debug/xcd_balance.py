"""Development aid: how evenly do the compositing kernels' XCD bands split the tile lists of a workload?"""
import math, sys, os, json
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import brush_amd
from brush_amd import render as R
from brush_amd.synthetic import synthetic_cloud

CFG = {"S1": (1 << 20, 1920, 1080, 3, 1.0, None), "dense": (1 << 20, 1920, 1080, 3, 0.25, None),
       "c3": (3_000_000, 1920, 1080, 3, 0.12, 40_000_000)}
name = sys.argv[1] if len(sys.argv) > 1 else "S1"
n, w, h, deg, mm, cap = CFG[name]
dev = torch.device("cuda:0")
p = {k: torch.as_tensor(v, device=dev) for k, v in synthetic_cloud(n, deg, seed=4, mean_mult=mm).items()}
focal = brush_amd.fov_to_focal(math.pi * 0.5, w)
cam = brush_amd.Camera([0.0, 0.0, -8.0], [0.0, 0.0, 0.0, 1.0], brush_amd.focal_to_fov(focal, w), brush_amd.focal_to_fov(focal, h), (0.5, 0.5))
out, aux, u = R._forward_impl(cam, (w, h), p["means"], p["log_scales"], p["quats"], p["sh"], p["raw_opac"], False, cap)
torch.cuda.synchronize()
tb = aux.tile_bins.view(-1, 2).long().cpu()
cnt = (tb[:, 1] - tb[:, 0]).clamp(min=0)
T = cnt.numel()
tbx = (w + 15) // 16
res = {"workload": name, "tiles": T, "mean": float(cnt.float().mean()), "std": float(cnt.float().std()), "max": int(cnt.max())}
wgs = (T + 3) // 4
G = ((wgs + 7) // 8) * 8
per = G // 8
band = []
for x in range(8):
    lo, hi = x * per * 4, min(T, (x + 1) * per * 4)
    band.append(int(cnt[lo:hi].sum()))
res["band_sums"] = band
res["band_max_over_mean"] = max(band) / (sum(band) / 8)
for gran in (1, 2, 4, 8, 15, 30):  # interleave in groups of `gran` workgroups
    sums = [0] * 8
    for g0 in range(0, wgs, gran):
        x = (g0 // gran) % 8
        sums[x] += int(cnt[g0 * 4:min(T, (g0 + gran) * 4)].sum())
    res[f"interleave_{gran}_wg_max_over_mean"] = max(sums) / (sum(sums) / 8)
rows = cnt[: (T // tbx) * tbx].view(-1, tbx).sum(1)
res["row_sums"] = [int(v) for v in rows]
print(json.dumps(res))
