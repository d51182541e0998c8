#!/usr/bin/env python3
"""The reference-shaped bench series (crates/brush-render/benches/render_bench.rs:23-30,135-197).

Same workload definition as the reference's divan bench: one seeded 2^21-splat cloud (bench distribution, SH
degree 0), the first `density * 2^21` splats of it, means scaled by 1.0 ("base"/"hd") or 0.25 ("dense"), camera at
(0,0,-8) fov 90 deg, 512x512 ("base", "dense") or 1024x1024 ("hd"); `fwd` = packed-RGBA8 forward without a gradient
graph (`splats.render(.., true)`), `bwd` = forward + backward of mean(img).  Like the reference, one sample is
INTERNAL_ITERS = 4 back-to-back iterations followed by one device sync, and the figure is milliseconds per 4
iterations, so the numbers sit next to the only published ones (test_cases/NerfStudioRefGen.ipynb:485-492, hardware
unstated).  Writes one JSON document to --out.
"""
from __future__ import annotations

import argparse
import json
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

DENSITIES = [0.1, 0.2, 0.3, 0.4, 0.5, 0.6, 0.7, 0.8, 0.9, 1.0]
INTERNAL_ITERS = 4
# NerfStudioRefGen.ipynb:485-492, ms per 4 iterations, GPU unstated
REFERENCE_MS_PER_4 = {
    "bwd_base": [12.04, 11.48, 15.75, 21.15, 27.11, 31.42, 37.59, 41.36, 45.13, 50.42],
    "bwd_dense": [14.86, 18.36, 20.12, 22.65, 26.06, 30.06, 34.31, 39.51, 43.65, 47.8],
    "bwd_hd": [19.29, 24.32, 31.02, 35.11, 41.33, 48.38, 55.74, 62.54, 69.79, 76.81],
    "fwd_base": [2.679, 3.485, 4.867, 6.565, 7.962, 9.237, 10.89, 11.96, 13.23, 14.91],
    "fwd_dense": [4.608, 5.745, 6.232, 7.115, 8.301, 9.588, 11.37, 12.49, 13.66, 14.75],
    "fwd_hd": [4.432, 5.395, 6.918, 9.053, 11.14, 13.23, 15.05, 15.83, 17.5, 19.22],
}
GROUPS = {"base": (1.0, 512), "dense": (0.25, 512), "hd": (1.0, 1024)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "s2_series.json"))
    ap.add_argument("--samples", type=int, default=5)  # TARGET_SAMPLE_COUNT
    ap.add_argument("--densities", default=",".join(str(d) for d in DENSITIES))
    args = ap.parse_args()
    import brush_amd
    from brush_amd import render as R
    from brush_amd.synthetic import synthetic_cloud

    dev = torch.device("cuda:0")
    full = synthetic_cloud(1 << 21, 0, seed=4, mean_mult=1.0)
    full = {k: torch.as_tensor(v, device=dev) for k, v in full.items()}
    dens_list = [float(d) for d in args.densities.split(",")]
    res = {"what": __doc__.split("\n\n")[1].replace("\n", " "), "unit": "ms per 4 iterations (median of samples)",
           "internal_iters": INTERNAL_ITERS, "samples": args.samples, "series": {}}
    for gname, (mult, side) in GROUPS.items():
        w = h = side
        focal = brush_amd.fov_to_focal(math.pi * 0.5, w)
        cam = brush_amd.Camera([0.0, 0.0, -8.0], [0.0, 0.0, 0.0, 1.0], brush_amd.focal_to_fov(focal, w),
                               brush_amd.focal_to_fov(focal, h), (0.5, 0.5))
        v_out = torch.full((h, w, 4), 1.0 / (4 * w * h), dtype=torch.float32, device=dev)
        for mode in ("fwd", "bwd"):
            key = f"{mode}_{gname}"
            rows = []
            for dens in dens_list:
                n = int((1 << 21) * dens)
                p = {k: v[:n].contiguous() for k, v in full.items()}
                p["means"] = p["means"] * mult
                block = torch.empty(R.grad_block_layout(n, 1)[1], dtype=torch.float32, device=dev)

                def it():
                    if mode == "fwd":
                        return R._forward_impl(cam, (w, h), p["means"], p["log_scales"], p["quats"], p["sh"],
                                               p["raw_opac"], True, None)[1]
                    out, aux, u = R._forward_impl(cam, (w, h), p["means"], p["log_scales"], p["quats"], p["sh"],
                                                  p["raw_opac"], False, None)
                    R._backward_impl(u, aux, p["means"], p["log_scales"], p["quats"], p["raw_opac"], 1, out, v_out, block)
                    return aux

                for _ in range(3):
                    aux = it()
                torch.cuda.synchronize()
                eager = []
                for _ in range(args.samples):
                    t0 = time.perf_counter()
                    for _ in range(INTERNAL_ITERS):
                        aux = it()
                    torch.cuda.synchronize()
                    eager.append((time.perf_counter() - t0) * 1e3)
                # the same 4 iterations as one captured hipGraph (the op is capture-safe)
                g = torch.cuda.CUDAGraph()
                side_stream = torch.cuda.Stream()
                side_stream.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(side_stream):
                    it()
                torch.cuda.current_stream().wait_stream(side_stream)
                torch.cuda.synchronize()
                with torch.cuda.graph(g):
                    for _ in range(INTERNAL_ITERS):
                        it()
                g.replay()
                torch.cuda.synchronize()
                graph = []
                for _ in range(args.samples):
                    t0 = time.perf_counter()
                    g.replay()
                    torch.cuda.synchronize()
                    graph.append((time.perf_counter() - t0) * 1e3)
                eager.sort(); graph.sort()
                ref = REFERENCE_MS_PER_4[key][DENSITIES.index(dens)] if dens in DENSITIES else None
                rows.append({"density": dens, "splats": n, "width": w, "height": h, "mean_mult": mult,
                             "num_visible": aux.read_num_visible(), "num_intersections": aux.read_num_intersections(),
                             "overflow": int(aux.overflow.item()),
                             "ms_per_4_eager": round(eager[len(eager) // 2], 4),
                             "ms_per_4_graph": round(graph[len(graph) // 2], 4),
                             "reference_ms_per_4_unstated_gpu": ref})
                print(key, rows[-1], flush=True)
                del g, block, p
            res["series"][key] = rows
    os.makedirs(os.path.dirname(args.out), exist_ok=True)
    json.dump(res, open(args.out, "w"), indent=1)
    print("wrote", args.out)


if __name__ == "__main__":
    main()
