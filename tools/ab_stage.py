#!/usr/bin/env python3
"""A/B aid: per-stage times (hipEvents, eager launches) and the graph-replay step time of one named workload, for
comparing builds or environment switches (e.g. BRUSH_PB_HANDBACK=0/1) process by process.

    python tools/ab_stage.py S3 [steps]      # workloads: S1, mid (3072 tiles), hd (3600 tiles), dense, c3, S3
"""
import json
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import brush_amd  # noqa: E402
from brush_amd import render as R  # noqa: E402
from brush_amd.profiler import StageProfiler  # noqa: E402
from brush_amd.synthetic import synthetic_cloud  # noqa: E402

CFG = {"S1": (1 << 20, 1920, 1080, 3, 1.0, None), "mid": (1 << 19, 1024, 768, 3, 1.0, None),
       "hd": (1 << 19, 1280, 720, 3, 1.0, None), "sq1k": (1 << 20, 1024, 1024, 3, 1.0, None),
       "sq512": (1 << 20, 512, 512, 3, 1.0, None), "t2040": (1 << 19, 960, 540, 3, 1.0, None),
       "t2500": (1 << 19, 800, 800, 3, 1.0, None), "t1728": (1 << 19, 768, 576, 3, 1.0, None),
       "t972": (1 << 19, 576, 432, 3, 1.0, None), "t2500d": (1 << 20, 800, 800, 3, 0.25, None), "vga": (1 << 19, 640, 480, 3, 1.0, None), "dense": (1 << 20, 1920, 1080, 3, 0.25, None),
       "c3": (3_000_000, 1920, 1080, 3, 0.12, 40_000_000), "S3": (20_971_520, 3840, 2160, 3, 1.0, 24_000_000)}
name = sys.argv[1] if len(sys.argv) > 1 else "S1"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
n, w, h, deg, mm, cap = CFG[name]
dev = torch.device("cuda:0")
cloud = synthetic_cloud(n, deg, seed=4, mean_mult=mm)
p = {k: torch.as_tensor(v, device=dev) for k, v in cloud.items()}
del cloud
focal = brush_amd.fov_to_focal(math.pi * 0.5, w)
cam = brush_amd.Camera([0.0, 0.0, -8.0], [0.0, 0.0, 0.0, 1.0], brush_amd.focal_to_fov(focal, w), brush_amd.focal_to_fov(focal, h),
                       (0.5, 0.5))
C = (deg + 1) ** 2
v_out = torch.full((h, w, 4), 1.0 / (4 * w * h), device=dev)
block = torch.zeros(R.grad_block_layout(n, C)[1], device=dev)


def fwd_bwd():
    out, aux, u = R._forward_impl(cam, (w, h), p["means"], p["log_scales"], p["quats"], p["sh"], p["raw_opac"], False, cap)
    R._backward_impl(u, aux, p["means"], p["log_scales"], p["quats"], p["raw_opac"], C, out, v_out, block)
    return aux


for _ in range(3):
    aux = fwd_bwd()
torch.cuda.synchronize()
with StageProfiler() as prof:
    acc = None
    for _ in range(steps):
        fwd_bwd()
        ms = prof.read_ms()
        acc = ms if acc is None else {k: acc[k] + ms[k] for k in ms}
stage = {k: round(v / steps, 5) for k, v in acc.items()}
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    fwd_bwd()
for _ in range(3):
    g.replay()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(steps):
    g.replay()
torch.cuda.synchronize()
ms_step = (time.perf_counter() - t0) * 1e3 / steps
env = {k: v for k, v in os.environ.items() if k.startswith("BRUSH_")}
print(json.dumps({"workload": name, "env": env, "ms_per_step_graph": round(ms_step, 4), "stage_ms": stage,
                  "num_visible": aux.read_num_visible(), "num_intersections": aux.read_num_intersections()}))
