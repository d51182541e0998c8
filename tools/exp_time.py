#!/usr/bin/env python3
"""Times the pieces of the view-sharded step on ONE GPU (RCCL itself needs a multi-GPU node): the per-view record
kernel, and the per-splat reduction over W views (dense output and fused Adam) with W real views rendered from
different cameras of the S1 cloud.  DESIGN.md's N = 8 arithmetic is built from these numbers."""
import ctypes as C
import json
import math
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import brush_amd  # noqa: E402
from brush_amd import _lib, dist as BD, render as R  # noqa: E402
from brush_amd.synthetic import synthetic_cloud  # noqa: E402

dev = torch.device("cuda:0")
n, w, h, deg = 1 << 20, 1920, 1080, 3
Cc = (deg + 1) ** 2
cloud = synthetic_cloud(n, deg, seed=4)
p = {k: torch.as_tensor(v, device=dev) for k, v in cloud.items()}
v_out = torch.full((h, w, 4), 1.0 / (4 * w * h), device=dev)


def cam(rank):
    focal = brush_amd.fov_to_focal(math.pi * 0.5, w)
    ang = 0.35 * rank
    return brush_amd.Camera([-8.0 * math.sin(ang), 0.0, -8.0 * math.cos(ang)], [0.0, math.sin(ang / 2), 0.0, math.cos(ang / 2)],
                            brush_amd.focal_to_fov(focal, w), brush_amd.focal_to_fov(focal, h), (0.5, 0.5))


def t_us(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps


res = {"workload": f"{n} splats @{w}x{h}, SH degree {deg}"}
views = []
for r in range(8):
    out, aux, u = R._forward_impl(cam(r), (w, h), p["means"], p["log_scales"], p["quats"], p["sh"], p["raw_opac"], False, None)
    x = BD.ViewExchange(n, Cc, dev)
    x.begin(aux)
    x.backward_records(u, aux, p["means"], p["log_scales"], p["quats"], p["raw_opac"], out, v_out)
    recs = x.gather()
    views.append((recs[0].clone(), x.metas.clone(), aux.read_num_visible()))
    if r == 0:
        res["backward_records_us"] = round(t_us(lambda: x.backward_records(u, aux, p["means"], p["log_scales"], p["quats"],
                                                                          p["raw_opac"], out, v_out)), 1)
        blk = torch.empty(R.grad_block_layout(n, Cc)[1], device=dev)
        res["backward_dense_us"] = round(t_us(lambda: R._backward_impl(u, aux, p["means"], p["log_scales"], p["quats"],
                                                                      p["raw_opac"], Cc, out, v_out, blk)), 1)
res["visible_per_view"] = [v[2] for v in views]
for W in (1, 2, 4, 8):
    # the packed layout the exact-size all-gather leaves: view after view, row offsets on the device
    rows = max(v[0].shape[0] for v in views[:W])
    counts = [v[0].shape[0] for v in views[:W]]
    offs = [sum(counts[:i]) for i in range(W)]
    x = BD.ViewExchange(n, Cc, dev)
    x.world = W
    x.metas = torch.cat([v[1] for v in views[:W]], 0).contiguous()
    x._ensure_capacity(rows)
    x._rows = sum(counts)
    x._offsets_dev = torch.tensor(offs, dtype=torch.int32, device=dev)
    flat = x.gathered[:x._rows * 16].view(-1, 16)
    for i, v in enumerate(views[:W]):
        flat[offs[i]:offs[i] + counts[i]] = v[0]
    blk = torch.empty(R.grad_block_layout(n, Cc)[1], device=dev)
    res[f"reduce_dense_W{W}_us"] = round(t_us(lambda: x.reduce_dense(p["means"], blk)), 1)
    prm = {k: v.clone() for k, v in p.items()}
    m1 = torch.zeros(n * (11 + 3 * Cc), device=dev)
    m2 = torch.zeros_like(m1)
    nxt = torch.empty_like(prm["quats"])
    acc, cnt = torch.zeros(n, device=dev), torch.zeros(n, device=dev)
    cfg = _lib.BrushAdamConfig(1.6e-4, 0.01, 0.002, 0.05, 0.004, 0.05, 0.9, 0.999, 1e-15, 1, 1)
    res[f"reduce_adam_W{W}_us"] = round(t_us(lambda: x.reduce_adam(cfg, (w, h), prm["means"], prm["log_scales"], prm["quats"],
                                                                   prm["raw_opac"], prm["sh"], m1, m2, nxt, acc, cnt)), 1)
    res[f"gathered_MB_per_rank_W{W}"] = round((sum(counts) - counts[0]) * 64 / 1e6, 2)
    res[f"gathered_MB_per_rank_padded_W{W}"] = round((W - 1) * rows * 64 / 1e6, 2)
print(json.dumps(res))
if len(sys.argv) > 1:
    json.dump(res, open(sys.argv[1], "w"), indent=1)
