#!/usr/bin/env python3
"""Feasibility probe: does an HBM-bound stream on a second (lower-priority) HIP queue overlap with the S1 training
iteration's VALU-/launch-bound kernels?  Times the iteration alone, a read-modify-write of `MB` megabytes alone, and
both with the RMW forked at the start of every iteration and joined at its end.

    python tools/exp_overlap.py [steps] [MB]
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
from brush_amd.synthetic import synthetic_cloud  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
mb = int(sys.argv[2]) if len(sys.argv) > 2 else 640
dev = torch.device("cuda:0")
n, w, h, deg = 1 << 20, 1920, 1080, 3
cloud = synthetic_cloud(n, deg, seed=4, mean_mult=1.0)
p = {k: torch.as_tensor(v, device=dev) for k, v in cloud.items()}
focal = brush_amd.fov_to_focal(math.pi * 0.5, w)
cam = brush_amd.Camera([0.0, 0.0, -8.0], [0.0, 0.0, 0.0, 1.0], brush_amd.focal_to_fov(focal, w), brush_amd.focal_to_fov(focal, h),
                       (0.5, 0.5))
splats = brush_amd.Splats(p["means"], p["sh"], p["quats"], p["raw_opac"], p["log_scales"])
trainer = brush_amd.SplatTrainer(splats, brush_amd.TrainConfig(warmup_steps=0, max_refine_step=0))
gt = torch.rand((h, w, 3), dtype=torch.float32, device=dev)
dummy = torch.ones(mb * 1024 * 1024 // 4, device=dev)


def timed(fn, k, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(k):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) * 1e3 / k


res = {"rmw_MB_read_plus_written": 2 * mb}
for main_prio, side_prio, tag in ((0, 0, "equal priority"), (-1, 0, "iteration on a high-priority queue")):
    main = torch.cuda.Stream(device=dev, priority=main_prio)
    side = torch.cuda.Stream(device=dev, priority=side_prio)
    ev_fork, ev_join = torch.cuda.Event(), torch.cuda.Event()

    def train_only():
        with torch.cuda.stream(main):
            trainer.step(splats, cam, gt, 1.0, 1, None, None)

    def rmw_only():
        with torch.cuda.stream(side):
            dummy.mul_(1.0)

    def both():
        with torch.cuda.stream(main):
            ev_fork.record(main)
            side.wait_event(ev_fork)
            with torch.cuda.stream(side):
                dummy.mul_(1.0)
                ev_join.record(side)
            trainer.step(splats, cam, gt, 1.0, 1, None, None)
            main.wait_event(ev_join)

    def serial():
        with torch.cuda.stream(main):
            dummy.mul_(1.0)
            trainer.step(splats, cam, gt, 1.0, 1, None, None)

    t_train, t_rmw, t_serial, t_both = timed(train_only, steps), timed(rmw_only, steps), timed(serial, steps), timed(both, steps)
    res[tag] = {"train_ms": round(t_train, 4), "rmw_ms": round(t_rmw, 4), "serial_one_stream_ms": round(t_serial, 4),
                "forked_ms": round(t_both, 4), "hidden_fraction_of_rmw": round((t_serial - t_both) / t_rmw, 3)}
print(json.dumps(res))
