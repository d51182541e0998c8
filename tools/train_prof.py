"""Runs a few full training iterations on the S1 workload (for rocprofv3 --kernel-trace --stats)."""
import sys, time
import torch
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import math
from brush_amd.synthetic import synthetic_cloud
import brush_amd
dev = torch.device("cuda:0")
n, w, h, deg = 1 << 20, 1920, 1080, 3
cloud = synthetic_cloud(n, deg, seed=4)
p = {k: torch.from_numpy(v).to(dev) for k, v in cloud.items()}
focal = brush_amd.fov_to_focal(math.pi * 0.5, w)  # render_bench.rs:163-174: (0,0,-8), fov 90 deg on x
cam = brush_amd.Camera([0.0, 0.0, -8.0], [0.0, 0.0, 0.0, 1.0], brush_amd.focal_to_fov(focal, w), brush_amd.focal_to_fov(focal, h), (0.5, 0.5))
splats = brush_amd.Splats(p["means"], p["sh"], p["quats"], p["raw_opac"], p["log_scales"])
deferred = (sys.argv[2] if len(sys.argv) > 2 else "deferred") == "deferred"  # argv[2]: deferred | eager (SH block's Adam)
trainer = brush_amd.SplatTrainer(splats, brush_amd.TrainConfig(warmup_steps=0, max_refine_step=0, deferred_sh_adam=deferred))
gt = torch.rand((h, w, 3), dtype=torch.float32, device=dev)
for _ in range(3):
    trainer.step(splats, cam, gt)
torch.cuda.synchronize()
t0 = time.perf_counter()
K = int(sys.argv[1]) if len(sys.argv) > 1 else 10
for _ in range(K):
    trainer.step(splats, cam, gt)
t_host = time.perf_counter()  # every launch of the K iterations is enqueued; the GPU may still be working
torch.cuda.synchronize()
t1 = time.perf_counter()
print("host enqueue ms/iter", (t_host - t0) / K * 1e3)
trainer.sync(splats)  # the deferred SH steps of the K iterations (every splat this camera never sees)
torch.cuda.synchronize()
t2 = time.perf_counter()
print("mode", "deferred" if deferred else "eager", "ms/iter", (t1 - t0) / K * 1e3, "final sync ms", (t2 - t1) * 1e3,
      "ms/iter incl. sync", (t2 - t0) / K * 1e3)
