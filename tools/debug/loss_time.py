"""Development aid: time of brush_l1_ssim_loss (both kernels) at 1080p, and its loss value / gradient checksum."""
import sys, os, json
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from brush_amd import train as T
dev = torch.device("cuda:0")
h, w = 1080, 1920
g = torch.Generator(device=dev).manual_seed(1)
pred = torch.rand((h, w, 4), device=dev, generator=g)
gt = torch.rand((h, w, 3), device=dev, generator=g)
for _ in range(5):
    loss, v = T.l1_ssim_loss(pred, gt, 0.2)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
K = 50
e0.record()
for _ in range(K):
    loss, v = T.l1_ssim_loss(pred, gt, 0.2)
e1.record()
torch.cuda.synchronize()
print(json.dumps({"lib": os.environ.get("BRUSH_HIP_LIB", "default").split("/")[-1], "us_per_call": e0.elapsed_time(e1) / K * 1e3,
                  "loss": float(loss.item()), "grad_abs_sum": float(v.double().abs().sum().item())}))
