import sys
import numpy as np, torch
sys.path.insert(0, ".")
from tests import helpers as H
import brush_amd
from tests.torch_trainer import TorchSplatTrainer
from brush_amd import render as R
from brush_amd.train import l1_ssim_loss
dev = torch.device("cuda:0")
cloud = H.synthetic_cloud(3000, 2, seed=9, mean_mult=0.0005)
cloud["log_scales"] = cloud["log_scales"] - 3.0
w, h = 128, 80
c = H.reference_test_camera(w, h)
cam = brush_amd.Camera(c["position"], c["rotation_xyzw"], c["fov_x"], c["fov_y"], c["center_uv"])
def mk():
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    return brush_amd.Splats(t(cloud["means"]), t(cloud["sh"]), t(cloud["quats"]), t(cloud["raw_opac"]), t(cloud["log_scales"]))
torch.manual_seed(3)
gt = torch.rand((h, w, 3), device=dev)
a, b = mk(), mk()
cfg = brush_amd.TrainConfig(warmup_steps=0)
ta, tb = brush_amd.SplatTrainer(a, cfg), TorchSplatTrainer(b, cfg)
# gradients of both paths at the start
means, ls, q, sh, ro = [x.detach() for x in (a.means, a.log_scales, a.rotation, a.sh_coeffs, a.raw_opacity)]
pred, aux, u = R._forward_impl(cam, (w, h), means, ls, q, sh, ro, False, None)
loss, v_pred = l1_ssim_loss(pred, gt, 0.2, 11, 1.0)
g, block = R._backward_impl(u, aux, means, ls, q, ro, 9, pred, v_pred)
g = {k: v.clone() for k, v in g.items()}
for p in (b.means, b.raw_opacity, b.sh_coeffs, b.rotation, b.log_scales): p.grad = None
pred2, aux2 = b.render(cam, (w, h), False)
l2 = (pred2[..., :3] - gt).abs().mean() * 0.8 - tb.ssim.ssim(pred2[None, ..., :3], gt[None]) * 0.2
l2.backward()
print("loss", float(loss), float(l2), "V", aux.read_num_visible())
for k, p in (("v_means", b.means), ("v_scales", b.log_scales), ("v_quats", b.rotation), ("v_opac", b.raw_opacity), ("v_sh", b.sh_coeffs)):
    x, y = g[k].flatten(), p.grad.flatten()
    nz = (x != 0) | (y != 0)
    flips = ((x * y) < 0).sum().item()
    onezero = (((x == 0) != (y == 0))).sum().item()
    print(k, "maxabs", float(y.abs().max()), "maxdiff", float((x - y).abs().max()), "nonzero", int(nz.sum()), "signflips", flips, "one-zero", onezero)
la, _, _ = ta.step(a, cam, gt)
lb, _, _ = tb.step(b, cam, gt)
for name, lr in (("means", cfg.lr_mean), ("log_scales", cfg.lr_scale), ("rotation", cfg.lr_rotation), ("raw_opacity", cfg.lr_opac), ("sh_coeffs", cfg.lr_coeffs_dc)):
    d = (getattr(a, name).detach() - getattr(b, name).detach()).abs()
    print(name, "maxdiff/lr", float(d.max()) / lr, "frac>0.05lr", float((d > 0.05 * lr).float().mean()))
la, _, _ = ta.step(a, cam, gt)
lb, _, _ = tb.step(b, cam, gt)
print("loss after 1 step", float(la), float(lb))
