"""GPU checks of the fused training-iteration kernels against the plain-PyTorch
restatement in tests/torch_trainer.py.  Parity unpinned: the reference holds no fixtures for its
trainer; the checker follows the text of train.rs / ssim.rs / burn's Adam."""
import math

import numpy as np
import pytest

from tests import helpers as H

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    import torch

    import brush_amd  # noqa: F401

    assert torch.cuda.is_available()
    return torch.device("cuda:0")


def _torch_loss(pred, gt, ssim_weight, window=11):
    """train.rs:243-268 with the 2-D window of ssim.rs:36-40 (not the separable form)."""
    import torch
    import torch.nn.functional as F

    g = torch.tensor([math.exp(-((x - window // 2) ** 2) / (2.0 * 1.5 ** 2)) for x in range(window)],
                     dtype=torch.float32, device=pred.device)
    g = g / g.sum()
    w2 = torch.outer(g, g)[None, None].repeat(3, 1, 1, 1)
    pred_rgb = pred[..., :3]
    cmp = pred if gt.shape[-1] == 4 else pred_rgb
    loss = (cmp - gt).abs().mean()
    if ssim_weight > 0:
        x, y = pred_rgb[None].permute(0, 3, 1, 2), gt[None, ..., :3].permute(0, 3, 1, 2)
        blur = lambda t: F.conv2d(t, w2, None, stride=1, padding=-(-window // 2), groups=3)  # div_ceil, ssim.rs:49
        mu_x, mu_y = blur(x), blur(y)
        s_xx = (blur(x * x) - mu_x * mu_x).clamp_min(0)
        s_yy = (blur(y * y) - mu_y * mu_y).clamp_min(0)
        s_xy = blur(x * y) - mu_x * mu_y
        c1, c2 = 0.01 ** 2, 0.03 ** 2
        ssim = (((mu_x * mu_y * 2 + c1) * (s_xy * 2 + c2)) / ((mu_x * mu_x + mu_y * mu_y + c1) * (s_xx + s_yy + c2))).mean()
        loss = loss * (1.0 - ssim_weight) - ssim * ssim_weight
    return loss


@pytest.mark.parametrize("w,h,gtc,ssim_w,scale,window",
                         [(123, 82, 3, 0.2, 1.0, 11), (64, 64, 4, 0.2, 0.5, 11), (200, 37, 3, 0.0, 1.0, 11),
                          (33, 95, 4, 0.0, 0.25, 11), (31, 9, 3, 0.5, 1.0, 11), (1920, 1080, 3, 0.2, 1.0, 11),
                          # other TrainConfig::ssim_window_size values (train.rs:63): every compiled odd size
                          (123, 82, 3, 0.2, 1.0, 3), (97, 131, 4, 0.2, 1.0, 5), (123, 82, 3, 0.3, 0.5, 7),
                          (150, 40, 3, 0.2, 1.0, 9), (123, 82, 4, 0.2, 1.0, 13), (260, 75, 3, 0.2, 1.0, 15)])
def test_l1_ssim_loss_matches_torch(dev, w, h, gtc, ssim_w, scale, window):
    import torch

    from brush_amd.train import l1_ssim_loss

    torch.manual_seed(w * 7 + h)
    pred = torch.rand((h, w, 4), device=dev)
    gt = torch.rand((h, w, gtc), device=dev)
    gt[: h // 2] = (pred[: h // 2, :, :gtc] + 0.05 * torch.randn((h // 2, w, gtc), device=dev)).clamp(0, 1)  # correlated part
    gt[0, 0] = pred[0, 0, :gtc]  # exact ties: sign(0) = 0
    loss, v_pred = l1_ssim_loss(pred, gt, ssim_w, window, scale)
    p = pred.clone().requires_grad_(True)
    want = _torch_loss(p, gt, ssim_w, window)
    (want * scale).backward()
    assert abs(float(loss) - float(want)) <= 2e-6 + 1e-5 * abs(float(want))
    ref = p.grad
    err = float((v_pred - ref).abs().max())
    assert err <= 2e-5 * float(ref.abs().max()), (err, float(ref.abs().max()))
    if gtc == 3:
        assert float(v_pred[..., 3].abs().max()) == 0.0


def test_l1_ssim_loss_rejects_bad_arguments(dev):
    import torch

    from brush_amd import _lib
    from brush_amd.train import l1_ssim_loss

    pred = torch.rand((8, 8, 4), device=dev)
    with pytest.raises(ValueError):
        l1_ssim_loss(pred, torch.rand((8, 8, 2), device=dev), 0.2)
    for bad in (4, 1, 17):  # even windows change the map size (ssim.rs:49); sizes above 15 are not compiled
        with pytest.raises(_lib.BrushError):
            l1_ssim_loss(pred, torch.rand((8, 8, 3), device=dev), 0.2, window=bad)


@pytest.mark.parametrize("n,deg,vjp", [(1024, 3, 0), (1003, 1, 1), (5, 0, 0), (4096, 2, 1)])
def test_adam_step_matches_burn_form(dev, n, deg, vjp):
    """brush_adam_step vs burn 0.16 Adam::step restated in torch, 3 steps, with the SH-rest lerp."""
    import ctypes as C

    import torch

    from brush_amd import _lib

    torch.manual_seed(n)
    ncoef = (deg + 1) ** 2
    shapes = [(n, 3), (n, 3), (n, 4), (n,), (n, ncoef, 3)]
    params = [torch.randn(s, device=dev) for s in shapes]
    ref = [p.clone() for p in params]
    lrs = [1.6e-4, 0.01, 0.002, 0.05, 0.004]  # means, scales, quats, opac, sh
    m1 = torch.zeros(n * (11 + 3 * ncoef), device=dev)
    m2 = torch.zeros_like(m1)
    rm = [torch.zeros_like(p) for p in params]
    rv = [torch.zeros_like(p) for p in params]
    b1, b2, eps, lerp = 0.9, 0.999, 1e-15, 1.0 / 20.0
    l = _lib.lib()
    for t in range(1, 4):
        grads = [torch.randn(s, device=dev) * (10.0 ** float(torch.randint(-6, 1, (1,)))) for s in shapes]
        grads[4][::3] = 0.0  # untouched splats: zero gradient rows
        cfg = _lib.BrushAdamConfig(lrs[0], lrs[1], lrs[2], lrs[3], lrs[4], lerp, b1, b2, eps, t, vjp)
        with torch.cuda.device(dev):
            _lib.check(l.brush_adam_step(C.byref(cfg), n, deg, *[p.data_ptr() for p in params],
                                         *[g.data_ptr() for g in grads], m1.data_ptr(), m2.data_ptr(),
                                         torch.cuda.current_stream().cuda_stream), "brush_adam_step")
        if vjp:  # autograd through rot / |rot| (gaussian_splats.rs:174-175)
            rot = ref[2].clone().requires_grad_(True)
            (rot / torch.sqrt(torch.sum(rot * rot, dim=1, keepdim=True))).backward(grads[2])
            grads[2] = rot.grad
        for i in range(5):
            rm[i] = rm[i] * b1 + grads[i] * (1.0 - b1)
            rv[i] = rv[i] * b2 + (grads[i] * grads[i]) * (1.0 - b2)
            delta = (rm[i] / (1.0 - b1 ** t)) / ((rv[i] / (1.0 - b2 ** t)).sqrt() + eps)
            stepped = ref[i] - delta * lrs[i]
            if i == 4 and ncoef > 1:
                stepped[:, 1:] = ref[i][:, 1:] * (1.0 - lerp) + stepped[:, 1:] * lerp
            ref[i] = stepped
    for p, r in zip(params, ref):
        assert float((p - r).abs().max()) <= 2e-6 * (1.0 + float(r.abs().max()))
    off = 0
    for r in rm:
        assert float((m1[off:off + r.numel()] - r.flatten()).abs().max()) <= 2e-6 * float(r.abs().max())
        off += r.numel()


def test_refine_stats_match_torch(dev):
    import ctypes as C

    import torch

    import brush_amd
    from brush_amd import _lib
    from brush_amd import dist as BD
    from brush_amd import render as R

    n, w, h = 30000, 320, 200
    cloud = H.synthetic_cloud(n, 1, seed=12, mean_mult=0.01)
    p = {k: torch.from_numpy(v).to(dev) for k, v in cloud.items()}
    c = H.reference_test_camera(w, h)
    cam = brush_amd.Camera(c["position"], c["rotation_xyzw"], c["fov_x"], c["fov_y"], c["center_uv"])
    out, aux, u = R._forward_impl(cam, (w, h), p["means"], p["log_scales"], p["quats"], p["sh"], p["raw_opac"], False, None)
    g, _ = R._backward_impl(u, aux, p["means"], p["log_scales"], p["quats"], p["raw_opac"], 4, out,
                            torch.randn((h, w, 4), device=dev))
    accum = torch.rand(n, device=dev)
    counts = torch.ones(n, device=dev)
    want = BD.densification_stats(g["v_xy"], aux, (w, h))
    want_accum, want_counts = accum + want[0], counts + want[1]
    s = aux._as_struct()
    with torch.cuda.device(dev):
        _lib.check(_lib.lib().brush_refine_stats(C.byref(s), g["v_xy"].data_ptr(), n, w, h, accum.data_ptr(),
                                                 counts.data_ptr(), torch.cuda.current_stream().cuda_stream),
                   "brush_refine_stats")
    assert torch.equal(counts, want_counts)
    assert float((accum - want_accum).abs().max()) <= 1e-6 * (1.0 + float(want_accum.abs().max()))
    assert 0 < int(want[1].sum()) == aux.read_num_visible()


def test_hip_trainer_tracks_torch_trainer(dev):
    """Same scene, same target, three iterations: the fused HIP iteration and the PyTorch restatement
    (autograd through the op + conv2d SSIM + burn-form Adam) stay together."""
    import torch

    import brush_amd
    from tests.torch_trainer import TorchSplatTrainer

    cloud = H.synthetic_cloud(3000, 2, seed=9, mean_mult=0.0005)
    cloud["log_scales"] = cloud["log_scales"] - 3.0  # many small splats in front of the camera, each one a small part of the image
    w, h = 128, 80
    c = H.reference_test_camera(w, h)
    cam = brush_amd.Camera(c["position"], c["rotation_xyzw"], c["fov_x"], c["fov_y"], c["center_uv"])

    def mk():
        t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
        return brush_amd.Splats(t(cloud["means"]), t(cloud["sh"]), t(cloud["quats"]), t(cloud["raw_opac"]),
                                t(cloud["log_scales"]))

    torch.manual_seed(3)
    gt = torch.rand((h, w, 3), device=dev)
    a, b = mk(), mk()
    cfg = brush_amd.TrainConfig(warmup_steps=0, max_refine_step=0)  # the torch restatement has no refinement
    ta, tb = brush_amd.SplatTrainer(a, cfg), TorchSplatTrainer(b, cfg)
    for i in range(3):
        la, _, _ = ta.step(a, cam, gt)
        lb, _, _ = tb.step(b, cam, gt)
        assert abs(float(la) - float(lb)) <= (1e-5 if i == 0 else 2e-4), (i, float(la), float(lb))
    ta.sync(a)
    # Adam's first steps move every touched parameter by ~lr regardless of gradient size, so a
    # sign flip of a ~0 gradient is a 2*lr difference: compare in units of the learning rate.
    for name, lr in (("means", cfg.lr_mean), ("log_scales", cfg.lr_scale), ("rotation", cfg.lr_rotation),
                     ("raw_opacity", cfg.lr_opac), ("sh_coeffs", cfg.lr_coeffs_dc)):
        pa, pb = getattr(a, name).detach(), getattr(b, name).detach()
        d = (pa - pb).abs()
        assert float(d.max()) <= 3 * 2.2 * lr, name  # at worst opposite directions on all three steps
        assert float((d > 0.05 * lr).float().mean()) < 0.02, name  # and that is rare
    assert torch.equal(ta.xy_grad_counts, tb.xy_grad_counts)
    assert float((ta.grad_2d_accum - tb.grad_2d_accum).abs().max()) <= 1e-3 * (1e-9 + float(tb.grad_2d_accum.abs().max()))


def test_refine_splats_follows_reference_rules(dev):
    """train.rs:395-579 on a hand-built state: clones come from the pre-step parameters, splits append
    one down-scaled splat at a sampled offset (the source is left as is), pruning by opacity then by
    scale with cumulative counts, opacity reset on refine step 0, statistics and Adam state reset."""
    import torch

    import brush_amd
    from brush_amd.train import quaternion_vec_multiply

    n, ncoef = 8, 4
    g = torch.Generator().manual_seed(0)
    means = torch.randn((n, 3), generator=g)
    quats = torch.nn.functional.normalize(torch.randn((n, 4), generator=g), dim=1)
    sh = torch.randn((n, ncoef, 3), generator=g)
    # scales: splats 0-3 small (< 0.005), 4-7 large; splat 7 huge (> cull_scale_thresh 5.0)
    log_scales = torch.log(torch.tensor([[0.001] * 3] * 4 + [[0.1, 0.2, 0.3]] * 3 + [[6.0, 0.1, 0.1]]))
    raw_opac = torch.tensor([2.0, 2.0, -7.0, 2.0, 2.0, 2.0, 2.0, 2.0])  # splat 2 nearly transparent
    splats = brush_amd.Splats(means.to(dev), sh.to(dev), quats.to(dev), raw_opac.to(dev), log_scales.to(dev))
    cfg = brush_amd.TrainConfig(warmup_steps=0, refine_every=100)
    tr = brush_amd.SplatTrainer(splats, cfg)
    tr.iter = 201  # refine step 2: no opacity reset
    # mean 2-D gradient: big for 0, 1 (clone), 2 (clone, then pruned by opacity), 4, 5 (split), 7 (split, pruned by scale)
    tr.grad_2d_accum = torch.tensor([1e-3, 4e-4, 1.0, 1e-4, 1.0, 6e-4, 1e-4, 1.0], device=dev)
    tr.xy_grad_counts = torch.tensor([1.0, 2.0, 1.0, 1.0, 0.0, 3.0, 1.0, 1.0], device=dev)
    pre = {"means": (means + 100.0).to(dev), "rotation": (quats * 2).to(dev), "sh": (sh + 1).to(dev),
           "opac": (raw_opac + 0.5).to(dev), "scales": (log_scales - 1).to(dev)}
    tr.moment1 += 1.0
    stats = tr.refine_splats(splats, pre)
    assert (stats.num_cloned, stats.num_split) == (3, 3)          # clones 0,1,2; splits 4,5,7
    assert stats.num_transparent_pruned == 2                       # splat 2 and its clone (pre-step opacity -6.5)
    assert stats.num_scale_pruned == 3                             # cumulative (train.rs:551): + splat 7; its child is 6/1.6 = 3.75
    keep = [0, 1, 3, 4, 5, 6]
    assert splats.num_splats() == 11
    f = lambda t: t.detach().cpu()
    assert torch.equal(f(splats.means)[:6], means[keep]) and torch.equal(f(splats.log_scales)[:6], log_scales[keep])
    assert torch.equal(f(splats.raw_opacity)[:6], raw_opac[keep])            # no opacity reset on refine step 2
    assert torch.equal(f(splats.means)[6:8], means[[0, 1]] + 100.0)             # clones: pre-step parameters
    assert torch.equal(f(splats.rotation)[6:8], quats[[0, 1]] * 2) and torch.equal(f(splats.sh_coeffs)[6:8], sh[[0, 1]] + 1)
    assert torch.equal(f(splats.raw_opacity)[6:8], raw_opac[[0, 1]] + 0.5)
    sp = [4, 5, 7]
    assert torch.equal(f(splats.rotation)[8:], quats[sp]) and torch.equal(f(splats.sh_coeffs)[8:], sh[sp])
    assert torch.allclose(f(splats.log_scales)[8:], torch.log(log_scales[sp].exp() / 1.6), atol=1e-6)
    gen = torch.Generator(device=dev)
    gen.manual_seed(cfg.seed)
    torch.randn((3, 3), generator=gen, device=dev)                              # the dropped source re-sample
    off = quaternion_vec_multiply(quats[sp].to(dev), torch.randn((3, 3), generator=gen, device=dev) * 0.5
                                  * log_scales[sp].exp().to(dev))
    assert torch.allclose(splats.means.detach()[8:], (means[sp] + 100.0).to(dev) + off, atol=1e-5)
    assert tr.grad_2d_accum.shape == (11,) and float(tr.grad_2d_accum.abs().sum()) == 0.0
    assert tr.moment1.numel() == 11 * (11 + 3 * ncoef) and float(tr.moment1.abs().sum()) == 0.0 and tr.opt_time == 0
    assert splats.xys_dummy.shape == (11, 2)
    tr.iter = 1                                                                  # refine step 0: opacity reset
    tr.refine_splats(splats, {"means": splats.means.detach(), "rotation": splats.rotation.detach(),
                              "sh": splats.sh_coeffs.detach(), "opac": splats.raw_opacity.detach(),
                              "scales": splats.log_scales.detach()})
    assert torch.allclose(torch.sigmoid(splats.raw_opacity.detach()), torch.full((11,), 0.004, device=dev), atol=1e-6)


def test_training_with_refinement_runs(dev):
    """step() drives refine_splats on iterations with iter % refine_every == 1 and keeps training on the
    re-sized cloud (Adam state and statistics re-created)."""
    import torch

    import brush_amd

    cloud = H.synthetic_cloud(2000, 1, seed=5, mean_mult=0.0005)
    cloud["log_scales"] = cloud["log_scales"] - 3.0
    w, h = 96, 64
    c = H.reference_test_camera(w, h)
    cam = brush_amd.Camera(c["position"], c["rotation_xyzw"], c["fov_x"], c["fov_y"], c["center_uv"])
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    splats = brush_amd.Splats(t(cloud["means"]), t(cloud["sh"]), t(cloud["quats"]), t(cloud["raw_opac"]), t(cloud["log_scales"]))
    cfg = brush_amd.TrainConfig(warmup_steps=3, refine_every=3, densify_grad_thresh=1e-6, densify_size_thresh=0.05,
                                reset_alpha_every_refine=1000)
    tr = brush_amd.SplatTrainer(splats, cfg)
    torch.manual_seed(0)
    gt = torch.rand((h, w, 3), device=dev)
    sizes, refined = [], []
    for i in range(8):
        loss, pred, aux = tr.step(splats, cam, gt)
        assert math.isfinite(float(loss))
        sizes.append(splats.num_splats())
        refined.append(tr.last_refine)
    assert [r is not None for r in refined] == [False, False, False, False, True, False, False, True]
    first = refined[4]
    assert first.num_cloned + first.num_split > 0
    assert sizes[3] == 2000 and sizes[4] == 2000 + first.num_cloned + first.num_split - first.num_scale_pruned
    assert tr.moment1.numel() == splats.num_splats() * (11 + 3 * 4) and tr.opt_time == 0


@pytest.mark.parametrize("n,deg", [(3000, 2), (3001, 3), (1026, 0)])
def test_fused_backward_adam_equals_separate_calls(dev, n, deg):
    """brush_render_backward_adam == brush_render_backward followed by brush_adam_step (same update,
    the gradient elements just never visit HBM); float4 and scalar layouts (n % 4 != 0)."""
    import torch

    import brush_amd

    cloud = H.synthetic_cloud(n, deg, seed=13, mean_mult=0.0005)
    cloud["log_scales"] = cloud["log_scales"] - 3.0
    w, h = 128, 80
    c = H.reference_test_camera(w, h)
    cam = brush_amd.Camera(c["position"], c["rotation_xyzw"], c["fov_x"], c["fov_y"], c["center_uv"])
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    mk = lambda: brush_amd.Splats(t(cloud["means"]), t(cloud["sh"]), t(cloud["quats"] * 1.7), t(cloud["raw_opac"]),
                                  t(cloud["log_scales"]))
    torch.manual_seed(5)
    gt = torch.rand((h, w, 3), device=dev)
    a, b = mk(), mk()
    cfg = brush_amd.TrainConfig(warmup_steps=0, max_refine_step=0)
    ta, tb = brush_amd.SplatTrainer(a, cfg), brush_amd.SplatTrainer(b, cfg)
    tb.fused_backward = False
    for i in range(3):
        la, _, _ = ta.step(a, cam, gt)
        lb, _, _ = tb.step(b, cam, gt)
        assert abs(float(la) - float(lb)) <= 2e-5, (i, float(la), float(lb))
    ta.sync(a), tb.sync(b)
    for name, lr in (("means", cfg.lr_mean), ("log_scales", cfg.lr_scale), ("rotation", cfg.lr_rotation),
                     ("raw_opacity", cfg.lr_opac), ("sh_coeffs", cfg.lr_coeffs_dc)):
        d = (getattr(a, name).detach() - getattr(b, name).detach()).abs()
        # the two runs differ only by the order of the raster backward's float atomics
        assert float((d > 0.05 * lr).float().mean()) < 0.02, (name, float(d.max()) / lr)
    assert float((ta.moment2 - tb.moment2).abs().max()) <= 1e-4 * float(tb.moment2.abs().max())
    assert torch.equal(ta.xy_grad_counts, tb.xy_grad_counts)
    assert float((ta.grad_2d_accum - tb.grad_2d_accum).abs().max()) <= 1e-3 * float(tb.grad_2d_accum.abs().max())


def _orbit_cameras(w, h, count):
    """Cameras on a circle around the cloud, each looking at its centre: every view sees a different subset."""
    import math

    import brush_amd

    cams = []
    for i in range(count):
        a = 2.0 * math.pi * i / count
        pos = [4.0 * math.sin(a), 0.0, -4.0 * math.cos(a)]
        # rotation about +y by `a` (xyzw), so that the camera's +z axis points at the origin
        rot = [0.0, -math.sin(a / 2.0), 0.0, math.cos(a / 2.0)]
        cams.append(brush_amd.Camera(pos, rot, 0.4, 0.3, (0.5, 0.5)))
    return cams


@pytest.mark.parametrize("n,deg,table_rows", [(4000, 3, 2048), (4096, 1, 2048), (4000, 3, 5)])
def test_deferred_sh_adam_equals_eager_bitwise(dev, n, deg, table_rows):
    """Deferred Adam of the SH block (BrushLazySh): the optimizer steps of splats a view does not see stay pending and
    are replayed when the block is next needed.  In deterministic mode (bitwise reproducible gradients) a run over views
    with changing visibility must give the SAME BITS as the eager optimizer: the loss of every step (the forward renders
    from the replayed coefficients), and after sync() every parameter and both moments.  table_rows = 5 forces the
    per-step constant table to be rebuilt (flush + new base) several times inside the run."""
    import torch

    import brush_amd
    from brush_amd import render as R

    cloud = H.synthetic_cloud(n, deg, seed=21, mean_mult=0.0003)
    cloud["log_scales"] = cloud["log_scales"] - 3.5
    w, h = 160, 96
    cams = _orbit_cameras(w, h, 5)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    mk = lambda: brush_amd.Splats(t(cloud["means"]), t(cloud["sh"]), t(cloud["quats"]), t(cloud["raw_opac"]),
                                  t(cloud["log_scales"]))
    torch.manual_seed(8)
    gts = [torch.rand((h, w, 3), device=dev) for _ in cams]
    a, b = mk(), mk()
    ta = brush_amd.SplatTrainer(a, brush_amd.TrainConfig(warmup_steps=0, max_refine_step=0, deferred_sh_adam=True))
    tb = brush_amd.SplatTrainer(b, brush_amd.TrainConfig(warmup_steps=0, max_refine_step=0, deferred_sh_adam=False))
    ta.LAZY_TABLE_ROWS = table_rows
    order = [0, 1, 2, 3, 4, 2, 0, 3, 3, 1, 4, 0, 2]
    saved, R.DETERMINISTIC = R.DETERMINISTIC, True
    try:
        seen = torch.zeros(n, dtype=torch.bool, device=dev)
        lagged = 0
        for i, v in enumerate(order):
            la, _, aux = ta.step(a, cams[v], gts[v])
            lb, _, _ = tb.step(b, cams[v], gts[v])
            assert float(la) == float(lb), (i, float(la), float(lb))
            V = aux.read_num_visible()
            assert 0 < V < n
            seen[aux.global_from_compact_gid[:V].long()] = True
            if ta._lazy is not None:
                lagged = max(lagged, int((ta._lazy_bufs[0] < ta.opt_time).sum()))
        assert ta._lazy is not None and lagged > 0, "nothing was deferred: the test would prove nothing"
        if table_rows > len(order):
            # before the flush the stored coefficients of some splats ARE behind the eager ones
            assert not torch.equal(a.sh_coeffs.detach(), b.sh_coeffs.detach())
        ta.sync(a)
        assert bool((ta._lazy_bufs[0] == ta.opt_time).all())
        for name in ("means", "log_scales", "rotation", "raw_opacity", "sh_coeffs"):
            assert torch.equal(getattr(a, name).detach(), getattr(b, name).detach()), name
        assert torch.equal(ta.moment1, tb.moment1) and torch.equal(ta.moment2, tb.moment2)
        assert torch.equal(ta.grad_2d_accum, tb.grad_2d_accum) and torch.equal(ta.xy_grad_counts, tb.xy_grad_counts)
        # a reader that bypasses the trainer: Splats.render syncs by itself and sees the eager image
        ta.step(a, cams[1], gts[1]), tb.step(b, cams[1], gts[1])
        ia, _ = a.render(cams[3], (w, h))
        ib, _ = b.render(cams[3], (w, h))
        assert torch.equal(ia, ib)
    finally:
        R.DETERMINISTIC = saved


def test_deferred_sh_adam_through_refinement(dev):
    """Refinement clones / splits / prunes whole parameter rows (train.rs:395-579): the trainer flushes what is pending
    before it takes the pre-step copies and before refine_splats reads the post-step coefficients, so a deferred run
    and an eager run refine to the same splats, bit for bit (deterministic mode)."""
    import torch

    import brush_amd
    from brush_amd import render as R

    n, deg, w, h = 2000, 3, 160, 96
    cloud = H.synthetic_cloud(n, deg, seed=23, mean_mult=0.0003)
    cloud["log_scales"] = cloud["log_scales"] - 3.5
    cams = _orbit_cameras(w, h, 3)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    mk = lambda: brush_amd.Splats(t(cloud["means"]), t(cloud["sh"]), t(cloud["quats"]), t(cloud["raw_opac"]),
                                  t(cloud["log_scales"]))
    torch.manual_seed(9)
    gts = [torch.rand((h, w, 3), device=dev) for _ in cams]
    a, b = mk(), mk()
    kw = dict(warmup_steps=2, refine_every=4, densify_grad_thresh=1e-7, densify_size_thresh=0.05, cull_scale_thresh=50.0)
    ta = brush_amd.SplatTrainer(a, brush_amd.TrainConfig(deferred_sh_adam=True, **kw))
    tb = brush_amd.SplatTrainer(b, brush_amd.TrainConfig(deferred_sh_adam=False, **kw))
    saved, R.DETERMINISTIC = R.DETERMINISTIC, True
    try:
        refined = 0
        for i in range(11):
            v = i % len(cams)
            la, _, _ = ta.step(a, cams[v], gts[v])
            lb, _, _ = tb.step(b, cams[v], gts[v])
            assert float(la) == float(lb), i
            assert a.num_splats() == b.num_splats(), i
            refined += ta.last_refine is not None
        assert refined >= 2 and a.num_splats() != n
        ta.sync(a)
        for name in ("means", "log_scales", "rotation", "raw_opacity", "sh_coeffs"):
            assert torch.equal(getattr(a, name).detach(), getattr(b, name).detach()), name
        assert torch.equal(ta.moment1, tb.moment1) and torch.equal(ta.moment2, tb.moment2)
    finally:
        R.DETERMINISTIC = saved


def test_deferred_sh_adam_equals_eager_at_headline_size(dev):
    """The same bitwise comparison at the bench's size (1 048 576 splats @1920x1080, SH degree 3, the bench cloud and three
    of its orbit views): six deterministic iterations, losses equal step by step, every parameter and both moments equal
    after sync(); ~90 % of the SH blocks are deferred at any time."""
    import math

    import torch

    import brush_amd
    from brush_amd import render as R

    n, deg, w, h = 1 << 20, 3, 1920, 1080
    cloud = H.synthetic_cloud(n, deg, seed=4, mean_mult=1.0)
    focal = brush_amd.fov_to_focal(math.pi * 0.5, w)
    fx, fy = brush_amd.focal_to_fov(focal, w), brush_amd.focal_to_fov(focal, h)
    cams = []
    for v in range(3):  # bench.py:view_camera
        ang = 0.35 * v
        cams.append(brush_amd.Camera([-8.0 * math.sin(ang), 0.0, -8.0 * math.cos(ang)],
                                     [0.0, math.sin(ang / 2), 0.0, math.cos(ang / 2)], fx, fy, (0.5, 0.5)))
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    mk = lambda: brush_amd.Splats(t(cloud["means"]), t(cloud["sh"]), t(cloud["quats"]), t(cloud["raw_opac"]),
                                  t(cloud["log_scales"]))
    torch.manual_seed(11)
    gt = torch.rand((h, w, 3), device=dev)
    a, b = mk(), mk()
    ta = brush_amd.SplatTrainer(a, brush_amd.TrainConfig(warmup_steps=0, max_refine_step=0, deferred_sh_adam=True))
    tb = brush_amd.SplatTrainer(b, brush_amd.TrainConfig(warmup_steps=0, max_refine_step=0, deferred_sh_adam=False))
    saved, R.DETERMINISTIC = R.DETERMINISTIC, True
    try:
        for i in range(6):
            la, _, aux = ta.step(a, cams[i % 3], gt)
            lb, _, _ = tb.step(b, cams[i % 3], gt)
            assert float(la) == float(lb), (i, float(la), float(lb))
        behind = int((ta._lazy_bufs[0] < ta.opt_time).sum())
        assert behind > n // 2, behind
        ta.sync(a)
        for name in ("means", "log_scales", "rotation", "raw_opacity", "sh_coeffs"):
            assert torch.equal(getattr(a, name).detach(), getattr(b, name).detach()), name
        assert torch.equal(ta.moment1, tb.moment1) and torch.equal(ta.moment2, tb.moment2)
    finally:
        R.DETERMINISTIC = saved


@pytest.mark.parametrize("n,deg,deferred", [(3000, 2, False), (3001, 3, False), (1026, 0, False), (4096, 3, True), (4000, 1, True)])
def test_fused_and_separate_optimizer_paths_give_the_same_bits(dev, n, deg, deferred):
    """One arithmetic for every optimizer entry point (adam_stepped, no FMA contraction): with bitwise reproducible
    gradients (deterministic mode) brush_render_backward + brush_adam_step, brush_render_backward_adam and the
    deferred-SH form of the latter leave the same bits in every parameter and moment."""
    import torch

    import brush_amd
    from brush_amd import render as R

    cloud = H.synthetic_cloud(n, deg, seed=13, mean_mult=0.0005)
    cloud["log_scales"] = cloud["log_scales"] - 3.0
    w, h = 128, 80
    c = H.reference_test_camera(w, h)
    cam = brush_amd.Camera(c["position"], c["rotation_xyzw"], c["fov_x"], c["fov_y"], c["center_uv"])
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    mk = lambda: brush_amd.Splats(t(cloud["means"]), t(cloud["sh"]), t(cloud["quats"] * 1.7), t(cloud["raw_opac"]),
                                  t(cloud["log_scales"]))
    torch.manual_seed(5)
    gt = torch.rand((h, w, 3), device=dev)
    a, b = mk(), mk()
    ta = brush_amd.SplatTrainer(a, brush_amd.TrainConfig(warmup_steps=0, max_refine_step=0, deferred_sh_adam=deferred))
    tb = brush_amd.SplatTrainer(b, brush_amd.TrainConfig(warmup_steps=0, max_refine_step=0, deferred_sh_adam=False))
    tb.fused_backward = False
    saved, R.DETERMINISTIC = R.DETERMINISTIC, True
    try:
        for i in range(4):
            la, _, _ = ta.step(a, cam, gt)
            lb, _, _ = tb.step(b, cam, gt)
            assert float(la) == float(lb), (i, float(la), float(lb))
        assert (ta._lazy is not None) == deferred
        ta.sync(a)
        for name in ("means", "log_scales", "rotation", "raw_opacity", "sh_coeffs"):
            assert torch.equal(getattr(a, name).detach(), getattr(b, name).detach()), name
        assert torch.equal(ta.moment1, tb.moment1) and torch.equal(ta.moment2, tb.moment2)
        assert torch.equal(ta.xy_grad_counts, tb.xy_grad_counts) and torch.equal(ta.grad_2d_accum, tb.grad_2d_accum)
    finally:
        R.DETERMINISTIC = saved
