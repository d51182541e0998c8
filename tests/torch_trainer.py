"""Plain-PyTorch restatement of the reference's training iteration — TEST INFRASTRUCTURE: the checker
for brush_amd.train's fused HIP loss / Adam / statistics kernels (parity unpinned: the reference
holds no fixtures for its trainer, so this follows the text of train.rs / ssim.rs / burn's Adam).

The reference's `SplatTrainer::step`
(crates/brush-train/src/train.rs:211-393) minus refinement/densification: render, loss
= L1*(1-w) - SSIM*w (train.rs:243-268, ssim.rs:42-101), backward, screen-space gradient
statistics (train.rs:284-316), five Adam steps with eps 1e-15 in the reference's order and the
higher-order-SH learning-rate lerp (train.rs:318-359).  Nothing here is on the hot path's C ABI.
"""
from __future__ import annotations

import math

import torch
import torch.nn.functional as F

from brush_amd.camera import Camera
from brush_amd.dist import allreduce_densification_stats, densification_stats
from brush_amd.gaussian_splats import Splats
from brush_amd.train import TrainConfig


class Ssim:
    """ssim.rs:1-103: 11x11 Gaussian window (sigma 1.5), grouped conv2d, padding = ceil(window/2)."""

    def __init__(self, window_size: int, channels: int, device):
        ext = window_size // 2
        g = torch.tensor([math.exp(-((x - ext) ** 2) / (2.0 * 1.5 ** 2)) for x in range(window_size)],
                         dtype=torch.float32, device=device)
        g = g / g.sum()
        # The reference convolves with the 2-D window outer(g, g) (ssim.rs:36-40) and notes a
        # separable version as a TODO (ssim.rs:17-32); the two 1-D passes below are the same linear
        # operator (same zero padding) at 2/11 of the multiply-adds.
        self.wv = g.reshape(1, 1, window_size, 1).repeat(channels, 1, 1, 1)
        self.wh = g.reshape(1, 1, 1, window_size).repeat(channels, 1, 1, 1)
        self.channels = channels
        self.padding = -(-window_size // 2)  # div_ceil, as the reference (ssim.rs:49)

    def _blur(self, x):
        x = F.conv2d(x, self.wv, None, stride=1, padding=(self.padding, 0), groups=self.channels)
        return F.conv2d(x, self.wh, None, stride=1, padding=(0, self.padding), groups=self.channels)

    def ssim(self, img1: torch.Tensor, img2: torch.Tensor) -> torch.Tensor:
        a = img1.permute(0, 3, 1, 2)
        b = img2.permute(0, 3, 1, 2)
        mu_x, mu_y = self._blur(a), self._blur(b)
        mu_xx, mu_yy, mu_xy = mu_x * mu_x, mu_y * mu_y, mu_x * mu_y
        s_xx = (self._blur(a * a) - mu_xx).clamp_min(0.0)
        s_yy = (self._blur(b * b) - mu_yy).clamp_min(0.0)
        s_xy = self._blur(a * b) - mu_xy
        c1, c2 = 0.01 ** 2, 0.03 ** 2
        m = ((mu_xy * 2.0 + c1) * (s_xy * 2.0 + c2)) / ((mu_xx + mu_yy + c1) * (s_xx + s_yy + c2))
        return m.mean()


class TorchSplatTrainer:
    def __init__(self, splats: Splats, config: TrainConfig | None = None):
        self.config = config or TrainConfig()
        dev = splats.means.device
        self.iter = 0
        self.ssim = Ssim(self.config.ssim_window_size, 3, dev)
        n = splats.num_splats()
        self.grad_2d_accum = torch.zeros(n, device=dev)
        self.xy_grad_counts = torch.zeros(n, device=dev)
        # AdamConfig::new().with_epsilon(1e-15) (train.rs:184); burn 0.16 Adam::step restated below
        self.state = {}

    def _adam(self, p: torch.Tensor, lr: float):
        b1, b2, eps = 0.9, 0.999, 1e-15
        m, v, t = self.state.get(id(p), (torch.zeros_like(p), torch.zeros_like(p), 0))
        g = p.grad
        m = m * b1 + g * (1.0 - b1)
        v = v * b2 + (g * g) * (1.0 - b2)
        t += 1
        delta = (m / (1.0 - b1 ** t)) / ((v / (1.0 - b2 ** t)).sqrt() + eps)
        self.state[id(p)] = (m, v, t)
        with torch.no_grad():
            p -= delta * lr

    def _lr_mean(self, scene_extent: float) -> float:
        c = self.config
        gamma = c.lr_mean_decay ** (1.0 / c.total_steps)
        return c.lr_mean * gamma ** self.iter * scene_extent

    def step(self, splats: Splats, camera: Camera, gt_image: torch.Tensor, scene_extent: float = 1.0,
             batch_views: int = 1, grad_sync=None):
        """One reference training iteration on one view (batch size is 1 in the reference,
        train.rs:216-219).  With view-sharded data parallelism call it on each rank with
        `batch_views` = world size; gradients are then averaged by the caller's all-reduce."""
        c = self.config
        h, w = gt_image.shape[0], gt_image.shape[1]
        for p in (splats.means, splats.raw_opacity, splats.sh_coeffs, splats.rotation, splats.log_scales):
            p.grad = None
        splats.xys_dummy.grad = None
        pred, aux = splats.render(camera, (w, h), False)
        pred_rgb = pred[..., :3]
        pred_cmp = pred if gt_image.shape[-1] == 4 else pred_rgb
        loss = (pred_cmp - gt_image).abs().mean()
        if c.ssim_weight > 0.0:
            ssim = self.ssim.ssim(pred_rgb[None], gt_image[None, ..., :3])
            loss = loss * (1.0 - c.ssim_weight) - ssim * c.ssim_weight
        (loss / batch_views).backward()
        if grad_sync is not None:  # view-sharded data parallelism: sum the per-view gradients
            grad_sync([splats.means.grad, splats.log_scales.grad, splats.rotation.grad, splats.raw_opacity.grad,
                       splats.sh_coeffs.grad])

        if self.iter > c.warmup_steps:  # housekeeping, train.rs:284-316
            stats = densification_stats(splats.xys_dummy.grad, aux, (w, h))
            allreduce_densification_stats(stats)
            self.grad_2d_accum += stats[0]
            self.xy_grad_counts += stats[1]

        self._adam(splats.means, self._lr_mean(scene_extent))
        self._adam(splats.raw_opacity, c.lr_opac)
        old_coeffs = splats.sh_coeffs.detach().clone()
        self._adam(splats.sh_coeffs, c.lr_coeffs_dc)
        if splats.sh_coeffs.shape[1] > 1:  # SH-rest learning rate = lr / 20 via lerp (train.rs:336-351)
            a = 1.0 / c.lr_coeffs_sh_scale
            with torch.no_grad():
                splats.sh_coeffs[:, 1:] = old_coeffs[:, 1:] * (1.0 - a) + splats.sh_coeffs[:, 1:] * a
        self._adam(splats.rotation, c.lr_rotation)
        self._adam(splats.log_scales, c.lr_scale)
        self.iter += 1
        return loss.detach(), pred.detach(), aux
