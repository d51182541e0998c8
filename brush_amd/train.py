"""Training iteration around the op — what the metric's train iters/s measures.

Mirror of the reference's `SplatTrainer::step` (crates/brush-train/src/train.rs:211-393) minus
refinement/densification: render, loss = L1*(1-w) - SSIM*w (train.rs:243-268, ssim.rs:42-101),
backward, screen-space gradient statistics (train.rs:284-316), Adam with eps 1e-15 on the five
parameter groups and the higher-order-SH learning-rate lerp (train.rs:318-359).

The reference builds the loss and the optimizer from Burn tensor ops; here they are the fused HIP
entry points of include/brush_hip.h (brush_l1_ssim_loss, brush_adam_step, brush_refine_stats) called
straight on the op's forward/backward, so one iteration is ≈40 kernel launches and no autograd graph.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass
from typing import Callable, Optional

import torch

from . import _lib
from . import render as R
from .camera import Camera
from .dist import allreduce_densification_stats
from .gaussian_splats import Splats


@dataclass
class TrainConfig:
    """Defaults of train.rs:19-87 and the UI's learning-rate schedule (panels/load_data.rs:41-71)."""
    warmup_steps: int = 500
    ssim_weight: float = 0.2
    ssim_window_size: int = 11
    lr_mean: float = 1.6e-4
    lr_mean_decay: float = 0.01   # total decay factor over `total_steps`
    total_steps: int = 30000
    lr_coeffs_dc: float = 0.004
    lr_coeffs_sh_scale: float = 20.0
    lr_opac: float = 0.05
    lr_scale: float = 0.01
    lr_rotation: float = 0.002


def l1_ssim_loss(pred: torch.Tensor, gt: torch.Tensor, ssim_weight: float, window: int = 11,
                 grad_scale: float = 1.0):
    """(loss [1] device tensor, d loss / d pred [h,w,4]) through brush_l1_ssim_loss."""
    assert pred.is_cuda and gt.is_cuda, "brush_amd has no CPU path: tensors must live on the GPU"
    h, w = int(pred.shape[0]), int(pred.shape[1])
    if tuple(pred.shape) != (h, w, 4) or tuple(gt.shape[:2]) != (h, w) or gt.shape[2] not in (3, 4):
        raise ValueError(f"pred must be [h,w,4] and gt [h,w,3|4], got {tuple(pred.shape)} / {tuple(gt.shape)}")
    pred, gt = pred.contiguous().float(), gt.contiguous().float()
    l = _lib.lib()
    nbytes = C.c_size_t()
    _lib.check(l.brush_loss_workspace_size(w, h, C.byref(nbytes)), "brush_loss_workspace_size")
    ws = torch.empty(nbytes.value, dtype=torch.uint8, device=pred.device)
    loss = torch.empty(1, dtype=torch.float32, device=pred.device)
    v_pred = torch.empty_like(pred)
    with torch.cuda.device(pred.device):
        _lib.check(l.brush_l1_ssim_loss(pred.data_ptr(), gt.data_ptr(), w, h, int(gt.shape[2]), float(ssim_weight),
                                        int(window), float(grad_scale), loss.data_ptr(), v_pred.data_ptr(),
                                        ws.data_ptr(), nbytes.value, torch.cuda.current_stream().cuda_stream),
                   "brush_l1_ssim_loss")
    return loss, v_pred


class SplatTrainer:
    def __init__(self, splats: Splats, config: TrainConfig | None = None):
        self.config = config or TrainConfig()
        dev = splats.means.device
        assert dev.type == "cuda", "brush_amd has no CPU path: the splats must live on the GPU"
        self.iter = 0
        n = splats.num_splats()
        ncoef = int(splats.sh_coeffs.shape[1])
        self.grad_2d_accum = torch.zeros(n, device=dev)
        self.xy_grad_counts = torch.zeros(n, device=dev)
        # Adam moments of the five groups, [means|log_scales|quats|raw_opac|sh] (AdamConfig of train.rs:184)
        self.moment1 = torch.zeros(n * (11 + 3 * ncoef), device=dev)
        self.moment2 = torch.zeros(n * (11 + 3 * ncoef), device=dev)

    def _lr_mean(self, scene_extent: float) -> float:
        c = self.config
        gamma = c.lr_mean_decay ** (1.0 / c.total_steps)
        return c.lr_mean * gamma ** self.iter * scene_extent

    def step(self, splats: Splats, camera: Camera, gt_image: torch.Tensor, scene_extent: float = 1.0,
             batch_views: int = 1, grad_sync: Optional[Callable] = None):
        """One reference training iteration on one view (batch size is 1 in the reference,
        train.rs:216-219).  With view-sharded data parallelism call it on each rank with
        `batch_views` = world size and `grad_sync(block, aux)` summing the gradient block over views
        (brush_amd.dist.allreduce_param_grads[_compact])."""
        c = self.config
        h, w = int(gt_image.shape[0]), int(gt_image.shape[1])
        n, ncoef = splats.num_splats(), int(splats.sh_coeffs.shape[1])
        if self.moment1.numel() != n * (11 + 3 * ncoef):
            raise ValueError("the number of splats changed: build a new SplatTrainer (refinement is not part of step)")
        means, log_scales, quats = splats.means.detach(), splats.log_scales.detach(), splats.rotation.detach()
        sh, raw_opac = splats.sh_coeffs.detach(), splats.raw_opacity.detach()
        for t in (means, log_scales, quats, sh, raw_opac):
            assert t.is_contiguous() and t.dtype == torch.float32
        l = _lib.lib()
        stream = torch.cuda.current_stream(means.device).cuda_stream
        norm_rot = torch.empty_like(quats)  # Splats::render feeds rotation / |rotation| (gaussian_splats.rs:174-175)
        with torch.cuda.device(means.device):
            _lib.check(l.brush_normalize_quats(quats.data_ptr(), norm_rot.data_ptr(), n, stream), "brush_normalize_quats")
        pred, aux, u = R._forward_impl(camera, (w, h), means, log_scales, norm_rot, sh, raw_opac, False, None)
        loss, v_pred = l1_ssim_loss(pred, gt_image, c.ssim_weight, c.ssim_window_size, 1.0 / batch_views)
        grads, block = R._backward_impl(u, aux, means, log_scales, norm_rot, raw_opac, ncoef, pred, v_pred)
        if grad_sync is not None:  # view-sharded data parallelism: sum the per-view gradients
            grad_sync(block, aux)

        with torch.cuda.device(means.device):
            if self.iter > c.warmup_steps:  # housekeeping, train.rs:284-316
                v_xy = grads["v_xy"]
                if torch.distributed.is_available() and torch.distributed.is_initialized() and batch_views > 1:
                    from .dist import densification_stats
                    stats = densification_stats(v_xy, aux, (w, h))
                    allreduce_densification_stats(stats)
                    self.grad_2d_accum += stats[0]
                    self.xy_grad_counts += stats[1]
                else:
                    s = aux._as_struct()
                    _lib.check(l.brush_refine_stats(C.byref(s), v_xy.data_ptr(), n, w, h, self.grad_2d_accum.data_ptr(),
                                                    self.xy_grad_counts.data_ptr(), stream), "brush_refine_stats")
            cfg = _lib.BrushAdamConfig(self._lr_mean(scene_extent), c.lr_scale, c.lr_rotation, c.lr_opac,
                                       c.lr_coeffs_dc, 1.0 / c.lr_coeffs_sh_scale, 0.9, 0.999, 1e-15, self.iter + 1, 1)
            _lib.check(l.brush_adam_step(C.byref(cfg), n, R.sh_degree_from_coeffs(ncoef), means.data_ptr(),
                                         log_scales.data_ptr(), quats.data_ptr(), raw_opac.data_ptr(), sh.data_ptr(),
                                         grads["v_means"].data_ptr(), grads["v_scales"].data_ptr(),
                                         grads["v_quats"].data_ptr(), grads["v_opac"].data_ptr(),
                                         grads["v_sh"].data_ptr(), self.moment1.data_ptr(), self.moment2.data_ptr(),
                                         stream),
                       "brush_adam_step")
        self.iter += 1
        return loss, pred, aux
