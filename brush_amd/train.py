"""Training iteration around the op — what the metric's train iters/s measures.

Mirror of the reference's `SplatTrainer::step` (crates/brush-train/src/train.rs:211-393) and
`refine_splats` (train.rs:395-579; periodic, plain torch tensor ops): render, loss = L1*(1-w) - SSIM*w (train.rs:243-268, ssim.rs:42-101),
backward, screen-space gradient statistics (train.rs:284-316), Adam with eps 1e-15 on the five
parameter groups and the higher-order-SH learning-rate lerp (train.rs:318-359).

The reference builds the loss and the optimizer from Burn tensor ops; here they are the fused HIP
entry points of include/brush_hip.h (brush_l1_ssim_loss, brush_adam_step, brush_refine_stats) called
straight on the op's forward/backward, so one iteration is ≈40 kernel launches and no autograd graph;
with a single view the optimizer step runs inside the backward's last kernel (brush_render_backward_adam).
"""
from __future__ import annotations

import ctypes as C
import math
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
    # refinement (train.rs:25-53)
    refine_every: int = 100
    max_refine_step: int = 15000
    reset_alpha_value: float = 0.004
    cull_alpha_thresh: float = 0.005
    cull_scale_thresh: float = 5.0
    reset_alpha_every_refine: int = 30
    densify_grad_thresh: float = 0.0002
    densify_size_thresh: float = 0.005
    seed: int = 42
    # Build extension (not in the reference): deferred Adam of the SH block (include/brush_hip.h: BrushLazySh).  Same
    # parameter values as the eager optimizer, bit for bit; SplatTrainer.sync() brings `splats.sh_coeffs` up to date
    # for readers that do not go through the trainer (Splats.render / to_ply call it themselves).
    deferred_sh_adam: bool = True


@dataclass
class RefineStats:
    """train.rs:95-101"""
    num_split: int
    num_cloned: int
    num_transparent_pruned: int
    num_scale_pruned: int


def quaternion_vec_multiply(q: torch.Tensor, v: torch.Tensor) -> torch.Tensor:
    """train.rs:140-177, term by term (quaternions as (w, x, y, z))."""
    qw, qx, qy, qz = q[:, 0:1], q[:, 1:2], q[:, 2:3], q[:, 3:4]
    vx, vy, vz = v[:, 0:1], v[:, 1:2], v[:, 2:3]
    t1 = qw * vx + qy * vz - qz * vy
    t2 = qw * vy - qx * vz + qz * vx
    t3 = qw * vz + qx * vy - qy * vx
    t4 = qx * vx + qy * vy + qz * vz
    rx = vx + (qw * t1 + qx * t4 - qy * t3 + qz * t2) * 2.0
    ry = vy + (qw * t2 - qx * t3 + qy * t4 + qz * t1) * 2.0
    rz = vz + (qw * t3 + qx * t2 - qy * t1 + qz * t4) * 2.0
    return torch.cat([rx, ry, rz], dim=1)


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
        self.opt_time = 0  # Adam's per-parameter step count (reset with the optimizer at refinement)
        self.fused_backward = True  # single view: brush_render_backward_adam instead of backward + brush_adam_step
        # rotation/|rotation| left by the previous fused backward, valid only for the very Parameter object the
        # trainer updated (identity + storage + autograd version); see invalidate_cached_rotation().
        self._norm_rot, self._norm_rot_key, self._norm_rot_owner = None, None, None
        self.last_refine: Optional[RefineStats] = None
        self.rng = torch.Generator(device=dev)
        self.rng.manual_seed(self.config.seed)
        # deferred Adam of the SH block: per-splat optimizer time of the stored block, table of per-step constants
        self._lazy: Optional[_lib.BrushLazySh] = None
        self._lazy_bufs = None      # (sh_time, table) tensors the struct points into
        self._lazy_pending = False  # some block may be behind opt_time
        splats.lazy_sh_owner = self

    def invalidate_cached_rotation(self):
        """Call after writing `splats.rotation` behind autograd's back (`.data.copy_(ckpt)`, an external
        kernel): such writes do not bump the tensor version the cache key watches.  In-place ops on the
        Parameter itself, replacing the Parameter and refine_splats are detected without it."""
        self._norm_rot, self._norm_rot_key, self._norm_rot_owner = None, None, None

    LAZY_TABLE_ROWS = 2048

    def _lazy_state(self, splats: Splats, n: int, ncoef: int) -> Optional[_lib.BrushLazySh]:
        """The BrushLazySh of this step (optimizer time self.opt_time), or None when the SH block is stepped eagerly:
        rows of whole 16-byte chunks only (SH degree 1 or 3), n a multiple of 4."""
        c = self.config
        if not c.deferred_sh_adam or (3 * ncoef) % 4 != 0 or n % 4 != 0 or n == 0:
            return None
        dev = splats.means.device
        if self._lazy is not None and self.opt_time + 1 > self._lazy.base + self._lazy.capacity:
            self.sync(splats)   # the table ends here: bring every block to opt_time, then start a new table
            self._lazy = None
        if self._lazy is None:
            import numpy as np
            rows = np.empty((self.LAZY_TABLE_ROWS, 4), dtype=np.float32)
            _lib.check(_lib.lib().brush_lazy_sh_fill_table(0.9, 0.999, c.lr_coeffs_dc, 1.0 / c.lr_coeffs_sh_scale,
                                                           self.opt_time, self.LAZY_TABLE_ROWS, rows.ctypes.data),
                       "brush_lazy_sh_fill_table")
            table = torch.from_numpy(rows).to(dev)
            sh_time = torch.full((n,), self.opt_time, dtype=torch.int32, device=dev)
            z = _lib.BrushLazySh()
            z.table, z.base, z.capacity = table.data_ptr(), self.opt_time, self.LAZY_TABLE_ROWS
            z.sh_time = sh_time.data_ptr()
            z.beta1, z.beta2, z.epsilon = 0.9, 0.999, 1e-15
            self._lazy, self._lazy_bufs = z, (sh_time, table)
        z = self._lazy
        z.now = self.opt_time
        z.sh_moment1 = self.moment1.data_ptr() + 4 * 11 * n
        z.sh_moment2 = self.moment2.data_ptr() + 4 * 11 * n
        return z

    @torch.no_grad()
    def sync(self, splats: Splats):
        """Applies every pending SH step (brush_lazy_sh_flush): afterwards splats.sh_coeffs and the moments are what the
        eager optimizer holds.  No-op when nothing is pending."""
        if self._lazy is None or not self._lazy_pending:
            return
        sh = splats.sh_coeffs.detach()
        n, ncoef = sh.shape[0], sh.shape[1]
        z = self._lazy
        z.now = self.opt_time
        with torch.cuda.device(sh.device):
            _lib.check(_lib.lib().brush_lazy_sh_flush(C.byref(z), sh.data_ptr(), n, R.sh_degree_from_coeffs(ncoef),
                                                      torch.cuda.current_stream(sh.device).cuda_stream),
                       "brush_lazy_sh_flush")
        self._lazy_pending = False

    def _reset(self, n: int, ncoef: int, dev):
        """reset_stats + `self.optim = self.opt_config.init()` (train.rs:201-204,559-563)."""
        self._lazy, self._lazy_bufs, self._lazy_pending = None, None, False
        self.grad_2d_accum = torch.zeros(n, device=dev)
        self.xy_grad_counts = torch.zeros(n, device=dev)
        self.moment1 = torch.zeros(n * (11 + 3 * ncoef), device=dev)
        self.moment2 = torch.zeros(n * (11 + 3 * ncoef), device=dev)
        self.opt_time = 0

    @torch.no_grad()
    def refine_splats(self, splats: Splats, pre_step: dict) -> RefineStats:
        """train.rs:395-579.  `splats` holds the post-step parameters and is rebuilt in place
        (clone / split / prune / opacity reset); `pre_step` holds the parameters before the optimizer
        step (clones and split positions are taken from those).  As in the reference, the shrunken
        scale and the re-sampled mean of a split *source* are computed on temporaries that are dropped
        (train.rs:402,500-526 vs 528): only the appended splats change."""
        c = self.config
        dev = splats.means.device
        post = {"means": splats.means.detach(), "rotation": splats.rotation.detach(), "sh": splats.sh_coeffs.detach(),
                "opac": splats.raw_opacity.detach(), "scales": splats.log_scales.detach()}
        grads = self.grad_2d_accum / self.xy_grad_counts.clamp_min(1.0)
        big_grad = grads >= c.densify_grad_thresh
        small = post["scales"].exp().max(dim=1).values < c.densify_size_thresh
        app = {k: [] for k in post}
        clone_inds = torch.nonzero(small & big_grad).squeeze(1)
        if clone_inds.numel() > 0:
            for k in post:
                app[k].append(pre_step[k][clone_inds])
        split_inds = torch.nonzero(~small & big_grad).squeeze(1)
        ns = int(split_inds.numel())
        if ns > 0:
            cur_rots = post["rotation"][split_inds]
            cur_scale = post["scales"][split_inds].exp()
            app["rotation"].append(cur_rots)
            app["sh"].append(post["sh"][split_inds])
            app["opac"].append(post["opac"][split_inds])
            app["scales"].append((cur_scale / 1.6).log())
            cur_means = pre_step["means"][split_inds]
            # first sample: the source's re-sampled mean, dropped by the reference (kept for the RNG stream)
            torch.randn((ns, 3), generator=self.rng, device=dev)
            samples_new = quaternion_vec_multiply(cur_rots, torch.randn((ns, 3), generator=self.rng, device=dev) * 0.5
                                                  * cur_scale)
            app["means"].append(cur_means + samples_new)
        new = {k: (torch.cat([post[k]] + app[k], 0) if app[k] else post[k]) for k in post}
        start = new["means"].shape[0]

        def prune(mask):
            keep = torch.nonzero(~mask).squeeze(1)
            if keep.numel() < mask.numel():
                for k in new:
                    new[k] = new[k][keep]

        prune(torch.sigmoid(new["opac"]) < c.cull_alpha_thresh)
        alpha_pruned = start - new["means"].shape[0]
        prune(new["scales"].exp().max(dim=1).values > c.cull_scale_thresh)
        scale_pruned = start - new["means"].shape[0]  # cumulative, as train.rs:551
        if (self.iter // c.refine_every) % c.reset_alpha_every_refine == 0:
            new["opac"] = torch.zeros_like(new["opac"]) + math.log(c.reset_alpha_value / (1.0 - c.reset_alpha_value))
        splats.means = torch.nn.Parameter(new["means"].contiguous())
        splats.rotation = torch.nn.Parameter(new["rotation"].contiguous())
        splats.sh_coeffs = torch.nn.Parameter(new["sh"].contiguous())
        splats.raw_opacity = torch.nn.Parameter(new["opac"].contiguous())
        splats.log_scales = torch.nn.Parameter(new["scales"].contiguous())
        n = splats.means.shape[0]
        splats.xys_dummy = torch.zeros((n, 2), dtype=torch.float32, device=dev, requires_grad=True)
        self._reset(n, int(splats.sh_coeffs.shape[1]), dev)
        return RefineStats(ns, int(clone_inds.numel()), alpha_pruned, scale_pruned)

    def _lr_mean(self, scene_extent: float) -> float:
        c = self.config
        gamma = c.lr_mean_decay ** (1.0 / c.total_steps)
        return c.lr_mean * gamma ** self.iter * scene_extent

    def step(self, splats: Splats, camera: Camera, gt_image: torch.Tensor, scene_extent: float = 1.0,
             batch_views: int = 1, grad_sync: Optional[Callable] = None, exchange=None):
        """One reference training iteration on one view (batch size is 1 in the reference,
        train.rs:216-219).  With view-sharded data parallelism call it on each rank with
        `batch_views` = world size and either `exchange` = a brush_amd.dist.ViewExchange (per-view gradient
        records all-gathered, summed per splat in view order and fed straight into Adam: every rank applies
        the same bits) or `grad_sync(block, aux)` summing the dense gradient block over views
        (brush_amd.dist.allreduce_param_grads)."""
        c = self.config
        h, w = int(gt_image.shape[0]), int(gt_image.shape[1])
        n, ncoef = splats.num_splats(), int(splats.sh_coeffs.shape[1])
        if self.moment1.numel() != n * (11 + 3 * ncoef):
            raise ValueError("the number of splats changed outside refine_splats: build a new SplatTrainer")
        means, log_scales, quats = splats.means.detach(), splats.log_scales.detach(), splats.rotation.detach()
        sh, raw_opac = splats.sh_coeffs.detach(), splats.raw_opacity.detach()
        for t in (means, log_scales, quats, sh, raw_opac):
            assert t.is_contiguous() and t.dtype == torch.float32
        l = _lib.lib()
        stream = torch.cuda.current_stream(means.device).cuda_stream
        # Splats::render feeds rotation / |rotation| (gaussian_splats.rs:174-175).  The fused backward of the
        # previous step already wrote it for the updated rotation; recompute when anyone else touched it.
        key = (quats.data_ptr(), n, splats.rotation._version)
        if self._norm_rot is not None and self._norm_rot_key == key and self._norm_rot_owner is splats.rotation:
            norm_rot = self._norm_rot
        else:
            norm_rot = torch.empty_like(quats)
            with torch.cuda.device(means.device):
                _lib.check(l.brush_normalize_quats(quats.data_ptr(), norm_rot.data_ptr(), n, stream), "brush_normalize_quats")
        self.invalidate_cached_rotation()
        # the optimizer runs inside a kernel that sees which splats the step touches: the single-view fused backward, or
        # the data-parallel reduction of the views' records (both take BrushAdamConfig.lazy_sh)
        fused = grad_sync is None and (exchange is not None or self.fused_backward)
        lazy = self._lazy_state(splats, n, ncoef) if fused else None
        if lazy is None:
            self.sync(splats)  # this step reads / steps every SH block: nothing may stay pending
            self._lazy = None
        pred, aux, u = R._forward_impl(camera, (w, h), means, log_scales, norm_rot, sh, raw_opac, False, None,
                                       lazy_sh=lazy)
        if exchange is not None:
            exchange.begin(aux)  # the per-view counts start travelling while the loss and the backward run
        loss, v_pred = l1_ssim_loss(pred, gt_image, c.ssim_weight, c.ssim_window_size, 1.0 / batch_views)
        do_refine = self.iter < c.max_refine_step and self.iter >= c.warmup_steps and self.iter % c.refine_every == 1
        pre_step = None
        if do_refine:  # refinement clones / splits the parameters *before* the optimizer step (train.rs:361-372)
            self.sync(splats)  # (the forward above has read the pending state; the clones need the eager values)
            pre_step = {"means": means.clone(), "rotation": quats.clone(), "sh": sh.clone(), "opac": raw_opac.clone(),
                        "scales": log_scales.clone()}
        cfg = _lib.BrushAdamConfig(self._lr_mean(scene_extent), c.lr_scale, c.lr_rotation, c.lr_opac,
                                   c.lr_coeffs_dc, 1.0 / c.lr_coeffs_sh_scale, 0.9, 0.999, 1e-15, self.opt_time + 1, 1,
                                   # v_pred carries 1/batch_views: keep the densification statistic at the magnitude the
                                   # reference's threshold was tuned for (batch 1, train.rs:284-316)
                                   float(batch_views))
        if lazy is not None:
            cfg.lazy_sh = C.pointer(lazy)
        want_stats = self.iter > c.warmup_steps  # housekeeping, train.rs:284-316
        with torch.cuda.device(means.device):
            if exchange is not None:
                # view-sharded data parallelism: records of this view -> all-gather -> per-splat sum -> Adam
                exchange.backward_records(u, aux, means, log_scales, norm_rot, raw_opac, pred, v_pred)
                exchange.gather()
                next_rot = torch.empty_like(quats)
                exchange.reduce_adam(cfg, (w, h), means, log_scales, quats, raw_opac, sh, self.moment1, self.moment2,
                                     next_rot, self.grad_2d_accum if want_stats else None,
                                     self.xy_grad_counts if want_stats else None)
                self._norm_rot, self._norm_rot_key = next_rot, (quats.data_ptr(), n, splats.rotation._version)
                self._norm_rot_owner = splats.rotation
            elif grad_sync is None and self.fused_backward:
                # single view: gradients go straight through the optimizer inside the backward kernel
                nbytes = C.c_size_t()
                _lib.check(l.brush_bwd_workspace_size_flags(n, w, h, int(u.sh_degree), int(aux.max_intersects),
                                                            int(aux.flags), C.byref(nbytes)),
                           "brush_bwd_workspace_size_flags")
                ws, s_aux = aux.backward_workspace(nbytes.value, means.device)
                v_xy = torch.empty((max(n, 1), 2), dtype=torch.float32, device=means.device)
                next_rot = torch.empty_like(quats)
                _lib.check(l.brush_render_backward_adam(C.byref(u), C.byref(s_aux), C.byref(cfg), means.data_ptr(),
                                                        log_scales.data_ptr(), norm_rot.data_ptr(), quats.data_ptr(),
                                                        raw_opac.data_ptr(), sh.data_ptr(), n, pred.data_ptr(),
                                                        v_pred.data_ptr(), v_xy.data_ptr(), self.moment1.data_ptr(),
                                                        self.moment2.data_ptr(), next_rot.data_ptr(),
                                                        self.grad_2d_accum.data_ptr() if want_stats else None,
                                                        self.xy_grad_counts.data_ptr() if want_stats else None,
                                                        ws.data_ptr(), nbytes.value, stream),
                           "brush_render_backward_adam")
                self._norm_rot, self._norm_rot_key = next_rot, (quats.data_ptr(), n, splats.rotation._version)
                self._norm_rot_owner = splats.rotation
            else:
                grads, block = R._backward_impl(u, aux, means, log_scales, norm_rot, raw_opac, ncoef, pred, v_pred)
                if grad_sync is not None:  # view-sharded data parallelism: sum the per-view gradients
                    grad_sync(block, aux)
                if want_stats:
                    v_xy = grads["v_xy"]
                    if torch.distributed.is_available() and torch.distributed.is_initialized() and batch_views > 1:
                        from .dist import densification_stats
                        stats = densification_stats(v_xy, aux, (w, h))
                        allreduce_densification_stats(stats)
                        self.grad_2d_accum += stats[0] * float(batch_views)  # undo the 1/batch of the loss scale
                        self.xy_grad_counts += stats[1]
                    else:
                        s_aux = aux._as_struct()
                        _lib.check(l.brush_refine_stats(C.byref(s_aux), v_xy.data_ptr(), n, w, h,
                                                        self.grad_2d_accum.data_ptr(), self.xy_grad_counts.data_ptr(),
                                                        stream), "brush_refine_stats")
                _lib.check(l.brush_adam_step(C.byref(cfg), n, R.sh_degree_from_coeffs(ncoef), means.data_ptr(),
                                             log_scales.data_ptr(), quats.data_ptr(), raw_opac.data_ptr(), sh.data_ptr(),
                                             grads["v_means"].data_ptr(), grads["v_scales"].data_ptr(),
                                             grads["v_quats"].data_ptr(), grads["v_opac"].data_ptr(),
                                             grads["v_sh"].data_ptr(), self.moment1.data_ptr(), self.moment2.data_ptr(),
                                             stream),
                           "brush_adam_step")
        self.opt_time += 1
        if lazy is not None:
            self._lazy_pending = True
        if do_refine:
            self.sync(splats)  # refinement reads the post-step coefficients of every splat
        self.last_refine = self.refine_splats(splats, pre_step) if do_refine else None
        self.iter += 1
        return loss, pred, aux
