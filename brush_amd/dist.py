"""Per-view data parallelism over RCCL/xGMI (a build extension — the reference is
single-device, batch fixed to 1: crates/brush-train/src/train.rs:216-219).

One process per GPU, splat parameters replicated, rank r renders view r of the batch.  The loss of
the reference is a mean over the stacked batch (train.rs:239-268), so B-view data parallelism is
the mean of per-view gradients: each rank scales its upstream gradient by 1/B (or averages after
the reduce) and ONE all-reduce(sum) over the contiguous parameter-gradient prefix of the gradient
block `[v_means|v_scales|v_quats|v_opac|v_sh]` (brush_amd.render.grad_block_layout) makes every
rank hold the batch gradient.  The screen-space statistics the trainer keeps for densification
(train.rs:284-316) are per view, so their *norms* and visibility counts are reduced separately in
one small second message.

xGMI is point-to-point (7 links/GPU): at N = 1 M, SH degree 3 the block is 247 MB, i.e.
2*(7/8)*247 MB ≈ 432 MB per GPU through the ring; RCCL picks the algorithm, nothing here assumes
a switch.
"""
from __future__ import annotations

from typing import Optional

import torch
import torch.distributed as dist

from .render import RenderAux, grad_block_layout


def param_grad_floats(n: int, ncoef: int) -> int:
    """Length (floats) of the all-reduced prefix [v_means|v_scales|v_quats|v_opac|v_sh]."""
    layout, _ = grad_block_layout(n, ncoef)
    off, sz = layout["v_sh"]
    return off + sz


def allreduce_param_grads(block: torch.Tensor, n: int, ncoef: int, average: bool = False,
                          group: Optional[dist.ProcessGroup] = None) -> torch.Tensor:
    """In-place sum (or mean) of the parameter-gradient prefix of `block` over all ranks."""
    prefix = block[:param_grad_floats(n, ncoef)]
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(prefix, op=dist.ReduceOp.SUM, group=group)
        if average:
            prefix.div_(dist.get_world_size(group))
    return block


def densification_stats(v_xy: torch.Tensor, aux: RenderAux, img_size) -> torch.Tensor:
    """Per-view statistics of train.rs:284-316 packed as [2, N] f32:
    row 0 = ||v_xy * (w/2, h/2)||, row 1 = 1 for splats visible in this view else 0."""
    w, h = float(img_size[0]), float(img_size[1])
    scale = torch.tensor([w / 2.0, h / 2.0], dtype=v_xy.dtype, device=v_xy.device)
    norm = torch.sqrt(torch.sum((v_xy * scale) ** 2, dim=1))
    n = v_xy.shape[0]
    # The reference scatters `arange(N) < num_visible` through global_from_compact_gid
    # (xy_grad_counts.select_assign, train.rs:306-312); the inverse map of this build gives the
    # same visibility mask without a scatter.
    visible = (aux.compact_from_global_gid[:n] >= 0).to(v_xy.dtype)
    return torch.stack([norm, visible])


def allreduce_densification_stats(stats: torch.Tensor, group: Optional[dist.ProcessGroup] = None) -> torch.Tensor:
    """Sum over views of the per-view norms and visibility counts (one small message)."""
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(stats, op=dist.ReduceOp.SUM, group=group)
    return stats


def shard_views(num_views: int, rank: int, world: int):
    """Views rendered by `rank`: view indices rank, rank+world, ... (independent units, no exchange
    besides the gradient reduce)."""
    return list(range(rank, num_views, world))
