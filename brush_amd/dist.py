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

import os
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


# --------------------------------------------------------------------------------------------
# Compact gradient exchange
#
# One view touches ~10 % of the splats and its SH gradient row is rank one,
# v_sh[g] = Y(dir_view(g)) (x) v_rgb[g] (gather_grads.wgsl:186-222), so the view's whole parameter
# gradient is described by 15 numbers per VISIBLE splat:
#     gid | v_means(3) | v_scales(3) | v_quats(4) | v_opac | v_sh[g,0,:](3)      (60 bytes)
# instead of 52+12C bytes per splat (244 at SH degree 3).  Every rank all-gathers the records of
# all views (RCCL all_gather, padded to the largest view) and expands + sums them locally into the
# dense block; Y is recomputed from the replicated means and each view's camera term.  The result
# on every rank is the same dense sum an all-reduce would give (up to f32 summation order and one
# extra rounding in v_rgb = v_sh0 / Y0); the bytes on xGMI drop from ~2*(W-1)/W*247 MB to
# (W-1)*6 MB per rank at N = 1 M, SH degree 3, which is what makes 2..8 GPUs scale when a view
# takes only ~0.5 ms to render.
# --------------------------------------------------------------------------------------------
_REC = 16  # floats per record (15 used, 64-byte rows)
_SH_C0 = 0.2820947917738781


def sh_basis_torch(degree: int, d: torch.Tensor) -> torch.Tensor:
    """Sloan basis of project_visible.wgsl:51-147 / gather_grads.wgsl:17-112 for unit dirs [M,3] -> [M,C]."""
    x, y, z = d[:, 0], d[:, 1], d[:, 2]
    Y = [torch.full_like(x, _SH_C0)]
    if degree >= 1:
        a = 0.48860251190292
        Y += [-a * y, a * z, -a * x]
    if degree >= 2:
        z2 = z * z
        f0b = -1.092548430592079 * z
        f1a = 0.5462742152960395
        c1 = x * x - y * y
        s1 = 2.0 * x * y
        p6 = 0.9461746957575601 * z2 - 0.3153915652525201
        Y += [f1a * s1, f0b * y, p6, f0b * x, f1a * c1]
    if degree >= 3:
        f0c = -2.285228997322329 * z2 + 0.4570457994644658
        f1b = 1.445305721320277 * z
        f2a = -0.5900435899266435
        c2 = x * c1 - y * s1
        s2 = x * s1 + y * c1
        p12 = z * (1.865881662950577 * z2 - 1.119528997770346)
        Y += [f2a * s2, f1b * s1, f0c * y, p12, f0c * x, f1b * c1, f2a * c2]
    if degree >= 4:
        f0d = z * (-4.683325804901025 * z2 + 2.007139630671868)
        f1c = 3.31161143515146 * z2 - 0.47308734787878
        f2b = -1.770130769779931 * z
        f3a = 0.6258357354491763
        c3 = x * c2 - y * s2
        s3 = x * s2 + y * c2
        Y += [f3a * s3, f2b * s2, f1c * s1, f0d * y, 1.984313483298443 * z * p12 - 1.006230589874905 * p6, f0d * x,
              f1c * c1, f2b * c2, f3a * c3]
    return torch.stack(Y, dim=1)


def _seg_ptr(block, layout, name):
    return block.data_ptr() + layout[name][0] * 4


def pack_view_records(block: torch.Tensor, aux: RenderAux, n: int, ncoef: int, rows: int) -> torch.Tensor:
    """[rows, 16] f32 records of this view's visible splats through brush_pack_view_records (rows
    beyond num_visible are left uninitialised; consumers use the per-view row counts)."""
    import ctypes as C

    from . import _lib

    assert block.is_cuda, "brush_amd has no CPU path: tensors must live on the GPU"
    layout, _ = grad_block_layout(n, ncoef)
    rec = torch.empty((rows, _REC), dtype=torch.float32, device=block.device)
    s = aux._as_struct()
    degree = int(round(ncoef ** 0.5)) - 1
    with torch.cuda.device(block.device):
        stream = torch.cuda.current_stream().cuda_stream
        _lib.check(_lib.lib().brush_pack_view_records(C.byref(s), n, degree, _seg_ptr(block, layout, "v_means"),
                                                      _seg_ptr(block, layout, "v_scales"),
                                                      _seg_ptr(block, layout, "v_quats"), _seg_ptr(block, layout, "v_opac"),
                                                      _seg_ptr(block, layout, "v_sh"), rec.data_ptr(), rows, stream),
                   "brush_pack_view_records")
    return rec


def pack_view_records_torch(block: torch.Tensor, aux: RenderAux, n: int, ncoef: int, rows: int) -> torch.Tensor:
    """Plain-torch restatement of brush_pack_view_records (test reference; CPU gloo test)."""
    layout, _ = grad_block_layout(n, ncoef)
    dev = block.device
    idx = torch.arange(rows, device=dev)
    valid = idx < aux.num_visible.to(idx.dtype)
    gid = torch.where(valid, aux.global_from_compact_gid[:n].long()[idx.clamp(max=max(n - 1, 0))], torch.zeros_like(idx))

    def seg(name, width):
        off, sz = layout[name]
        return block[off:off + sz].view(n, width)

    rec = torch.zeros((rows, _REC), dtype=torch.float32, device=dev)
    rec[:, 0] = gid.to(torch.int32).view(torch.float32)
    rec[:, 1:4] = seg("v_means", 3)[gid]
    rec[:, 4:7] = seg("v_scales", 3)[gid]
    rec[:, 7:11] = seg("v_quats", 4)[gid]
    rec[:, 11] = seg("v_opac", 1)[gid, 0]
    rec[:, 12:15] = seg("v_sh", ncoef * 3)[gid, 0:3]
    rec[:, 15] = 1.0
    return rec


def expand_view_records(recs: torch.Tensor, view_rows: torch.Tensor, campos: torch.Tensor, means: torch.Tensor,
                        block: torch.Tensor, n: int, ncoef: int, own_view: Optional[int] = None) -> torch.Tensor:
    """Sums the records of the views (recs [W, rows, 16], valid rows per view `view_rows` int32 [W],
    camera terms [W, 3] = viewmat[3].xyz of each view) into the parameter prefix of `block` with the
    HIP kernel brush_expand_view_records.  own_view=None: the prefix is overwritten by the sum of
    all W views; own_view=r: `block` already holds view r's dense gradients and the other views
    are added on top.  Device tensors only (no CPU path)."""
    from . import _lib

    assert block.is_cuda and recs.is_cuda and means.is_cuda, "brush_amd has no CPU path: tensors must live on the GPU"
    layout, _ = grad_block_layout(n, ncoef)
    W, rows, _ = recs.shape
    recs, campos, means = recs.contiguous(), campos.contiguous(), means.contiguous()
    view_rows = view_rows.to(torch.int32).contiguous()
    degree = int(round(ncoef ** 0.5)) - 1
    with torch.cuda.device(block.device):
        stream = torch.cuda.current_stream().cuda_stream
        skip = 0xFFFFFFFF if own_view is None else int(own_view)
        _lib.check(_lib.lib().brush_expand_view_records(recs.data_ptr(), W * rows, rows, view_rows.data_ptr(),
                                                        campos.data_ptr(), means.data_ptr(), n, degree, skip,
                                                        _seg_ptr(block, layout, "v_means"), _seg_ptr(block, layout, "v_scales"),
                                                        _seg_ptr(block, layout, "v_quats"), _seg_ptr(block, layout, "v_opac"),
                                                        _seg_ptr(block, layout, "v_sh"), stream),
                   "brush_expand_view_records")
    return block


def expand_view_records_torch(recs: torch.Tensor, view_rows: torch.Tensor, campos: torch.Tensor, means: torch.Tensor,
                              block: torch.Tensor, n: int, ncoef: int, own_view: Optional[int] = None) -> torch.Tensor:
    """Plain-torch restatement of brush_expand_view_records (test reference; CPU gloo test)."""
    layout, _ = grad_block_layout(n, ncoef)
    W, rows, _ = recs.shape
    flat = recs.reshape(W * rows, _REC)
    ridx = torch.arange(W * rows, device=flat.device)
    view = ridx // rows
    keep = (ridx - view * rows) < view_rows.to(ridx.dtype)[view]
    if own_view is not None:
        keep = keep & (view != own_view)
    gid = flat[:, 0].contiguous().view(torch.int32).long()
    gid = torch.where(keep, gid, torch.zeros_like(gid))
    flat = torch.where(keep[:, None], flat, torch.zeros_like(flat))  # rows beyond a view's count are garbage
    w = keep.to(torch.float32)[:, None]
    cam = campos.repeat_interleave(rows, dim=0)
    d = means[gid] - cam
    d = d / torch.sqrt(torch.sum(d * d, dim=1, keepdim=True))
    degree = int(round(ncoef ** 0.5)) - 1
    Y = sh_basis_torch(degree, d)                        # [M, C]
    rgb = flat[:, 12:15] * (w / _SH_C0)                  # v_rgb = v_sh0 / Y0 (0 for padding rows)
    Y = torch.where(keep[:, None], Y, torch.zeros_like(Y))  # padding rows may have NaN dirs
    if own_view is None:
        block[:param_grad_floats(n, ncoef)].zero_()

    def seg(name, width):
        off, sz = layout[name]
        return block[off:off + sz].view(n, width)

    seg("v_means", 3).index_add_(0, gid, flat[:, 1:4] * w)
    seg("v_scales", 3).index_add_(0, gid, flat[:, 4:7] * w)
    seg("v_quats", 4).index_add_(0, gid, flat[:, 7:11] * w)
    seg("v_opac", 1).index_add_(0, gid, flat[:, 11:12] * w)
    seg("v_sh", ncoef * 3).index_add_(0, gid, (Y[:, :, None] * rgb[:, None, :]).reshape(-1, ncoef * 3))
    return block


def _padded_rows(count: int, chunks: int = 1) -> int:
    q = 256 * chunks
    return max(q, -(-int(count) // q) * q)


# rows used by the previous exchange of the same (group, cloud size), with headroom: lets the next
# exchange be enqueued before the host has seen the new counts (they are all-gathered, hence the
# same on every rank, so every rank sizes its collectives identically).
_ROWS_HINT = {}


def allreduce_param_grads_compact(block: torch.Tensor, aux: RenderAux, means: torch.Tensor, n: int, ncoef: int,
                                  group: Optional[dist.ProcessGroup] = None, pack=None, expand=None) -> torch.Tensor:
    """Same result as allreduce_param_grads (sum over views, on every rank) through an all-gather of
    compact per-view records.  One 16-byte all-gather carries the per-view counts that size the padded
    record exchange.  On the GPU the exchange is enqueued optimistically with the previous step's
    size (+12.5 %) while the counts travel to the host asynchronously; they are checked before the
    expansion and, should a view have outgrown the hint, pack + all-gather are simply repeated at the
    right size — the host never leaves the GPU idle waiting for a number."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return block
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    dev = block.device
    pack = pack or pack_view_records
    # one small message per rank: [num_visible | viewmat[3].xyz bits] (SURVEY §2b-1 for the camera term)
    meta = torch.cat([aux.num_visible.reshape(-1)[:1].to(torch.int32), aux.uniforms_buffer[12:15].to(torch.int32)])
    metas = torch.empty((world, 4), dtype=torch.int32, device=dev)
    dist.all_gather_into_tensor(metas.view(-1), meta, group=group)
    counts = metas[:, 0].contiguous()
    cams = metas[:, 1:4].contiguous().view(torch.float32)

    # From 4 views on, the records travel in two halves: the expansion of the first half (float atomics,
    # L2-bound) overlaps the all-gather of the second (xGMI-bound).
    chunks = int(os.environ.get("BRUSH_EXCHANGE_CHUNKS", "2" if world >= 4 else "1"))

    def exchange(rows):
        rec = pack(block, aux, n, ncoef, rows)
        per = rows // chunks
        outs, works = [], []
        for k in range(chunks):
            out = torch.empty((world, per, _REC), dtype=torch.float32, device=dev)
            part = rec[k * per:(k + 1) * per].reshape(-1)
            works.append(dist.all_gather_into_tensor(out.view(-1), part, group=group, async_op=chunks > 1))
            outs.append(out)
        return outs, works, per

    key = (id(group), world, n, ncoef, str(dev))
    hint = _ROWS_HINT.get(key)
    got = None
    if block.is_cuda and hint is not None:
        host = torch.empty(world, dtype=torch.int32, pin_memory=True)
        host.copy_(counts, non_blocking=True)
        seen = torch.cuda.Event()
        seen.record()
        got, rows = exchange(hint), hint  # enqueued while the counts are still on their way
        seen.synchronize()
        max_count = int(host.max())
        if max_count > rows:  # a view outgrew the hint: redo at the right size (same decision on every rank)
            if chunks > 1:
                for wk in got[1]:
                    wk.wait()
            got = None
    else:
        max_count = int(counts.max().item())
    if got is None:
        rows = _padded_rows(max_count, chunks)
        got = exchange(rows)
    _ROWS_HINT[key] = _padded_rows(max_count + max_count // 8, chunks)
    outs, works, per = got
    expand = expand or expand_view_records
    for k in range(chunks):
        if chunks > 1:
            works[k].wait()  # the current stream waits for this half only
        part_rows = counts if chunks == 1 else (counts - k * per).clamp(0, per)
        expand(outs[k], part_rows, cams, means, block, n, ncoef, rank)
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
