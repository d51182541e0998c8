"""Per-view data parallelism over RCCL/xGMI (a build extension — the reference is
single-device, batch fixed to 1: crates/brush-train/src/train.rs:216-219).

One process per GPU, splat parameters replicated, rank r renders view r of the batch.  The loss of
the reference is a mean over the stacked batch (train.rs:239-268), so B-view data parallelism is
the mean of per-view gradients: each rank scales its upstream gradient by 1/B and the step needs the
SUM over views on every rank.  Two forms:

  * `ViewExchange` (default): all-gather of 64-byte per-visible-splat records + one deterministic
    reduction kernel (dense sum, or straight into Adam) — sized for xGMI, bit-identical on all ranks;
  * `allreduce_param_grads`: ONE dense all-reduce(sum) over the contiguous parameter-gradient prefix
    `[v_means|v_scales|v_quats|v_opac|v_sh]` of the gradient block (what the north-star describes).
    xGMI is point-to-point (7 links/GPU): at N = 1 M, SH degree 3 that block is 247 MB, i.e.
    2*(7/8)*247 MB ≈ 432 MB per GPU through a ring; RCCL picks the algorithm, nothing here assumes a switch.

The screen-space statistics the trainer keeps for densification (train.rs:284-316) are per view: their
norms and visibility counts travel inside the records (or, with the dense form, in one small second message).
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


# --------------------------------------------------------------------------------------------
# Exchange of per-view gradient records (the default N>1 path)
#
# One view touches ~10 % of the splats and its SH gradient row is rank one,
# v_sh[g] = Y(dir_view(g)) (x) v_rgb[g] (gather_grads.wgsl:186-222), so the view's whole parameter
# gradient is described by 16 floats per VISIBLE splat (include/brush_hip.h):
#     gid | v_means(3) | v_scales(3) | v_quats(4) | v_opac | v_rgb(3) | |v_xy * (w/2, h/2)|      (64 bytes)
# instead of 52+12C bytes per splat (244 at SH degree 3).  brush_render_backward_records writes them
# straight from the compositing backward (no dense arrays), every rank all-gathers the records of all
# views (RCCL all_gather, padded to the largest view), and ONE kernel over global splat ids sums the
# <= W records of each splat in view order and either stores the dense sum (brush_reduce_view_records)
# or feeds it straight into the Adam update (brush_reduce_view_records_adam).  No float atomics: every
# rank computes the same bits from the same bytes, so the replicated parameters cannot drift apart
# (what an unordered atomic expansion or a per-rank "own view exact, others approximate" scheme allows).
# Bytes on xGMI per rank: (W-1) * 6.6 MB at N = 1 M, SH degree 3, vs ~2*(W-1)/W*247 MB for the dense all-reduce.
#
# Sizing without stalling the GPU: the per-view counts are known as soon as the FORWARD has projected the
# splats.  `begin()` (called right after the forward is enqueued) all-gathers [num_visible | camera term] and
# starts their copy to the host; the backward is enqueued with the records buffer's current capacity; by the
# time the host asks for the counts (`counts()`), that copy finished long ago (the GPU is still busy with the
# loss and the backward), so the exact-size all-gather is enqueued without the GPU ever idling.
# --------------------------------------------------------------------------------------------
_REC = 16  # floats per record


def sh_basis_torch(degree: int, d: torch.Tensor) -> torch.Tensor:
    """Sloan basis of project_visible.wgsl:51-147 / gather_grads.wgsl:17-112 for unit dirs [M,3] -> [M,C]."""
    x, y, z = d[:, 0], d[:, 1], d[:, 2]
    Y = [torch.full_like(x, 0.2820947917738781)]
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


def records_from_dense_torch(grads: dict, aux: RenderAux, n: int, img_size, rows: int) -> torch.Tensor:
    """Plain-torch restatement of the record layout (test reference; CPU gloo test): builds the [rows,16]
    records of a view from its DENSE gradients `grads` (v_means, v_scales, v_quats, v_opac, v_sh, v_xy):
    v_rgb = v_sh[g,0,:] / Y0."""
    dev = grads["v_means"].device
    V = min(int(aux.num_visible.reshape(-1)[0]), rows)
    gid = aux.global_from_compact_gid[:V].long()
    rec = torch.zeros((rows, _REC), dtype=torch.float32, device=dev)
    rec[:V, 0] = gid.to(torch.int32).view(torch.float32)
    rec[:V, 1:4] = grads["v_means"][gid]
    rec[:V, 4:7] = grads["v_scales"][gid]
    rec[:V, 7:11] = grads["v_quats"][gid]
    rec[:V, 11] = grads["v_opac"][gid]
    rec[:V, 12:15] = grads["v_sh"][gid, 0, :] / 0.2820947917738781
    w, h = float(img_size[0]), float(img_size[1])
    vxy = grads["v_xy"][gid]
    rec[:V, 15] = torch.sqrt((vxy[:, 0] * (w / 2.0)) ** 2 + (vxy[:, 1] * (h / 2.0)) ** 2)
    return rec


def reduce_view_records_torch(recs, view_rows: torch.Tensor, campos: torch.Tensor, means: torch.Tensor,
                              n: int, ncoef: int) -> dict:
    """Plain-torch restatement of brush_reduce_view_records: dense sum over views, accumulated in view order
    (every rank that runs this on the same bytes gets the same bits).  `recs`: [W, rows, 16] (padded form) or a
    list of W per-view [rows_v, 16] tensors (packed form, what ViewExchange.gather returns by default)."""
    W = len(recs)
    rows = None if isinstance(recs, (list, tuple)) else recs.shape[1]
    dev = recs[0].device
    degree = int(round(ncoef ** 0.5)) - 1
    out = {"v_means": torch.zeros((n, 3), device=dev), "v_scales": torch.zeros((n, 3), device=dev),
           "v_quats": torch.zeros((n, 4), device=dev), "v_opac": torch.zeros((n,), device=dev),
           "v_sh": torch.zeros((n, ncoef, 3), device=dev), "xy_norm": torch.zeros((n,), device=dev),
           "views_seen": torch.zeros((n,), device=dev)}
    for v in range(W):
        cnt = min(int(view_rows[v]), recs[v].shape[0] if rows is None else rows)
        r = recs[v][:cnt]
        gid = r[:, 0].contiguous().view(torch.int32).long()
        keep = (gid >= 0) & (gid < n)
        r, gid = r[keep], gid[keep]
        out["v_means"][gid] += r[:, 1:4]      # a gid appears at most once per view: plain indexed add
        out["v_scales"][gid] += r[:, 4:7]
        out["v_quats"][gid] += r[:, 7:11]
        out["v_opac"][gid] += r[:, 11]
        out["xy_norm"][gid] += r[:, 15]
        out["views_seen"][gid] += 1.0
        d = means[gid] - campos[v][None, :]
        d = d / torch.sqrt(torch.sum(d * d, dim=1, keepdim=True))
        Y = sh_basis_torch(degree, d)
        out["v_sh"][gid] += Y[:, :, None] * r[:, None, 12:15]
    return out


def _padded_rows(count: int) -> int:
    return max(256, -(-int(count) // 256) * 256)


class ViewExchange:
    """One view per rank: sizes, all-gathers and reduces the per-view gradient records of a step.

        xchg = ViewExchange(n, ncoef, device)
        ... enqueue forward ...
        xchg.begin(aux)                               # counts + camera terms start travelling
        ... enqueue loss ...
        xchg.backward_records(u, aux, params..., out, v_out)   # compositing backward -> this view's records
        xchg.gather()                                 # exact-size all-gather (host learns the counts here)
        grads = xchg.reduce_dense(means)   or   xchg.reduce_adam(cfg, params..., moments...)
    """

    def __init__(self, n: int, ncoef: int, device, group: Optional[dist.ProcessGroup] = None, packed: bool = False):
        self.n, self.ncoef, self.device, self.group = int(n), int(ncoef), torch.device(device), group
        # packed=False (default): every view padded to the largest (W x max rows), ONE equal-size
        # all_gather_into_tensor — the native RCCL collective.  packed=True: exactly num_visible records per view
        # (sum over views, ~2 % fewer bytes at 8 views) moved by one broadcast per view into a flat buffer that the
        # reduction reads through per-view row offsets; the same code path on every backend, so what the gloo tests pin is
        # what RCCL runs.  Neither form has run on RCCL hardware yet: bench.py --gpus N times both.
        self.packed = bool(packed)
        self.host_wait_s = 0.0     # time the host spent waiting for the per-view counts (counts())
        self.host_waits = 0        # ... and how often the counts were not there yet when asked for
        self._offsets_dev = None
        self._offsets_pinned = None
        self.degree = int(round(ncoef ** 0.5)) - 1
        live = dist.is_available() and dist.is_initialized()
        self.world = dist.get_world_size(group) if live else 1
        self.rank = dist.get_rank(group) if live else 0
        self.capacity = 0          # rows of the local records buffer
        self.local = None          # [capacity, 16]
        self.gathered = None       # [world * capacity * 16] backing store of the all-gather result
        self.index = None          # brush_view_index_size bytes
        self.metas = torch.empty((self.world, 4), dtype=torch.int32, device=self.device)
        self._host = None
        self._event = None
        self._counts_host = None
        self._rows = 0
        self.regrown = 0           # how often a view outgrew the buffer after the backward was enqueued

    # -- sizing ------------------------------------------------------------------------------
    def begin(self, aux: RenderAux):
        """After the forward is enqueued: all-gather [num_visible | viewmat[3].xyz bits] of every view and start
        the copy of the result to the host (SURVEY 2b-1 for the camera term)."""
        meta = torch.cat([aux.num_visible.reshape(-1)[:1].to(torch.int32), aux.uniforms_buffer[12:15].to(torch.int32)])
        if self.world > 1:
            dist.all_gather_into_tensor(self.metas.view(-1), meta, group=self.group)
        else:
            self.metas.view(-1).copy_(meta)
        self._counts_host = None
        if self.device.type == "cuda":
            if self._host is None:
                self._host = torch.empty(self.world, dtype=torch.int32, pin_memory=True)
            self._host.copy_(self.metas[:, 0], non_blocking=True)
            self._event = torch.cuda.Event()
            self._event.record()

    def counts(self):
        """Per-view visible counts on the host (waits for the copy `begin` started; normally long finished)."""
        if self._counts_host is None:
            if self.device.type == "cuda":
                if not self._event.query():  # forward + count all-gather still running: wait, and account for it
                    import time

                    t0 = time.perf_counter()
                    self._event.synchronize()
                    self.host_wait_s += time.perf_counter() - t0
                    self.host_waits += 1
                self._counts_host = [int(x) for x in self._host.tolist()]
            else:
                self._counts_host = [int(x) for x in self.metas[:, 0].tolist()]
        return self._counts_host

    def _ensure_capacity(self, rows: int):
        if rows <= self.capacity:
            return
        self.capacity = _padded_rows(rows + rows // 4)  # 25 % headroom: growth does not re-run the backward every step
        self.local = torch.empty((self.capacity, _REC), dtype=torch.float32, device=self.device)
        self.gathered = torch.empty(self.world * self.capacity * _REC, dtype=torch.float32, device=self.device)

    # -- this view's records -----------------------------------------------------------------
    def backward_records(self, u, aux: RenderAux, means, log_scales, quats, raw_opacity, out_img, v_out):
        """brush_render_backward_records into the local buffer.  The first call (no capacity yet) waits for the
        counts; later calls enqueue at once with the current capacity and `gather` re-runs the (rare) view that
        outgrew it."""
        import ctypes as C

        from . import _lib

        assert means.is_cuda, "brush_amd has no CPU path: tensors must live on the GPU"
        if self.capacity == 0:
            self._ensure_capacity(max(self.counts()))
        self._bwd_args = (u, aux, means, log_scales, quats, raw_opacity, out_img, v_out)
        l = _lib.lib()
        n = means.shape[0]
        w, h = int(u.img_size[0]), int(u.img_size[1])
        nbytes = C.c_size_t()
        _lib.check(l.brush_bwd_workspace_size_flags(n, w, h, int(u.sh_degree), int(aux.max_intersects), int(aux.flags),
                                                    C.byref(nbytes)), "brush_bwd_workspace_size_flags")
        self._ws, s = aux.backward_workspace(nbytes.value, self.device)
        with torch.cuda.device(self.device):
            _lib.check(l.brush_render_backward_records(C.byref(u), C.byref(s), means.data_ptr(), log_scales.data_ptr(),
                                                       quats.data_ptr(), raw_opacity.data_ptr(), n, out_img.data_ptr(),
                                                       v_out.contiguous().data_ptr(), self.local.data_ptr(), self.capacity,
                                                       self._ws.data_ptr(), nbytes.value,
                                                       torch.cuda.current_stream().cuda_stream),
                       "brush_render_backward_records")

    def set_local_records(self, rec: torch.Tensor):
        """Test hook (CPU / torch restatement): use `rec` [rows,16] as this view's records."""
        self._ensure_capacity(rec.shape[0])
        self.local[:rec.shape[0]].copy_(rec)

    # -- exchange ------------------------------------------------------------------------------
    def _gather_packed(self, counts):
        """Exact sizes: view v's count[v] records land at row offset sum(count[:v]) of the flat buffer, one broadcast
        per view on every backend (torch lowers an uneven all_gather on NCCL/RCCL to the same W broadcasts; issuing them
        here keeps one code path that the gloo tests exercise)."""
        offs = [0]
        for c in counts[:-1]:
            offs.append(offs[-1] + int(c))
        total = offs[-1] + int(counts[-1])
        flat = self.gathered[:max(total, 1) * _REC]
        slices = [flat[offs[r] * _REC:(offs[r] + int(counts[r])) * _REC] for r in range(self.world)]
        mine = self.local[:int(counts[self.rank])].reshape(-1)
        if self.world > 1:
            slices[self.rank].copy_(mine)
            for r in range(self.world):
                if int(counts[r]) > 0:
                    dist.broadcast(slices[r], src=dist.get_global_rank(self.group, r) if self.group is not None else r,
                                   group=self.group)
        else:
            slices[0].copy_(mine)
        if self.device.type == "cuda":
            if self._offsets_pinned is None:
                self._offsets_pinned = torch.empty(self.world, dtype=torch.int32, pin_memory=True)
                self._offsets_dev = torch.empty(self.world, dtype=torch.int32, device=self.device)
            self._offsets_pinned.copy_(torch.tensor(offs, dtype=torch.int32))
            self._offsets_dev.copy_(self._offsets_pinned, non_blocking=True)
        else:
            self._offsets_dev = torch.tensor(offs, dtype=torch.int32)
        self._rows = total
        return [s_.view(-1, _REC) for s_ in slices]

    def gather(self):
        """All-gather of this step's records.  packed (default): exactly count[v] rows per view, returned as a list of
        W [count_v, 16] views of one flat buffer; padded: padded(max count) rows per view, returned as [W, rows, 16]."""
        counts = self.counts()
        need = max(counts) if counts else 0
        if need > self.capacity:  # a view grew by more than the headroom since the buffer was sized: redo its records
            self.regrown += 1
            self._ensure_capacity(need)
            if getattr(self, "_bwd_args", None) is not None:
                self.backward_records(*self._bwd_args)
        if self.packed:
            return self._gather_packed(counts)
        rows = min(_padded_rows(need), self.capacity)
        self._rows = rows
        out = self.gathered[:self.world * rows * _REC]
        part = self.local[:rows].reshape(-1)
        if self.world > 1:
            dist.all_gather_into_tensor(out, part, group=self.group)
        else:
            out.copy_(part)
        return out.view(self.world, rows, _REC)

    def _reduce_common(self, n: int):
        import ctypes as C

        from . import _lib

        nbytes = C.c_size_t()
        _lib.check(_lib.lib().brush_view_index_size(n, self.world, C.byref(nbytes)), "brush_view_index_size")
        if self.index is None or n != self.n or self.index.numel() < nbytes.value:
            # The splat count follows the PARAMETERS of the call, not the count this object was built with: refinement
            # (train.rs:395-579) clones, splits and prunes between steps, and the index is laid out [view][n].
            # all-ones = "no row": the reduction clears what it consumes, so the buffer never holds stale entries
            self.n = int(n)
            self.index = torch.full((nbytes.value,), 0xFF, dtype=torch.uint8, device=self.device)
        if self.packed:
            recs = self.gathered[:max(self._rows, 1) * _REC]
        else:
            recs = self.gathered[:self.world * self._rows * _REC]
        view_rows = self.metas[:, 0].contiguous()
        campos = self.metas[:, 1:4].contiguous().view(torch.float32)
        return recs, view_rows, campos, nbytes.value

    def _offsets_ptr(self):
        return self._offsets_dev.data_ptr() if self.packed else None

    def reduce_dense(self, means: torch.Tensor, block: Optional[torch.Tensor] = None):
        """Sum over views into the dense gradient block (brush_reduce_view_records).  Returns (grads, block);
        v_xy of the block is left untouched (per-view statistic)."""
        from . import _lib

        n, ncoef = int(means.shape[0]), self.ncoef
        layout, total = grad_block_layout(n, ncoef)
        if block is None:
            block = torch.empty(max(total, 1), dtype=torch.float32, device=self.device)
        recs, view_rows, campos, ibytes = self._reduce_common(n)
        with torch.cuda.device(self.device):
            _lib.check(_lib.lib().brush_reduce_view_records(
                recs.data_ptr(), self.world, self._rows, view_rows.data_ptr(), self._offsets_ptr(), campos.data_ptr(),
                means.data_ptr(), n, self.degree, _seg_ptr(block, layout, "v_means"), _seg_ptr(block, layout, "v_scales"),
                _seg_ptr(block, layout, "v_quats"), _seg_ptr(block, layout, "v_sh"), _seg_ptr(block, layout, "v_opac"),
                self.index.data_ptr(), ibytes, torch.cuda.current_stream().cuda_stream), "brush_reduce_view_records")
        self._keep = (view_rows, campos)  # alive until the kernel has run

        def seg(name, shape):
            off, sz = layout[name]
            return block[off:off + sz].view(shape)

        grads = {"v_means": seg("v_means", (n, 3)), "v_scales": seg("v_scales", (n, 3)), "v_quats": seg("v_quats", (n, 4)),
                 "v_opac": seg("v_opac", (n,)), "v_sh": seg("v_sh", (n, ncoef, 3))}
        return grads, block

    def reduce_adam(self, cfg, img_size, means, log_scales, rotation, raw_opacity, sh, moment1, moment2,
                    next_quats_fed=None, grad_2d_accum=None, xy_grad_counts=None):
        """Sum over views straight into the Adam update of every parameter (brush_reduce_view_records_adam)."""
        import ctypes as C

        from . import _lib

        n = int(means.shape[0])
        assert moment1.numel() == n * (11 + 3 * self.ncoef) == moment2.numel(), "moments are laid out for another splat count"
        recs, view_rows, campos, ibytes = self._reduce_common(n)
        ptr = lambda t: None if t is None else t.data_ptr()
        with torch.cuda.device(self.device):
            _lib.check(_lib.lib().brush_reduce_view_records_adam(
                recs.data_ptr(), self.world, self._rows, view_rows.data_ptr(), self._offsets_ptr(), campos.data_ptr(),
                C.byref(cfg),
                int(img_size[0]), int(img_size[1]), means.data_ptr(), log_scales.data_ptr(), rotation.data_ptr(),
                raw_opacity.data_ptr(), sh.data_ptr(), n, self.degree, moment1.data_ptr(), moment2.data_ptr(),
                ptr(next_quats_fed), ptr(grad_2d_accum), ptr(xy_grad_counts), self.index.data_ptr(), ibytes,
                torch.cuda.current_stream().cuda_stream), "brush_reduce_view_records_adam")
        self._keep = (view_rows, campos)


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
