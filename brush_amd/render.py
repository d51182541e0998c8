"""render_splats — host mirror of brush-render's op surface over the HIP C ABI.

Reference interface (paths relative to the reference checkout):
  trait Backend::render_splats         crates/brush-render/src/lib.rs:66-86
  struct RenderAux                     crates/brush-render/src/lib.rs:20-63
  impl Backend for Autodiff<..>        crates/brush-render/src/render.rs:366-463
  impl Backward<_,6> for RenderBackwards  render.rs:465-626

PyTorch is plumbing here: device memory, the current stream and autograd bookkeeping.  Every
kernel is in libbrush_hip.so; there is no eager/CPU fallback.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass
from typing import Optional, Tuple

import numpy as np
import torch

from . import _lib
from .camera import Camera

# When True every buffer the op allocates is filled with 0xCD bytes first, the counterpart of
# the reference's -12345 poison under cfg(test) (crates/brush-kernel/src/lib.rs:134-147).
DEBUG_POISON = False

# Deterministic gradients (BrushAux.flags & AUX_DETERMINISTIC), chosen per call: an explicit `deterministic=` argument
# wins, then this module-level override, then the environment default brush_deterministic() (BRUSH_DETERMINISTIC=1).
DETERMINISTIC: Optional[bool] = None


def deterministic_default() -> bool:
    if DETERMINISTIC is not None:
        return bool(DETERMINISTIC)
    return bool(_lib.lib().brush_deterministic())


def sh_coeffs_for_degree(degree: int) -> int:
    """render.rs:40-42"""
    return (degree + 1) ** 2


def sh_degree_from_coeffs(coeffs_per_channel: int) -> int:
    """render.rs:44-53 (panics on an invalid count)."""
    table = {1: 0, 4: 1, 9: 2, 16: 3, 25: 4}
    if coeffs_per_channel not in table:
        raise ValueError(f"Invalid nr. of sh bases {coeffs_per_channel}")
    return table[coeffs_per_channel]


def _empty(shape, dtype, device):
    t = torch.empty(shape, dtype=dtype, device=device)
    if DEBUG_POISON and t.numel() > 0:
        t.view(torch.uint8).fill_(0xCD)
    return t


@dataclass
class RenderAux:
    """Mirror of RenderAux (lib.rs:20-33); integer tensors are int32 like Burn's."""
    projected_splats: torch.Tensor         # [N,9] f32, first num_visible rows valid
    uniforms_buffer: torch.Tensor          # [28] i32
    num_intersections: torch.Tensor        # [1] i32
    num_visible: torch.Tensor              # [1] i32
    final_index: torch.Tensor              # [h,w] i32
    cum_tiles_hit: torch.Tensor            # [N] i32
    tile_bins: torch.Tensor                # [ty,tx,2] i32
    compact_gid_from_isect: torch.Tensor   # [max_intersects] i32
    global_from_compact_gid: torch.Tensor  # [N] i32
    # build extensions
    compact_from_global_gid: torch.Tensor  # [N] i32 (-1 = not visible)
    overflow: torch.Tensor                 # [1] i32, 1 if intersections were truncated
    max_intersects: int = 0
    # deterministic mode only: pre-sort position of every sorted intersection
    isect_unsorted_pos: Optional[torch.Tensor] = None
    flags: int = 0                         # BrushAux.flags this render ran with (forward and backward agree)
    # default mode: the backward's workspace, allocated at forward time so that the forward's last kernel zeroes the
    # accumulator rows (BrushAux::bwd_accum); `bwd_ws_zeroed` is consumed by the first backward of this render
    bwd_ws: Optional[torch.Tensor] = None
    bwd_ws_zeroed: bool = False
    # deferred Adam of the SH block (a _lib.BrushLazySh the trainer owns; None: the coefficients are current)
    lazy_sh: Optional[object] = None

    @property
    def deterministic(self) -> bool:
        return bool(self.flags & _lib.AUX_DETERMINISTIC)

    def backward_workspace(self, nbytes: int, device):
        """(workspace tensor, BrushAux struct) for ONE backward call of this render: the buffer the forward pre-zeroed
        (with BRUSH_AUX_ACCUM_ZEROED set, first backward only) or a fresh one."""
        s = self._as_struct()
        if self.bwd_ws is not None and self.bwd_ws.numel() >= nbytes:
            ws = self.bwd_ws
            if self.bwd_ws_zeroed:
                s.flags |= _lib.AUX_ACCUM_ZEROED
                self.bwd_ws_zeroed = False  # a second backward of the same forward zero-fills itself
        else:
            ws = torch.empty(max(int(nbytes), 1), dtype=torch.uint8, device=device)
            s.bwd_accum = None
        return ws, s

    def read_num_visible(self) -> int:
        """lib.rs:42-47 (a host readback; not on the hot path)."""
        return int(self.num_visible.item())

    def read_num_intersections(self) -> int:
        """lib.rs:49-54"""
        return int(self.num_intersections.item())

    def read_tile_depth(self) -> torch.Tensor:
        """lib.rs:56-62"""
        return self.tile_bins[..., 1] - self.tile_bins[..., 0]

    def _as_struct(self) -> _lib.BrushAux:
        s = _lib.BrushAux()
        for name in ("projected_splats", "uniforms_buffer", "num_intersections", "num_visible", "final_index",
                     "cum_tiles_hit", "tile_bins", "compact_gid_from_isect", "global_from_compact_gid",
                     "compact_from_global_gid", "overflow"):
            setattr(s, name, getattr(self, name).data_ptr())
        s.max_intersects = int(self.max_intersects)
        s.isect_unsorted_pos = None if self.isect_unsorted_pos is None else self.isect_unsorted_pos.data_ptr()
        s.flags = int(self.flags)
        s.bwd_accum = None if self.bwd_ws is None else self.bwd_ws.data_ptr()
        if self.lazy_sh is not None:
            s.lazy_sh = C.pointer(self.lazy_sh)
        return s


def pack_uniforms(cam: Camera, img_size, sh_degree: int, total_splats: int) -> _lib.BrushUniforms:
    """render.rs:82-116: tile bounds, viewmat (column-major), focal, centre."""
    w, h = int(img_size[0]), int(img_size[1])
    u = _lib.BrushUniforms()
    w2l = cam.world_to_local()  # row-major [r][c]
    u.viewmat[:] = [float(w2l[r][c]) for c in range(4) for r in range(4)]
    f = cam.focal((w, h))
    c = cam.center((w, h))
    u.focal[:] = [float(f[0]), float(f[1])]
    u.pixel_center[:] = [float(c[0]), float(c[1])]
    u.img_size[:] = [w, h]
    u.tile_bounds[:] = [-(-w // _lib.TILE_WIDTH), -(-h // _lib.TILE_WIDTH)]
    u.sh_degree = int(sh_degree)
    u.num_visible = 0
    u.total_splats = int(total_splats)
    u.padding = 0
    return u


def _check_inputs(means, xy_dummy, log_scales, quats, sh_coeffs, raw_opacity):
    """DimCheck of render.rs:74-79 (assert on mismatch) plus dtype/device requirements."""
    n = means.shape[0]
    assert means.dim() == 2 and means.shape[1] == 3, f"means must be [D,3], got {tuple(means.shape)}"
    assert log_scales.shape == (n, 3), f"log_scales must be [D,3], got {tuple(log_scales.shape)}"
    assert quats.shape == (n, 4), f"quats must be [D,4], got {tuple(quats.shape)}"
    assert sh_coeffs.dim() == 3 and sh_coeffs.shape[0] == n and sh_coeffs.shape[2] == 3, \
        f"sh_coeffs must be [D,C,3], got {tuple(sh_coeffs.shape)}"
    assert raw_opacity.shape == (n,), f"raw_opacity must be [D], got {tuple(raw_opacity.shape)}"
    assert xy_dummy is None or xy_dummy.shape == (n, 2)
    for t in (means, log_scales, quats, sh_coeffs, raw_opacity):
        assert t.is_cuda, "brush_amd has no CPU path: tensors must live on the GPU"
        assert t.dtype == torch.float32
    return n


def _forward_impl(cam: Camera, img_size, means, log_scales, quats, sh_coeffs, raw_opacity, render_u32: bool,
                  max_intersects: Optional[int], row_pitch: Optional[int] = None,
                  deterministic: Optional[bool] = None, expect_backward: Optional[bool] = None, lazy_sh=None):
    """expect_backward (default: a float image in default mode): allocate the backward's workspace now and let the
    forward zero its accumulator rows, so that the backward of this render needs no zero-fill launch.
    lazy_sh: a _lib.BrushLazySh (SplatTrainer's deferred Adam of the SH block): colours come from the coefficients
    with their pending optimizer steps replayed; `sh_coeffs` is not written."""
    l = _lib.lib()
    det = deterministic_default() if deterministic is None else bool(deterministic)
    n = means.shape[0]
    w, h = int(img_size[0]), int(img_size[1])
    dev = means.device
    sh_degree = sh_degree_from_coeffs(sh_coeffs.shape[1])
    u = pack_uniforms(cam, (w, h), sh_degree, n)
    tbx, tby = int(u.tile_bounds[0]), int(u.tile_bounds[1])
    cap = int(max_intersects) if max_intersects is not None else int(l.brush_default_max_intersects(n, w, h))
    cap = max(cap, 1)
    i32 = torch.int32
    nn = max(n, 1)
    aux = RenderAux(
        projected_splats=_empty((nn, _lib.PROJECTED_FLOATS), torch.float32, dev),
        uniforms_buffer=_empty((_lib.UNIFORM_WORDS,), i32, dev),
        num_intersections=_empty((1,), i32, dev),
        num_visible=_empty((1,), i32, dev),
        final_index=_empty((h, w), i32, dev),
        cum_tiles_hit=_empty((nn,), i32, dev),
        tile_bins=_empty((tby, tbx, 2), i32, dev),
        compact_gid_from_isect=_empty((cap,), i32, dev),
        global_from_compact_gid=_empty((nn,), i32, dev),
        compact_from_global_gid=_empty((nn,), i32, dev),
        overflow=_empty((1,), i32, dev),
        max_intersects=cap,
        isect_unsorted_pos=_empty((cap,), i32, dev) if det else None,
        flags=_lib.AUX_DETERMINISTIC if det else 0,
        lazy_sh=lazy_sh,
    )
    if row_pitch is not None:
        if not render_u32 or row_pitch < w:
            raise ValueError("row_pitch needs render_u32_buffer=True and row_pitch >= width")
        out = torch.zeros((h, int(row_pitch), 1), dtype=i32, device=dev)  # padding columns stay 0 (burn_texture.rs:21-24)
    else:
        out = _empty((h, w, 1), i32, dev) if render_u32 else _empty((h, w, 4), torch.float32, dev)
    if expect_backward is None:
        expect_backward = not render_u32
    if expect_backward and not det and not render_u32:
        bbytes = C.c_size_t()
        _lib.check(l.brush_bwd_workspace_size_flags(n, w, h, sh_degree, cap, int(aux.flags), C.byref(bbytes)),
                   "brush_bwd_workspace_size_flags")
        aux.bwd_ws = _empty((max(bbytes.value, 1),), torch.uint8, dev)
        aux.bwd_ws_zeroed = True
    nbytes = C.c_size_t()
    _lib.check(l.brush_fwd_workspace_size(n, w, h, sh_degree, cap, C.byref(nbytes)), "brush_fwd_workspace_size")
    ws = _empty((max(nbytes.value, 1),), torch.uint8, dev)
    means, log_scales, quats = means.contiguous(), log_scales.contiguous(), quats.contiguous()
    sh_coeffs, raw_opacity = sh_coeffs.contiguous(), raw_opacity.contiguous()
    s = aux._as_struct()
    with torch.cuda.device(dev):
        stream = torch.cuda.current_stream().cuda_stream
        if row_pitch is not None:
            _lib.check(l.brush_render_forward_rgba8(C.byref(u), means.data_ptr(), log_scales.data_ptr(),
                                                    quats.data_ptr(), sh_coeffs.data_ptr(), raw_opacity.data_ptr(), n,
                                                    out.data_ptr(), int(row_pitch), C.byref(s), ws.data_ptr(),
                                                    nbytes.value, stream), "brush_render_forward_rgba8")
        else:
            _lib.check(l.brush_render_forward(C.byref(u), means.data_ptr(), log_scales.data_ptr(), quats.data_ptr(),
                                              sh_coeffs.data_ptr(), raw_opacity.data_ptr(), n, 1 if render_u32 else 0,
                                              out.data_ptr(), C.byref(s), ws.data_ptr(), nbytes.value, stream),
                       "brush_render_forward")
    return out, aux, u


def grad_block_layout(n: int, ncoef: int):
    """Offsets (in floats) of [v_means | v_scales | v_quats | v_opac | v_sh | v_xy] in one block.

    The first five are the parameter gradients that data-parallel training all-reduces as one
    contiguous message; v_xy (densification statistic) sits at the tail.
    """
    sizes = [("v_means", n * 3), ("v_scales", n * 3), ("v_quats", n * 4), ("v_opac", n), ("v_sh", n * ncoef * 3),
             ("v_xy", n * 2)]
    off, layout = 0, {}
    for name, sz in sizes:
        layout[name] = (off, sz)
        off += (sz + 3) // 4 * 4  # keep every segment 16-byte aligned
    return layout, off


def _backward_impl(u, aux: RenderAux, means, log_scales, quats, raw_opacity, ncoef, out_img, v_out,
                   block: Optional[torch.Tensor] = None):
    l = _lib.lib()
    n = means.shape[0]
    dev = means.device
    w, h = int(u.img_size[0]), int(u.img_size[1])
    layout, total = grad_block_layout(n, ncoef)
    if block is None:
        block = _empty((max(total, 1),), torch.float32, dev)
    assert block.numel() >= total and block.dtype == torch.float32 and block.is_contiguous()

    def seg(name, shape):
        off, sz = layout[name]
        return block[off:off + sz].view(shape)

    g = {
        "v_means": seg("v_means", (n, 3)), "v_scales": seg("v_scales", (n, 3)), "v_quats": seg("v_quats", (n, 4)),
        "v_opac": seg("v_opac", (n,)), "v_sh": seg("v_sh", (n, ncoef, 3)), "v_xy": seg("v_xy", (n, 2)),
    }
    nbytes = C.c_size_t()
    _lib.check(l.brush_bwd_workspace_size_flags(n, w, h, int(u.sh_degree), int(aux.max_intersects), int(aux.flags),
                                                C.byref(nbytes)), "brush_bwd_workspace_size_flags")
    v_out = v_out.contiguous()
    ws, s = aux.backward_workspace(nbytes.value, dev)
    with torch.cuda.device(dev):
        stream = torch.cuda.current_stream().cuda_stream
        _lib.check(l.brush_render_backward(C.byref(u), C.byref(s), means.data_ptr(), log_scales.data_ptr(),
                                           quats.data_ptr(), raw_opacity.data_ptr(), n, out_img.data_ptr(),
                                           v_out.data_ptr(), g["v_means"].data_ptr(), g["v_xy"].data_ptr(),
                                           g["v_scales"].data_ptr(), g["v_quats"].data_ptr(), g["v_sh"].data_ptr(),
                                           g["v_opac"].data_ptr(), ws.data_ptr(), nbytes.value, stream),
                   "brush_render_backward")
    return g, block


class _RenderSplatsFn(torch.autograd.Function):
    """Parents in the order of render.rs:420-427: means, xy_dummy, log_scales, quats, sh, raw_opacity."""

    @staticmethod
    def forward(ctx, means, xy_dummy, log_scales, quats, sh_coeffs, raw_opacity, holder):
        out, aux, u = _forward_impl(holder["cam"], holder["img_size"], means, log_scales, quats, sh_coeffs,
                                    raw_opacity, False, holder["max_intersects"], deterministic=holder["deterministic"])
        holder["aux"] = aux
        ctx.u, ctx.aux, ctx.ncoef = u, aux, sh_coeffs.shape[1]
        ctx.save_for_backward(means, log_scales, quats, raw_opacity, out)
        ctx.mark_non_differentiable()
        return out

    @staticmethod
    def backward(ctx, v_output):
        means, log_scales, quats, raw_opacity, out = ctx.saved_tensors
        g, _ = _backward_impl(ctx.u, ctx.aux, means, log_scales, quats, raw_opacity, ctx.ncoef, out,
                              v_output.to(torch.float32))
        return g["v_means"], g["v_xy"], g["v_scales"], g["v_quats"], g["v_sh"], g["v_opac"], None


def render_splats(cam: Camera, img_size, means: torch.Tensor, xy_grad_dummy: Optional[torch.Tensor],
                  log_scales: torch.Tensor, quats: torch.Tensor, sh_coeffs: torch.Tensor,
                  raw_opacity: torch.Tensor, render_u32_buffer: bool = False,
                  max_intersects: Optional[int] = None,
                  deterministic: Optional[bool] = None) -> Tuple[torch.Tensor, RenderAux]:
    """Backend::render_splats (lib.rs:75-85).

    Returns (img, aux): img is float32 [h,w,4] (rgb, 1-T; no background blend) or, with
    `render_u32_buffer`, int32 [h,w,1] packed RGBA8.  `xy_grad_dummy` only carries the
    screen-space xy gradient (global order, pixel units).  `max_intersects` defaults to the
    reference's min(N*tiles, 128*65535); aux.overflow reports truncation.  `deterministic` (build extension)
    selects bitwise reproducible gradients for this call (default: render.DETERMINISTIC, then BRUSH_DETERMINISTIC).
    """
    _check_inputs(means, xy_grad_dummy, log_scales, quats, sh_coeffs, raw_opacity)
    tracked = torch.is_grad_enabled() and not render_u32_buffer and any(
        t is not None and t.requires_grad for t in (means, xy_grad_dummy, log_scales, quats, sh_coeffs, raw_opacity))
    if not tracked:
        # UnTracked branch (render.rs:453-460): plain forward, no state kept.
        with torch.no_grad():
            out, aux, _ = _forward_impl(cam, img_size, means, log_scales, quats, sh_coeffs, raw_opacity,
                                        render_u32_buffer, max_intersects, deterministic=deterministic,
                                        expect_backward=False)
        return out, aux
    if xy_grad_dummy is None:
        xy_grad_dummy = torch.zeros((means.shape[0], 2), dtype=torch.float32, device=means.device)
    holder = {"cam": cam, "img_size": img_size, "max_intersects": max_intersects, "deterministic": deterministic}
    out = _RenderSplatsFn.apply(means, xy_grad_dummy, log_scales, quats, sh_coeffs, raw_opacity, holder)
    return out, holder["aux"]


def rgba8_row_pitch(width: int) -> int:
    """Pixels per row the viewer's texture upload wants: ceil(width/64)*64 (burn_texture.rs:17-26)."""
    return int(_lib.lib().brush_rgba8_row_pitch(int(width)))


def render_rgba8(cam: Camera, img_size, means, log_scales, quats, sh_coeffs, raw_opacity,
                 row_pitch: Optional[int] = None, max_intersects: Optional[int] = None):
    """Forward-only display path: packed RGBA8 with rows `row_pitch` pixels apart (default
    rgba8_row_pitch(width)), i.e. the padded tensor burn_texture.rs:17-26 builds with a zero-fill
    and a slice_assign, written by the rasterizer directly.  Returns (int32 [h, row_pitch, 1], aux)."""
    _check_inputs(means, None, log_scales, quats, sh_coeffs, raw_opacity)
    pitch = rgba8_row_pitch(img_size[0]) if row_pitch is None else int(row_pitch)
    with torch.no_grad():
        out, aux, _ = _forward_impl(cam, img_size, means, log_scales, quats, sh_coeffs, raw_opacity, True,
                                    max_intersects, row_pitch=pitch)
    return out, aux


def uniforms_to_numpy(aux: RenderAux) -> dict:
    """Decode aux.uniforms_buffer (28 words) into typed fields (debug / test aid)."""
    words = aux.uniforms_buffer.cpu().numpy().astype(np.int32).view(np.uint32)
    f = words.view(np.float32)
    return {
        "viewmat": f[0:16].copy(), "focal": f[16:18].copy(), "img_size": words[18:20].copy(),
        "tile_bounds": words[20:22].copy(), "pixel_center": f[22:24].copy(), "sh_degree": int(words[24]),
        "num_visible": int(words[25]), "total_splats": int(words[26]),
    }
