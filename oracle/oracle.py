"""ctypes/numpy front-end of the CPU oracle (oracle/brush_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg.  Nothing under brush_amd/ imports this.

The camera helper restates crates/brush-render/src/camera.rs:28-58 and the uniform packing of
crates/brush-render/src/render.rs:102-116 independently of brush_amd's host mirror.
"""
from __future__ import annotations

import ctypes as C
import math
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "liboracle.so")

TILE_WIDTH = 16


class _Uniforms(C.Structure):
    _fields_ = [
        ("viewmat", C.c_float * 16),
        ("focal", C.c_float * 2),
        ("img_size", C.c_uint32 * 2),
        ("tile_bounds", C.c_uint32 * 2),
        ("pixel_center", C.c_float * 2),
        ("sh_degree", C.c_uint32),
        ("num_visible", C.c_uint32),
        ("total_splats", C.c_uint32),
        ("padding", C.c_uint32),
    ]


class _Aux(C.Structure):
    _fields_ = [
        ("projected_splats", C.c_void_p),
        ("num_intersections", C.c_void_p),
        ("num_visible", C.c_void_p),
        ("final_index", C.c_void_p),
        ("cum_tiles_hit", C.c_void_p),
        ("tile_bins", C.c_void_p),
        ("compact_gid_from_isect", C.c_void_p),
        ("global_from_compact_gid", C.c_void_p),
        ("tile_id_from_isect", C.c_void_p),
        ("flip_risk", C.c_void_p),
        ("max_intersects", C.c_uint32),
    ]


def build(force: bool = False) -> str:
    """Compile liboracle.so with the committed Makefile if it is missing or stale."""
    srcs = [os.path.join(_HERE, f) for f in ("brush_oracle.c", "brush_oracle_f64.c", "brush_oracle.h", "detmath.h",
                                             "Makefile")]
    stale = not os.path.exists(_LIB_PATH) or any(
        os.path.getmtime(s) > os.path.getmtime(_LIB_PATH) for s in srcs
    )
    if force or stale:
        subprocess.check_call(["make", "-C", _HERE, "-B" if force else "-s"], stdout=subprocess.DEVNULL)
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(_LIB_PATH)
        _lib.oracle_det_expf.restype = C.c_float
        _lib.oracle_det_expf.argtypes = [C.c_float]
        _lib.oracle_det_logf.restype = C.c_float
        _lib.oracle_det_logf.argtypes = [C.c_float]
        _lib.oracle_num_threads.restype = C.c_int
        _lib.oracle_render_forward.restype = C.c_int
        _lib.oracle_render_backward.restype = C.c_int
        _lib.oracle_render_backward_ex.restype = C.c_int
        _lib.oracle_render_backward_f64.restype = C.c_int
    return _lib


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def num_threads() -> int:
    return int(lib().oracle_num_threads())


def det_expf(x: float) -> float:
    return float(lib().oracle_det_expf(C.c_float(x)))


def det_logf(x: float) -> float:
    return float(lib().oracle_det_logf(C.c_float(x)))


# ----------------------------------------------------------------------------- camera


def fov_to_focal(fov_rad: float, pixels: int) -> float:
    """crates/brush-render/src/camera.rs:50-52"""
    return 0.5 * float(pixels) / math.tan(fov_rad * 0.5)


def focal_to_fov(focal: float, pixels: int) -> float:
    """crates/brush-render/src/camera.rs:55-57"""
    return 2.0 * math.atan(float(pixels) / (2.0 * focal))


def _quat_to_mat3(q_xyzw):
    x, y, z, w = [float(v) for v in q_xyzw]
    return np.array(
        [
            [1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)],
            [2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)],
            [2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)],
        ],
        dtype=np.float64,
    )


def make_uniforms(position, rotation_xyzw, fov_x, fov_y, center_uv, img_size, sh_degree):
    """Pack RenderUniforms (helpers.wgsl:7-30) the way render.rs:102-116 does.

    position / rotation are the camera's local-to-world translation and (x,y,z,w) quaternion
    (camera.rs:41-47); viewmat is the inverse, stored column-major.
    Returns a dict with the 28 words as typed fields.
    """
    w, h = int(img_size[0]), int(img_size[1])
    l2w = np.eye(4)
    l2w[:3, :3] = _quat_to_mat3(rotation_xyzw)
    l2w[:3, 3] = np.asarray(position, dtype=np.float64)
    w2l = np.linalg.inv(l2w).astype(np.float32)
    return {
        "viewmat": np.ascontiguousarray(w2l.T).reshape(16),  # column-major
        "focal": np.array([fov_to_focal(fov_x, w), fov_to_focal(fov_y, h)], dtype=np.float32),
        "img_size": np.array([w, h], dtype=np.uint32),
        "tile_bounds": np.array([-(-w // TILE_WIDTH), -(-h // TILE_WIDTH)], dtype=np.uint32),
        "pixel_center": np.array([center_uv[0] * w, center_uv[1] * h], dtype=np.float32),
        "sh_degree": int(sh_degree),
    }


def _to_struct(u: dict, n: int) -> _Uniforms:
    s = _Uniforms()
    s.viewmat[:] = [float(v) for v in u["viewmat"]]
    s.focal[:] = [float(v) for v in u["focal"]]
    s.img_size[:] = [int(v) for v in u["img_size"]]
    s.tile_bounds[:] = [int(v) for v in u["tile_bounds"]]
    s.pixel_center[:] = [float(v) for v in u["pixel_center"]]
    s.sh_degree = int(u["sh_degree"])
    s.num_visible = 0
    s.total_splats = int(n)
    s.padding = 0
    return s


def default_max_intersects(n: int, num_tiles: int) -> int:
    """crates/brush-render/src/render.rs:204-206"""
    return max(1, min(n * num_tiles, 128 * 65535))


# ----------------------------------------------------------------------------- primitives


def radix_argsort(keys, vals, n_sort=None, bits=32):
    """brush-sort/src/lib.rs:32-37 semantics on numpy uint32 arrays."""
    keys = np.ascontiguousarray(keys, dtype=np.uint32)
    vals = np.ascontiguousarray(vals, dtype=np.uint32)
    n = len(keys) if n_sort is None else int(n_sort)
    ko = keys.copy()
    vo = vals.copy()
    lib().oracle_radix_argsort(_p(keys), _p(vals), C.c_uint32(n), C.c_uint32(bits), _p(ko), _p(vo))
    return ko, vo


def inclusive_scan(x):
    """brush-prefix-sum/src/lib.rs:17 semantics."""
    x = np.ascontiguousarray(x, dtype=np.uint32)
    out = np.empty_like(x)
    lib().oracle_inclusive_scan(_p(x), C.c_uint32(len(x)), _p(out))
    return out


# ----------------------------------------------------------------------------- render


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def sh_basis_for_means(u: dict, means):
    """Y_k(dir) of gather_grads.wgsl:186-222 for every mean (dir as the shader takes it): [n, (sh_degree + 1)^2] f32."""
    means = _f32(means)
    n = means.shape[0]
    ncoef = (int(u["sh_degree"]) + 1) ** 2
    Y = np.empty((n, ncoef), np.float32)
    us = _to_struct(u, n)
    rc = lib().oracle_sh_basis_for_means(C.byref(us), _p(means), n, _p(Y))
    assert rc == 0
    return Y


def render_forward(u: dict, means, log_scales, quats, sh_coeffs, raw_opac, raster_u32=False,
                   max_intersects=None):
    """Returns (out_img, aux) with aux a dict of numpy arrays mirroring RenderAux."""
    means, log_scales, quats = _f32(means), _f32(log_scales), _f32(quats)
    sh_coeffs, raw_opac = _f32(sh_coeffs), _f32(raw_opac)
    n = means.shape[0]
    w, h = int(u["img_size"][0]), int(u["img_size"][1])
    tbx, tby = int(u["tile_bounds"][0]), int(u["tile_bounds"][1])
    ncoef = (int(u["sh_degree"]) + 1) ** 2
    assert sh_coeffs.shape == (n, ncoef, 3), sh_coeffs.shape
    cap = default_max_intersects(n, tbx * tby) if max_intersects is None else int(max_intersects)

    aux = {
        "projected_splats": np.zeros((max(n, 1), 9), np.float32),
        "num_intersections": np.zeros(1, np.uint32),
        "num_visible": np.zeros(1, np.uint32),
        "final_index": np.zeros((h, w), np.uint32),
        "cum_tiles_hit": np.zeros(max(n, 1), np.uint32),
        "tile_bins": np.zeros((tby, tbx, 2), np.uint32),
        "compact_gid_from_isect": np.zeros(cap, np.uint32),
        "global_from_compact_gid": np.zeros(max(n, 1), np.uint32),
        "tile_id_from_isect": np.zeros(cap, np.uint32),
        "flip_risk": np.zeros((h, w), np.uint8),
    }
    s = _Aux()
    for k in aux:
        setattr(s, k, aux[k].ctypes.data)
    s.max_intersects = cap
    out = np.zeros((h, w), np.uint32) if raster_u32 else np.zeros((h, w, 4), np.float32)
    us = _to_struct(u, n)
    rc = lib().oracle_render_forward(C.byref(us), _p(means), _p(log_scales), _p(quats), _p(sh_coeffs),
                                     _p(raw_opac), C.c_uint32(n), C.c_int(1 if raster_u32 else 0),
                                     _p(out), C.byref(s))
    aux["overflow"] = bool(rc == 1)
    aux["max_intersects"] = cap
    return out, aux


def render_backward(u: dict, aux: dict, means, log_scales, quats, raw_opac, out_img, v_out, f32_sums=False):
    """Returns dict of dense grads (+ compact-order intermediates).  f32_sums: also sum in f32 (pixels in
    tile order, a splat's tiles in ascending intersection order) instead of the order-free f64 sums."""
    means, log_scales, quats, raw_opac = _f32(means), _f32(log_scales), _f32(quats), _f32(raw_opac)
    out_img, v_out = _f32(out_img), _f32(v_out)
    n = means.shape[0]
    ncoef = (int(u["sh_degree"]) + 1) ** 2
    V = int(aux["num_visible"][0])
    g = {
        "v_means": np.zeros((n, 3), np.float32),
        "v_xy": np.zeros((n, 2), np.float32),
        "v_scales": np.zeros((n, 3), np.float32),
        "v_quats": np.zeros((n, 4), np.float32),
        "v_sh": np.zeros((n, ncoef, 3), np.float32),
        "v_opac": np.zeros((n,), np.float32),
        "v_xy_local": np.zeros((max(V, 1), 2), np.float32),
        "v_conics": np.zeros((max(V, 1), 3), np.float32),
        "v_colors": np.zeros((max(V, 1), 4), np.float32),
    }
    s = _Aux()
    for k in ("projected_splats", "num_intersections", "num_visible", "final_index", "cum_tiles_hit",
              "tile_bins", "compact_gid_from_isect", "global_from_compact_gid"):
        setattr(s, k, aux[k].ctypes.data)
    s.max_intersects = int(aux["max_intersects"])
    us = _to_struct(u, n)
    lib().oracle_render_backward_ex(C.byref(us), C.byref(s), _p(means), _p(log_scales), _p(quats),
                                    _p(raw_opac), C.c_uint32(n), _p(out_img), _p(v_out), _p(g["v_means"]),
                                    _p(g["v_xy"]), _p(g["v_scales"]), _p(g["v_quats"]), _p(g["v_sh"]),
                                    _p(g["v_opac"]), _p(g["v_xy_local"]), _p(g["v_conics"]),
                                    _p(g["v_colors"]), C.c_int(1 if f32_sums else 0))
    return g


def rasterize_forward_f64(u: dict, aux: dict):
    """Forward compositing in f64 from the f32 records and tile lists of `aux`, the walk's decisions as the f32
    restatement takes them (brush_oracle_f64.c).  Returns (out [h,w,4] float64, cond [h,w] float64): the arbiter of
    the pixel tolerance and, per pixel, what a relative error eps of the terms of every sigma can move (eps * cond)."""
    w, h = int(u["img_size"][0]), int(u["img_size"][1])
    out = np.zeros((h, w, 4), np.float64)
    cond = np.zeros((h, w), np.float64)
    s = _Aux()
    for k in ("projected_splats", "num_intersections", "num_visible", "final_index", "cum_tiles_hit",
              "tile_bins", "compact_gid_from_isect", "global_from_compact_gid"):
        setattr(s, k, aux[k].ctypes.data)
    s.max_intersects = int(aux["max_intersects"])
    us = _to_struct(u, int(aux["projected_splats"].shape[0]))
    lib().oracle_rasterize_forward_f64(C.byref(us), C.byref(s), _p(out), _p(cond))
    return out, cond


def render_backward_f64(u: dict, aux: dict, means, log_scales, quats, raw_opac, out_img, v_out, pix_weight=None,
                        final_index_alt=None, out_img_alt=None):
    """The f64 arbiter (brush_oracle_f64.c): same inputs / forward state, every value in double, the walk's
    decisions as the f32 restatement takes them.  Returns float64 dense grads and, under "mag_<name>", the sum of
    the magnitudes of the terms each element is made of (|f32 result - exact| <= eps_per_term * mag) and, under
    "flip_<name>", what the threshold decisions within f32 rounding of flipping can move.  pix_weight ([h,w],
    optional): relative uncertainty of each pixel's forward state (T_final); weight x |terms of the pixel| is added
    to the flip allowance (for gradients computed from a different forward state than the one passed here);
    final_index_alt ([h,w] uint32, optional): that other state's final_index — entries only one of the two walks
    reaches count towards the flip allowance in full; out_img_alt ([h,w,4] f32, optional): that state's image, from
    which the difference of the two runs' recovered T behind a stop mismatch is priced."""
    means, log_scales, quats, raw_opac = _f32(means), _f32(log_scales), _f32(quats), _f32(raw_opac)
    out_img, v_out = _f32(out_img), _f32(v_out)
    n = means.shape[0]
    ncoef = (int(u["sh_degree"]) + 1) ** 2
    g = {
        "v_means": np.zeros((n, 3), np.float64), "v_xy": np.zeros((n, 2), np.float64),
        "v_scales": np.zeros((n, 3), np.float64), "v_quats": np.zeros((n, 4), np.float64),
        "v_sh": np.zeros((n, ncoef, 3), np.float64), "v_opac": np.zeros((n,), np.float64),
    }
    for k in list(g):
        g["mag_" + k[2:]] = np.zeros_like(g[k])
        g["flip_" + k[2:]] = np.zeros_like(g[k])
        g["dep_" + k[2:]] = np.zeros_like(g[k])
    for k in ("means", "scales", "quats", "sh", "opac"):
        g["vjp_" + k] = np.zeros_like(g["v_" + k])
    s = _Aux()
    for k in ("projected_splats", "num_intersections", "num_visible", "final_index", "cum_tiles_hit",
              "tile_bins", "compact_gid_from_isect", "global_from_compact_gid"):
        setattr(s, k, aux[k].ctypes.data)
    s.max_intersects = int(aux["max_intersects"])
    us = _to_struct(u, n)
    lib().oracle_render_backward_f64(C.byref(us), C.byref(s), _p(means), _p(log_scales), _p(quats), _p(raw_opac),
                                     C.c_uint32(n), _p(out_img), _p(v_out), _p(g["v_means"]), _p(g["v_xy"]),
                                     _p(g["v_scales"]), _p(g["v_quats"]), _p(g["v_sh"]), _p(g["v_opac"]),
                                     _p(g["mag_means"]), _p(g["mag_xy"]), _p(g["mag_scales"]), _p(g["mag_quats"]),
                                     _p(g["mag_sh"]), _p(g["mag_opac"]),
                                     _p(g["flip_means"]), _p(g["flip_xy"]), _p(g["flip_scales"]), _p(g["flip_quats"]),
                                     _p(g["flip_sh"]), _p(g["flip_opac"]),
                                     _p(g["vjp_means"]), _p(g["vjp_scales"]), _p(g["vjp_quats"]),
                                     _p(None if pix_weight is None else np.ascontiguousarray(pix_weight, np.float64)),
                                     _p(None if final_index_alt is None else
                                        np.ascontiguousarray(final_index_alt, np.uint32)),
                                     _p(g["dep_means"]), _p(g["dep_xy"]), _p(g["dep_scales"]), _p(g["dep_quats"]),
                                     _p(g["dep_sh"]), _p(g["dep_opac"]), _p(g["vjp_sh"]), _p(g["vjp_opac"]),
                                     _p(None if out_img_alt is None else _f32(out_img_alt)))
    return g
