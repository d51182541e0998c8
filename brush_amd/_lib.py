"""ctypes binding of include/brush_hip.h.  Fails loudly when libbrush_hip.so is missing."""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# BRUSH_HIP_LIB: load another build of the same ABI (tests/test_gpu_gate.py loads the error-injected twin); it must
# exist just like the default one — there is no fallback of any kind.
LIB_PATH = os.environ.get("BRUSH_HIP_LIB") or os.path.join(_HERE, "lib", "libbrush_hip.so")

BRUSH_OK = 0
AUX_DETERMINISTIC = 1  # BrushAux.flags: BRUSH_AUX_DETERMINISTIC
AUX_ACCUM_ZEROED = 2   # BrushAux.flags: BRUSH_AUX_ACCUM_ZEROED (backward only)
UNIFORM_WORDS = 28
NUM_VISIBLE_WORD = 25
TILE_WIDTH = 16
PROJECTED_FLOATS = 9


class BrushUniforms(C.Structure):
    """helpers.wgsl:7-30"""
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


class BrushLazySh(C.Structure):
    """Deferred Adam of the SH block (include/brush_hip.h: BrushLazySh)."""
    _fields_ = [
        ("table", C.c_void_p),
        ("base", C.c_uint32),
        ("capacity", C.c_uint32),
        ("now", C.c_uint32),
        ("sh_time", C.c_void_p),
        ("sh_moment1", C.c_void_p),
        ("sh_moment2", C.c_void_p),
        ("beta1", C.c_float), ("beta2", C.c_float), ("epsilon", C.c_float),
    ]


class BrushAux(C.Structure):
    """Device-pointer mirror of RenderAux (crates/brush-render/src/lib.rs:20-33)."""
    _fields_ = [
        ("projected_splats", C.c_void_p),
        ("uniforms_buffer", C.c_void_p),
        ("num_intersections", C.c_void_p),
        ("num_visible", C.c_void_p),
        ("final_index", C.c_void_p),
        ("cum_tiles_hit", C.c_void_p),
        ("tile_bins", C.c_void_p),
        ("compact_gid_from_isect", C.c_void_p),
        ("global_from_compact_gid", C.c_void_p),
        ("compact_from_global_gid", C.c_void_p),
        ("overflow", C.c_void_p),
        ("max_intersects", C.c_uint32),
        ("isect_unsorted_pos", C.c_void_p),  # deterministic mode only (NULL otherwise)
        ("flags", C.c_uint32),               # AUX_* bits, per call
        ("bwd_accum", C.c_void_p),           # NULL or the backward's workspace (forward pre-zeroes its accumulators)
        ("lazy_sh", C.POINTER(BrushLazySh)),  # NULL or the deferred-Adam state of the SH block
    ]


class BrushAdamConfig(C.Structure):
    """Hyper-parameters of brush_adam_step (train.rs:184,275-282,336-351)."""
    _fields_ = [
        ("lr_mean", C.c_float), ("lr_scale", C.c_float), ("lr_rotation", C.c_float), ("lr_opac", C.c_float),
        ("lr_coeffs_dc", C.c_float), ("sh_rest_lerp", C.c_float),
        ("beta1", C.c_float), ("beta2", C.c_float), ("epsilon", C.c_float),
        ("time", C.c_uint32),
        ("rotation_grad_wrt_normalized", C.c_uint32),
        ("xy_stat_scale", C.c_float),  # batch_views for view-sharded steps (0 = 1)
        ("lazy_sh", C.POINTER(BrushLazySh)),  # NULL = every SH block is stepped (fused forms only)
    ]


class BrushError(RuntimeError):
    pass


_lib = None

# (name, restype, argtypes) for every symbol include/brush_hip.h declares.
_P = C.c_void_p
_SYMBOLS = [
    ("brush_version", C.c_char_p, []),
    ("brush_status_string", C.c_char_p, [C.c_int]),
    ("brush_last_hip_error", C.c_int, []),
    ("brush_default_max_intersects", C.c_uint32, [C.c_uint32, C.c_uint32, C.c_uint32]),
    ("brush_radix_argsort_workspace_size", C.c_int, [C.c_uint32, C.POINTER(C.c_size_t)]),
    ("brush_radix_argsort_u32", C.c_int,
     [_P, _P, _P, _P, _P, C.c_uint32, C.c_uint32, _P, C.c_size_t, _P]),
    ("brush_inclusive_scan_workspace_size", C.c_int, [C.c_uint32, C.POINTER(C.c_size_t)]),
    ("brush_inclusive_scan_u32", C.c_int, [_P, _P, C.c_uint32, _P, C.c_size_t, _P]),
    ("brush_fwd_workspace_size", C.c_int,
     [C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.POINTER(C.c_size_t)]),
    ("brush_render_forward", C.c_int,
     [C.POINTER(BrushUniforms), _P, _P, _P, _P, _P, C.c_uint32, C.c_int, _P, C.POINTER(BrushAux), _P,
      C.c_size_t, _P]),
    ("brush_rgba8_row_pitch", C.c_uint32, [C.c_uint32]),
    ("brush_render_forward_rgba8", C.c_int,
     [C.POINTER(BrushUniforms), _P, _P, _P, _P, _P, C.c_uint32, _P, C.c_uint32, C.POINTER(BrushAux), _P,
      C.c_size_t, _P]),
    ("brush_deterministic", C.c_int, []),
    ("brush_bwd_workspace_size_flags", C.c_int,
     [C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.POINTER(C.c_size_t)]),
    ("brush_bwd_workspace_size", C.c_int,
     [C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.POINTER(C.c_size_t)]),
    ("brush_bwd_workspace_size_ex", C.c_int,
     [C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.POINTER(C.c_size_t)]),
    ("brush_render_backward", C.c_int,
     [C.POINTER(BrushUniforms), C.POINTER(BrushAux), _P, _P, _P, _P, C.c_uint32, _P, _P, _P, _P, _P, _P,
      _P, _P, _P, C.c_size_t, _P]),
    ("brush_render_backward_records", C.c_int,
     [C.POINTER(BrushUniforms), C.POINTER(BrushAux), _P, _P, _P, _P, C.c_uint32, _P, _P, _P, C.c_uint32, _P,
      C.c_size_t, _P]),
    ("brush_view_index_size", C.c_int, [C.c_uint32, C.c_uint32, C.POINTER(C.c_size_t)]),
    ("brush_reduce_view_records", C.c_int,
     [_P, C.c_uint32, C.c_uint32, _P, _P, _P, _P, C.c_uint32, C.c_uint32, _P, _P, _P, _P, _P, _P, C.c_size_t, _P]),
    ("brush_loss_workspace_size", C.c_int, [C.c_uint32, C.c_uint32, C.POINTER(C.c_size_t)]),
    ("brush_l1_ssim_loss", C.c_int,
     [_P, _P, C.c_uint32, C.c_uint32, C.c_uint32, C.c_float, C.c_uint32, C.c_float, _P, _P, _P, C.c_size_t, _P]),
    ("brush_adam_step", C.c_int,
     [C.POINTER(BrushAdamConfig), C.c_uint32, C.c_uint32, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P]),
    ("brush_render_backward_adam", C.c_int,
     [C.POINTER(BrushUniforms), C.POINTER(BrushAux), C.POINTER(BrushAdamConfig), _P, _P, _P, _P, _P, _P, C.c_uint32,
      _P, _P, _P, _P, _P, _P, _P, _P, _P, C.c_size_t, _P]),
    ("brush_reduce_view_records_adam", C.c_int,
     [_P, C.c_uint32, C.c_uint32, _P, _P, _P, C.POINTER(BrushAdamConfig), C.c_uint32, C.c_uint32, _P, _P, _P, _P, _P,
      C.c_uint32, C.c_uint32, _P, _P, _P, _P, _P, _P, C.c_size_t, _P]),
    ("brush_lazy_sh_flush", C.c_int, [C.POINTER(BrushLazySh), _P, C.c_uint32, C.c_uint32, _P]),
    ("brush_lazy_sh_fill_table", C.c_int,
     [C.c_float, C.c_float, C.c_float, C.c_float, C.c_uint32, C.c_uint32, _P]),
    ("brush_normalize_quats", C.c_int, [_P, _P, C.c_uint32, _P]),
    ("brush_refine_stats", C.c_int,
     [C.POINTER(BrushAux), _P, C.c_uint32, C.c_uint32, C.c_uint32, _P, _P, _P]),
    ("brush_profiler_create", C.c_int, [C.POINTER(_P)]),
    ("brush_profiler_destroy", None, [_P]),
    ("brush_profiler_attach", None, [_P]),
    ("brush_profiler_read", C.c_int, [_P, C.POINTER(C.c_float)]),
    ("brush_profiler_stop_after", C.c_int, [_P, C.c_int]),
    ("brush_stage_name", C.c_char_p, [C.c_int]),
]

NUM_STAGES = 11

SYMBOL_NAMES = [s[0] for s in _SYMBOLS]


def lib():
    """Load libbrush_hip.so (after torch, so both share one HIP runtime) and bind symbols."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise BrushError(
            f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "or `make -C brush_amd/csrc`. brush_amd has no CPU fallback.")
    import torch  # noqa: F401  (loads torch's libamdhip64.so first; same SONAME is then reused)

    if os.environ.get("BRUSH_HIP_LIB"):
        import warnings

        warnings.warn(f"brush_amd: BRUSH_HIP_LIB is set, loading {LIB_PATH} instead of the product library",
                      RuntimeWarning, stacklevel=2)
    handle = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)
    for name, restype, argtypes in _SYMBOLS:
        fn = getattr(handle, name)  # AttributeError if the header and the library diverge
        fn.restype = restype
        fn.argtypes = argtypes
    _lib = handle
    return _lib


def check(status: int, what: str):
    if status != BRUSH_OK:
        l = lib()
        msg = l.brush_status_string(status).decode()
        raise BrushError(f"{what} failed: {msg} (status {status}, hipError {l.brush_last_hip_error()})")


def version() -> str:
    return lib().brush_version().decode()
