"""Shared test helpers: golden-case loading, the reference test's camera, synthetic clouds."""
import math
import os

import numpy as np

from oracle import oracle as O

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_case(name):
    from safetensors.numpy import load_file

    return load_file(os.path.join(GOLDEN, f"{name}.safetensors"))


def crab_rgb():
    return np.load(os.path.join(GOLDEN, "crab_rgb_u8.npy")).astype(np.float32) / 255.0


def reference_test_camera(w, h):
    """Camera of crates/brush-render/src/render.rs:734-746: (0,0,-8), identity, fov 90deg on x."""
    fov = math.pi * 0.5
    focal = O.fov_to_focal(fov, w)
    return dict(position=[0.0, 0.0, -8.0], rotation_xyzw=[0.0, 0.0, 0.0, 1.0],
                fov_x=O.focal_to_fov(focal, w), fov_y=O.focal_to_fov(focal, h), center_uv=[0.5, 0.5])


def reference_test_uniforms(w, h, sh_degree):
    c = reference_test_camera(w, h)
    return O.make_uniforms(c["position"], c["rotation_xyzw"], c["fov_x"], c["fov_y"], c["center_uv"],
                           [w, h], sh_degree)


def all_close_report(a, b, rtol, atol):
    """burn's all_close (|a-b| <= atol + rtol*|b|); returns (ok, max_abs_err, n_bad)."""
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    err = np.abs(a - b)
    bad = err > (atol + rtol * np.abs(b))
    return (not bad.any()), float(err.max() if err.size else 0.0), int(bad.sum())


def unnormalised_quat_grad(quats, v_quats):
    """VJP of q/|q| (crates/brush-render/src/gaussian_splats.rs:173-175) applied outside the op."""
    q = np.asarray(quats, np.float64)
    v = np.asarray(v_quats, np.float64)
    nrm = np.linalg.norm(q, axis=1, keepdims=True)
    return (v - q * np.sum(q * v, axis=1, keepdims=True) / nrm ** 2) / nrm


from brush_amd.synthetic import synthetic_cloud  # noqa: E402,F401  (re-exported for the tests)

