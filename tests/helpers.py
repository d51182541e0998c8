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


def synthetic_cloud(n, sh_degree=0, seed=4, mean_mult=1.0, extent=10000.0):
    """Seeded cloud shaped like crates/brush-render/benches/render_bench.rs:32-133."""
    rng = np.random.default_rng(seed)
    means = ((rng.random((n, 3), dtype=np.float32) - 0.5) * np.float32(extent) * np.float32(mean_mult))
    log_scales = np.log(rng.uniform(0.05, 15.0, (n, 3)).astype(np.float32))
    u = rng.random((n, 1), dtype=np.float32)
    v = rng.random((n, 1), dtype=np.float32) * np.float32(2 * math.pi)
    w = rng.random((n, 1), dtype=np.float32) * np.float32(2 * math.pi)
    quats = np.concatenate([np.sqrt(1 - u) * np.sin(v), np.sqrt(1 - u) * np.cos(v),
                            np.sqrt(u) * np.sin(w), np.sqrt(u) * np.cos(w)], axis=1).astype(np.float32)
    ncoef = (sh_degree + 1) ** 2
    sh = rng.uniform(-1.0, 1.0, (n, ncoef, 3)).astype(np.float32)
    raw_opac = rng.random(n, dtype=np.float32)
    return dict(means=means.astype(np.float32), log_scales=log_scales, quats=quats, sh=sh,
                raw_opac=raw_opac)
