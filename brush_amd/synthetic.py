"""Seeded synthetic splat clouds.  Plain numpy: bench.py's timed path, the profiling tools and the
tests all generate their inputs here, so neither bench.py nor tools/ import the test package or the CPU checker."""
import math

import numpy as np


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
